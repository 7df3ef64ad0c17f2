#!/bin/bash
# debug: binary search for the register whose stale lanes the node adjoint reads (UDE_EXP_POISON kind 5), lib in $1
L=$1; lo=${2:-256}; hi=${3:-384}
probe() { UDE_EXP_LIB=$L UDE_EXP_POISON=5,1,$1,$2 python tests/debug/dbg_node_fill.py 2>&1 | tail -1 | sed "s/.* : //" | grep -c "b"; }
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(((lo + hi) / 2))
  if [ "$(probe $lo $mid)" != "0" ]; then hi=$mid; else
    if [ "$(probe $mid $hi)" != "0" ]; then lo=$mid; else echo "neither half alone fails: [$lo,$mid) [$mid,$hi)"; break; fi
  fi
  echo "range [$lo,$hi)"
done
echo "RESULT [$lo,$hi)"
