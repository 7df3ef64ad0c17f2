#!/bin/bash
# Round-6 profiling pass on the MI355X box (run through gpurun): for every bench workload a rocprofv3 kernel trace (--stats), the PMC
# passes and then the bench line (one counter group per run, --kernel-trace only; FETCH_SIZE / WRITE_SIZE in
# their own runs).  Summaries land in gpurun_out/; the ones committed under profiles/ are copies of these files.
#   PROF_W="seir node" refreshes only those workloads; PROF_PMC=0 skips the counter passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
WL=${PROF_W:-"lv seir seir_fast kpp hjb node node_fast lv_tanh32 lv_discrete"}
has() { case " $WL " in *" $1 "*) return 0;; *) return 1;; esac; }
args() { case $1 in lv_tanh32) echo "--workload lv --net tanh32";; lv_tanh5) echo "--workload lv --net tanh5";; lv_shape8) echo "--workload lv --net shape8";;
          seir_shape63) echo "--workload seir --net shape63";; lv_discrete) echo "--workload lv --sensealg discrete";; seir_fast) echo "--workload seir --sensealg fast";;
          node_fast) echo "--workload node --sensealg fast";; *) echo "--workload $1";; esac; }
cd /tmp
for W in $WL; do
  case $W in lv*) ST="--steps 10 --warmup 2";; *) ST="--steps 2 --warmup 1";; esac
  B="python $R/bench.py $(args $W) $ST --no-cpu-baseline --no-others"
  rocprofv3 --kernel-trace --stats -d $O/p6_${W}_kt -o kt -- $B > $O/p6_${W}_kt.log 2>&1
  ( cd $R; python tools/rocpd_summary.py $(find $O/p6_${W}_kt -name "*.db" | head -1) $O/r06_kernel_stats_${W}.md > /dev/null 2>>$O/p6.err )
  if [ "${PROF_PMC:-1}" = 1 ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p6_${W}_f -o p -- $B > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p6_${W}_w -o p -- $B > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p6_${W}_1 -o p -- $B > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p6_${W}_2 -o p -- $B > /dev/null 2>&1
    ( cd $R; python tools/pmc_summary.py $O/r06_pmc_${W}.md $(find $O/p6_${W}_f $O/p6_${W}_w $O/p6_${W}_1 $O/p6_${W}_2 -name "*.db") > /dev/null 2>>$O/p6.err )
  fi
done
# the bench lines come LAST: bench.py takes roofline.traffic from profiles/r06_pmc_<workload>.md, i.e. from the counter passes just made
# (copied into this box's checkout; the committed copies are the same files, merged back through gpurun_out/)
cp $O/r06_pmc_*.md $R/profiles/ 2>/dev/null
cd $R
has lv && python bench.py --steps 20 --warmup 3 > $O/r06_bench_lv.json 2> $O/r06_bench.err
for W in $WL; do
  [ $W = lv ] && continue
  case $W in lv_*) ST="--steps 20 --warmup 3";; *) ST="--steps 3 --warmup 1";; esac
  python bench.py $(args $W) $ST --no-others > $O/r06_bench_$W.json 2>/dev/null
done
has hjb && python bench.py --workload hjb --steps 5 --warmup 2 --traj 8192 --no-cpu-baseline > $O/r06_bench_hjb_8k.json 2>/dev/null
rm -rf $O/p6_*_kt $O/p6_*_f $O/p6_*_w $O/p6_*_1 $O/p6_*_2
ls $O | head -60; for W in $WL; do head -5 $O/r06_kernel_stats_${W}.md; done
