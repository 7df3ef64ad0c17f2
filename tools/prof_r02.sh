#!/bin/bash
# Round-2 profiling pass on the MI355X box (run through gpurun): for every bench workload the bench line, a rocprofv3
# kernel trace (--stats) and the PMC passes (one counter group per run, --kernel-trace only; FETCH_SIZE / WRITE_SIZE in
# their own runs).  Summaries land in gpurun_out/; the ones committed under profiles/ are copies of these files.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
WL=${PROF_W:-"lv seir kpp hjb node"}   # PROF_W="seir node": refresh only those workloads
has() { case " $WL " in *" $1 "*) return 0;; *) return 1;; esac; }
has lv && python bench.py --steps 20 --warmup 3 > $O/r02_bench_lv.json 2> $O/r02_bench.err
has seir && python bench.py --workload seir --steps 3 --warmup 1 > $O/r02_bench_seir.json 2>/dev/null
has kpp && python bench.py --workload kpp --steps 3 --warmup 1 > $O/r02_bench_kpp.json 2>/dev/null
has hjb && python bench.py --workload hjb --steps 5 --warmup 2 > $O/r02_bench_hjb_16k.json 2>/dev/null
has hjb && python bench.py --workload hjb --steps 5 --warmup 2 --traj 8192 --no-cpu-baseline > $O/r02_bench_hjb.json 2>/dev/null
has lv && python bench.py --steps 20 --warmup 3 --sensealg discrete --no-cpu-baseline > $O/r02_bench_lv_discrete.json 2>/dev/null
has lv && python bench.py --steps 20 --warmup 3 --sensealg fast --no-cpu-baseline > $O/r02_bench_lv_fast.json 2>/dev/null
has node && python bench.py --workload node --steps 3 --warmup 1 > $O/r02_bench_node.json 2>/dev/null
cd /tmp
for W in $WL; do
  case $W in lv) ST="--steps 10 --warmup 2";; *) ST="--steps 2 --warmup 1";; esac
  B="python $R/bench.py --workload $W $ST --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats -d $O/p2_${W}_kt -o kt -- $B > $O/p2_${W}_kt.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p2_${W}_f -o p -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2_${W}_w -o p -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p2_${W}_1 -o p -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p2_${W}_2 -o p -- $B > /dev/null 2>&1
  ( cd $R; python tools/rocpd_summary.py $(find $O/p2_${W}_kt -name "*.db" | head -1) $O/r02_kernel_stats_${W}.md > /dev/null 2>>$O/p2.err
    python tools/pmc_summary.py $O/r02_pmc_${W}.md $(find $O/p2_${W}_f $O/p2_${W}_w $O/p2_${W}_1 $O/p2_${W}_2 -name "*.db") > /dev/null 2>>$O/p2.err )
done
rm -rf $O/p2_*_kt $O/p2_*_f $O/p2_*_w $O/p2_*_1 $O/p2_*_2 $O/pmch_* $O/prof_hjb
ls $O | head -40; for W in $WL; do head -4 $O/r02_kernel_stats_${W}.md; done
