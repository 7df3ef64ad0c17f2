#!/bin/bash
# the lv part of tools/prof_r02.sh alone (re-run after a change of the LV kernels)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/r02_bench_lv.json 2> $O/r02_bench.err
python bench.py --steps 20 --warmup 3 --sensealg discrete --no-cpu-baseline > $O/r02_bench_lv_discrete.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --sensealg fast --no-cpu-baseline > $O/r02_bench_lv_fast.json 2>/dev/null
cd /tmp
W=lv
B="python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/p2_${W}_kt -o kt -- $B > $O/p2_${W}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p2_${W}_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2_${W}_w -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p2_${W}_1 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p2_${W}_2 -o p -- $B > /dev/null 2>&1
( cd $R; python tools/rocpd_summary.py $(find $O/p2_${W}_kt -name "*.db" | head -1) $O/r02_kernel_stats_${W}.md > /dev/null 2>>$O/p2.err
  python tools/pmc_summary.py $O/r02_pmc_${W}.md $(find $O/p2_${W}_f $O/p2_${W}_w $O/p2_${W}_1 $O/p2_${W}_2 -name "*.db") > /dev/null 2>>$O/p2.err )
rm -rf $O/p2_*_kt $O/p2_*_f $O/p2_*_w $O/p2_*_1 $O/p2_*_2
head -6 $O/r02_kernel_stats_${W}.md; grep -E "SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU|FETCH_SIZE|WRITE_SIZE" $O/r02_pmc_${W}.md | head -6
