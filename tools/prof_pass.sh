#!/bin/bash
# One profiling pass of the C2 workload on the MI355X box (run through gpurun): bench lines, rocprofv3 kernel trace,
# PMC passes (one counter group per run, --kernel-trace only).  Outputs under gpurun_out/; copy the summaries to profiles/.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/v4_bench.json 2> $O/v4_bench.err
python bench.py --steps 20 --warmup 3 --sensealg discrete --no-cpu-baseline > $O/v4_bench_discrete.json 2>/dev/null
python bench.py --workload seir --steps 3 --warmup 1 --no-cpu-baseline > $O/v4_bench_seir.json 2>/dev/null
python bench.py --workload kpp --steps 3 --warmup 1 --no-cpu-baseline > $O/v4_bench_kpp.json 2>/dev/null
python bench.py --alg vern7 --steps 5 --warmup 2 --no-cpu-baseline > $O/v4_bench_lv_vern7.json 2>/dev/null
python tools/pcie_rate.py > $O/v4_pcie.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_v4 -o v4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/v4_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc4_fetch -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4_write -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc4_sq1 -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc4_sq2 -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_v4 -name "*.db" | head -1) $O/v4_kernel_stats.md > /dev/null 2>$O/v4_ks.err
python tools/pmc_summary.py $O/v4_pmc.md $(find $O/pmc4_fetch $O/pmc4_write $O/pmc4_sq1 $O/pmc4_sq2 -name "*.db") > /dev/null 2>$O/v4_pmc.err
ls $O
