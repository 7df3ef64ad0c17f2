#!/bin/bash
# Profiling pass of the N1 (hjb) workload on the MI355X box (run through gpurun): kernel trace + PMC passes
# (one counter group per run, --kernel-trace only).  Outputs under gpurun_out/; copy the summaries to profiles/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02}; mkdir -p $O; cd /tmp
B="python $R/bench.py --workload hjb --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/prof_hjb -o hjb -- $B > $O/${T}_hjb_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmch_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmch_w -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmch_1 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmch_2 -o p -- $B > $O/pmch_2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $O/pmch_3 -o p -- $B > $O/pmch_3.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_hjb -name "*.db" | head -1) $O/${T}_kernel_stats_hjb.md > /dev/null 2>$O/hjb_ks.err
python tools/pmc_summary.py $O/${T}_pmc_hjb.md $(find $O/pmch_f $O/pmch_w $O/pmch_1 $O/pmch_2 $O/pmch_3 -name "*.db") > /dev/null 2>$O/hjb_pmc.err
head -60 $O/${T}_pmc_hjb.md; head -20 $O/${T}_kernel_stats_hjb.md
