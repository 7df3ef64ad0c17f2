export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
python $R/bench.py --workload kpp --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmck_1 -o p -- python $R/bench.py --workload kpp --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmck_2 -o p -- python $R/bench.py --workload kpp --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH -d $O/pmck_3 -o p -- python $R/bench.py --workload kpp --steps 1 --warmup 0 --no-cpu-baseline > $O/pmck_3.log 2>&1
cd $R; python tools/pmc_summary.py $O/kpp_pmc.md $(find $O/pmck_1 $O/pmck_2 $O/pmck_3 -name "*.db") > /dev/null 2>$O/kpp_pmc.err; head -45 $O/kpp_pmc.md
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/pmck_4 -o p -- python $R/bench.py --workload kpp --steps 1 --warmup 0 --no-cpu-baseline > $O/pmck_4.log 2>&1
cd $R; python tools/pmc_summary.py $O/kpp_pmc_mfma.md $(find $O/pmck_4 -name "*.db") > /dev/null 2>$O/kpp_pmc_mfma.err; head -12 $O/kpp_pmc_mfma.md; tail -3 $O/pmck_4.log
