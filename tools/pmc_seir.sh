export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
python $R/bench.py --workload seir --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmcs_1 -o p -- python $R/bench.py --workload seir --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmcs_2 -o p -- python $R/bench.py --workload seir --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH -d $O/pmcs_3 -o p -- python $R/bench.py --workload seir --steps 2 --warmup 1 --no-cpu-baseline > $O/pmcs_3.log 2>&1
cd $R; python tools/pmc_summary.py $O/seir_pmc.md $(find $O/pmcs_1 $O/pmcs_2 $O/pmcs_3 -name "*.db") > /dev/null 2>$O/seir_pmc.err; head -45 $O/seir_pmc.md
