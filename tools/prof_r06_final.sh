#!/bin/bash
# The whole evidence pass of round 6 on the final binary (run through gpurun): tools/prof_r06.sh for the nine workloads with counter passes, the three
# run-time-shape lines (kernel trace + bench line), the driver's own command, and the stand-alone saturation lines with and without the cost-ordered launch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/prof_r06.sh > $O/prof_r06.log 2>&1
PROF_W="lv_tanh5 lv_shape8 seir_shape63" PROF_PMC=0 bash tools/prof_r06.sh > $O/prof_r06_shapes.log 2>&1
cd $R
python bench.py > $O/r06_bench_default_driver_command.json 2> $O/r06_bench_default.err
python bench.py --traj 40000 --steps 20 --warmup 3 --no-others --no-cpu-baseline > $O/r06_bench_lv_sat40k.json 2>/dev/null
python bench.py --traj 160000 --steps 10 --warmup 2 --no-others --no-cpu-baseline > $O/r06_bench_lv_sat160k.json 2>/dev/null
UDE_COST_SORT=0 python bench.py --traj 40000 --steps 20 --warmup 3 --no-others --no-cpu-baseline > $O/r06_bench_lv_sat40k_unsorted.json 2>/dev/null
UDE_COST_SORT=0 python bench.py --traj 160000 --steps 10 --warmup 2 --no-others --no-cpu-baseline > $O/r06_bench_lv_sat160k_unsorted.json 2>/dev/null
ls $O | grep r06_ | wc -l
