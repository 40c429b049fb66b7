#!/bin/bash
# round 5, the round's evidence on ONE commit: the driver's own commands (pytest -m gpu -x -q sequentially, smoke, bench.py --gpus 1 --steps 20 --warmup 5),
# then the bench forms, rocprofv3 kernel statistics and the PMC passes (summarised here: only the summaries travel back)
#   gpurun --timeout 2400 -- 'bash tools/gpu_r05_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_driver_form.log 2>&1; tail -6 $O/gpu_tests_driver_form.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1_driver_form.err; tail -c 300 $O/bench_n1_driver_form.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1_default_1000.json 2> $O/bench_n1_default.err; tail -c 300 $O/bench_n1_default_1000.json
timeout 200 python bench.py --envs 512 --no-cpu-baseline --no-extras > $O/bench_n1_512envs.json 2>/dev/null
timeout 200 python bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 200 --no-cpu-baseline --no-extras > $O/bench_pusht_camera_4096.json 2>/dev/null
timeout 200 python bench.py --env PegInsertionSide-v1 --steps 300 --no-cpu-baseline --no-extras > $O/bench_peg_insertion_4096.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_graph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_late -- python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_late.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pusht_cam -- python $R/bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_pusht_cam.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_peg -- python $R/bench.py --env PegInsertionSide-v1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_peg.log 2>&1
find $O -name '*kernel_trace.csv' -delete
cd $R
rm -rf $R/gpurun_out/pmc; timeout 500 bash tools/pmc_collect.sh > /dev/null 2>&1
python tools/pmc_summarise.py $R/gpurun_out/pmc $O/pmc_counters_4096.json "python bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras" r05-final > $O/pmc_summary.log 2>&1; tail -12 $O/pmc_summary.log
rm -rf $R/gpurun_out/pmc; timeout 500 bash tools/pmc_collect.sh --env PushT-v1 --obs-mode depth+segmentation > /dev/null 2>&1
python tools/pmc_summarise.py $R/gpurun_out/pmc $O/pmc_counters_camera_4096.json "python bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras --env PushT-v1 --obs-mode depth+segmentation" r05-final > $O/pmc_camera_summary.log 2>&1; tail -6 $O/pmc_camera_summary.log
rm -rf $R/gpurun_out/pmc
find $O -name "*kernel_stats.csv"
