#!/bin/bash
# round 5, GPU call 1: the whole -m gpu suite on HEAD (round 4 ended red after 14 of 140), then the driver's bench line with every leg, kernel statistics.
#   gpurun --timeout 1500 -- 'bash tools/gpu_calls/gpu_r05_1.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider > $O/gpu_tests_all.log 2>&1; tail -40 $O/gpu_tests_all.log
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1_driver_form.json
timeout 200 python bench.py --steps 1000 --no-cpu-baseline > $O/bench_n1_1000.json 2>> $O/bench_n1.err; tail -c 600 $O/bench_n1_1000.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
find $O -name '*kernel_trace.csv' -size +8M -delete
find $O -name '*kernel_stats.csv' | head -1 | xargs -r head -12
