#!/bin/bash
# Produces the round's measurement artefacts on the GPU box (gpurun_out/r03/...): bench lines, rocprofv3 kernel stats, PMC passes.
#   bash tools/gpu_round_artifacts.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
python bench.py --envs 512 --no-cpu-baseline > $O/bench_n1_512envs.json 2>/dev/null
python bench.py --steps 300 --control-freq 50 --no-cpu-baseline --no-extras > $O/bench_n1_control50hz.json 2>/dev/null
python bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 200 --no-cpu-baseline --no-extras > $O/bench_pusht_camera_4096.json 2>/dev/null
python bench.py --env PegInsertionSide-v1 --steps 300 --no-cpu-baseline --no-extras > $O/bench_peg_insertion_4096.json 2>/dev/null
python tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 1024 --steps 50 > $O/bench_reference_host_pusht_camera_1024.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_late -- python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_late.log 2>&1
cd $R && bash tools/pmc_collect.sh > /dev/null 2>&1
mkdir -p $O/pmc && cp -r $R/gpurun_out/pmc/* $O/pmc/ 2>/dev/null
find $O -name "*kernel_stats.csv" | head
tail -c 600 $O/bench_n1_default.json
