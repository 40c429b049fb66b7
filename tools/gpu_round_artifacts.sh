#!/bin/bash
# Produces the round's measurement artefacts on the GPU box (gpurun_out/$ART_DIR, default r06art): bench lines, rocprofv3 kernel stats, PMC passes.
#   bash tools/gpu_round_artifacts.sh [quick]        (quick: without the PMC passes and the secondary configs)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${ART_DIR:-r06art}
mkdir -p $O
cd $R
python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1_driver_form.json 2>/dev/null
python bench.py --envs 512 --no-cpu-baseline > $O/bench_n1_512envs.json 2>/dev/null
python bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 200 --no-cpu-baseline --no-extras > $O/bench_pusht_camera_4096.json 2>/dev/null
python bench.py --env PegInsertionSide-v1 --steps 300 --no-cpu-baseline --no-extras > $O/bench_peg_insertion_4096.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_late -- python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_late.log 2>&1
if [ "${1:-}" != "quick" ]; then
  cd $R
  python bench.py --steps 300 --control-freq 50 --no-cpu-baseline --no-extras > $O/bench_n1_control50hz.json 2>/dev/null
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pusht_cam -- python $R/bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_pusht_cam.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_peg -- python $R/bench.py --env PegInsertionSide-v1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_peg.log 2>&1
  cd $R && bash tools/pmc_collect.sh > /dev/null 2>&1
  mkdir -p $O/pmc && cp -r $R/gpurun_out/pmc/* $O/pmc/ 2>/dev/null
  rm -rf $R/gpurun_out/pmc
  bash tools/pmc_collect.sh --env PushT-v1 --obs-mode depth+segmentation > /dev/null 2>&1
  mkdir -p $O/pmc_camera && cp -r $R/gpurun_out/pmc/* $O/pmc_camera/ 2>/dev/null
fi
find $O -name '*kernel_trace.csv' -size +8M -delete     # gpurun copies at most 64 MiB back: the statistics stay
find $O -name "*kernel_stats.csv" | head
tail -c 900 $O/bench_n1_default.json
