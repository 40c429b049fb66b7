#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE configs (PushT camera, PegInsertionSide, OpenCabinetDrawer over the shim): run on the GPU box from the repo root;
# the kernel traces are deleted (gpurun copies at most 64 MiB back), the *_kernel_stats.csv files stay under gpurun_out/r03cfg/.
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03cfg
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_pusht_cam -- python $GRAFT_REPO_ROOT/bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_pusht_cam.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_peg -- python $GRAFT_REPO_ROOT/bench.py --env PegInsertionSide-v1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_peg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_cabinet -- python $GRAFT_REPO_ROOT/tools/gpu_cabinet_probe.py bench 1024 > $GRAFT_REPO_ROOT/gpurun_out/r03cfg/prof_cabinet.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r03cfg -name '*kernel_trace.csv' -delete
for d in prof_pusht_cam prof_peg prof_cabinet; do echo == $d; head -8 $GRAFT_REPO_ROOT/gpurun_out/r03cfg/$d/*/*_kernel_stats.csv | cut -c1-120; done
