#!/bin/bash
# round 6, call 22: the capacity of the packed solver class (msk_set_solver_classes): should envs of 9-16 blocks go to the one-env-per-wavefront class instead of stretching a packed wavefront?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_22; mkdir -p $O
cd $R
timeout 600 python tools/gpu_class_cap_probe.py PickCube 4096 16,12,10,8,6 2>&1 | grep "class 0" | tee $O/class_cap_pickcube.log
timeout 600 python tools/gpu_class_cap_probe.py Peg 4096 16,12,10,8 2>&1 | grep "class 0" | tee $O/class_cap_peg.log
