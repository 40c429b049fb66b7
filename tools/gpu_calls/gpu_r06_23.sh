#!/bin/bash
# round 6, call 23: where the time of a k_narrowphase workgroup goes (five stamps inside it) and where in the launch it runs (the 100 MHz clock at each workgroup's start and end, profiling build), PickCube and PegInsertionSide
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_23; mkdir -p $O
cd $R
PROBE_LIB=libmsk_prof_nostats.so PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; grep "first start\|first wave\|kind row" $O/phase_probe_pickcube.log | cut -c1-420
PROBE_LIB=libmsk_prof_nostats.so PROBE_ENV=Peg PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_peg.log 2>&1; grep "first start\|first wave\|kind row" $O/phase_probe_peg.log | cut -c1-420
