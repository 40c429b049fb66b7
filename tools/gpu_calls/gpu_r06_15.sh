#!/bin/bash
# round 6, call 15: k_dynamics<32,16> held to 128 VGPRs (four wavefronts per SIMD instead of three: the 3072 wavefronts of 4096 envs no longer fill the chip exactly, call 14's
# 100 MHz stamps showed workgroups starting 21-23 us late) -- parity nodes, A/B on this box against the library of the third evidence run (libmsk_base.so = 72d5aae's csrc),
# the launch-position probe in the early and the late regime
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_15; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_step_graph.py tests/test_fused_step.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, "gap/substep %.1f" % r.get("launch_gap_us_per_substep", -1), (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_base.so
( run new_1 $N; run old_1 $L; run new_2 $N; run old_2 $L
  STEPS=20 WARM=5 run new_20steps $N; STEPS=20 WARM=5 run old_20steps $L; STEPS=20 WARM=5 run new_20steps_b $N; STEPS=20 WARM=5 run old_20steps_b $L
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_new $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_old $L
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_old $L
  STEPS=300 EXTRA="--envs 512" run 512_new $N; STEPS=300 EXTRA="--envs 512" run 512_old $L
  STEPS=300 EXTRA="--envs 2048" run 2048_new $N; STEPS=300 EXTRA="--envs 2048" run 2048_old $L
  STEPS=300 EXTRA="--envs 8192" run 8192_new $N; STEPS=300 EXTRA="--envs 8192" run 8192_old $L
  STEPS=300 EXTRA="--envs 65536" run 65536_new $N; STEPS=300 EXTRA="--envs 65536" run 65536_old $L ) | tee $O/ab_dynamics_four_waves_per_simd.log
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; grep "first wave\|k_dynamics phases" $O/phase_probe_pickcube.log | cut -c1-520
