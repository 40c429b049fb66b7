#!/bin/bash
# round 6, call 16: k_dynamics with 256-thread workgroups (three dynamics wavefronts + the broadphase wavefront of their three blocks: one wavefront per SIMD per workgroup,
# 683 workgroups for 4096 envs = 89 % of the wavefront slots at 154 VGPRs) against the 192-thread form (1024 x 3 = the slots exactly) and against call 15's 128-VGPR library
# (libmsk_tight.so), via the MSK_DYN_THREADS knob of one library; parity nodes with the knob; the launch-position probe
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_16; mkdir -p $O
cd $R
MSK_DYN_THREADS=256 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_step_graph.py -m gpu -x -q > $O/pytest_parity_256.log 2>&1; echo "pytest parity (256 threads) rc $?"; tail -3 $O/pytest_parity_256.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, "gap/substep %.1f" % r.get("launch_gap_us_per_substep", -1), (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; T=MSK_LIB=maniskill_amd/csrc/libmsk_tight.so; W=MSK_DYN_THREADS=256
( run t256_1 $N $W; run t192_1 $N; run tight_1 $T; run t256_2 $N $W; run t192_2 $N; run tight_2 $T
  STEPS=20 WARM=5 run t256_20steps $N $W; STEPS=20 WARM=5 run t192_20steps $N; STEPS=20 WARM=5 run t256_20steps_b $N $W; STEPS=20 WARM=5 run t192_20steps_b $N
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_t256 $N $W; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_t192 $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_tight $T
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_t256 $N $W; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_t192 $N
  STEPS=300 EXTRA="--envs 2048" run 2048_t256 $N $W; STEPS=300 EXTRA="--envs 2048" run 2048_t128 $N; STEPS=300 EXTRA="--envs 2048" run 2048_t192 $N MSK_DYN_THREADS=192
  STEPS=300 EXTRA="--envs 3072" run 3072_t256 $N $W; STEPS=300 EXTRA="--envs 3072" run 3072_default $N
  STEPS=300 EXTRA="--envs 512" run 512_t256 $N $W; STEPS=300 EXTRA="--envs 512" run 512_t128 $N ) | tee $O/ab_dynamics_256_threads.log
MSK_DYN_THREADS=256 PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube_256.log 2>&1; grep "first wave\|k_dynamics phases" $O/phase_probe_pickcube_256.log | cut -c1-520
