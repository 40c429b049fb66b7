#!/bin/bash
# round 6, call 24: the narrowphase's rows dispatched hull first, then box-box, plane last (call 23: the workgroups queue for the one-per-SIMD slots, and the longest of them -- hull items --
# started last) against the library before it (libmsk_prev.so = 4a09b58's kernels); parity nodes; the launch-position probe of PegInsertionSide
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_24; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_hull_heaps.py tests/test_wide_solver.py tests/test_contact_trimming.py tests/test_fused_step.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_prev.so
( run new_1 $N; run prev_1 $L; run new_2 $N; run prev_2 $L
  STEPS=20 WARM=5 run new_20steps $N; STEPS=20 WARM=5 run prev_20steps $L; STEPS=20 WARM=5 run new_20steps_b $N; STEPS=20 WARM=5 run prev_20steps_b $L
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_new $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_prev $L
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_prev $L
  STEPS=300 EXTRA="--envs 512" run 512_new $N; STEPS=300 EXTRA="--envs 512" run 512_prev $L
  STEPS=300 EXTRA="--envs 2048 --env PegInsertionSide-v1" run peg2048_new $N; STEPS=300 EXTRA="--envs 2048 --env PegInsertionSide-v1" run peg2048_prev $L
  STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_new $N; STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_prev $L
  STEPS=300 EXTRA="--envs 65536" run 65536_new $N; STEPS=300 EXTRA="--envs 65536" run 65536_prev $L ) | tee $O/ab_narrowphase_hull_rows_first.log
PROBE_LIB=libmsk_prof_nostats.so PROBE_ENV=Peg PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_peg.log 2>&1; grep "k_narrowphase:\|kind row [0-6]:" $O/phase_probe_peg.log | cut -c1-330
