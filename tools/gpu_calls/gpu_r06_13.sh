#!/bin/bash
# round 6, call 13: three bit-equal instruction-count candidates in one library -- (a) the box-box manifold's feature planes (Newell normal + centroid) computed once per feature
# instead of once per clipped point (msk_collide_lane.h), (b) the solver's sweeps specialised on "no torsional block in this wavefront" (msk_solve.h), (c) PushT's observation
# with the link frames in one launch -- parity nodes, then A/B on this box against the library of the second evidence run (libmsk_r06ev.so = 0d903ca's csrc)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_13; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_fused_step.py tests/test_step_graph.py tests/test_render.py tests/test_wide_solver.py tests/test_hull_heaps.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_r06ev.so
( run new_1 $N; run old_1 $L; run new_2 $N; run old_2 $L
  STEPS=20 WARM=5 run new_20steps $N; STEPS=20 WARM=5 run old_20steps $L
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_new $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_old $L
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_old $L
  STEPS=300 EXTRA="--envs 512" run 512_new $N; STEPS=300 EXTRA="--envs 512" run 512_old $L ) | tee $O/ab_instruction_count_batch.log
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; grep "boxbox\|k_dynamics phases\|n=" $O/phase_probe_pickcube.log | cut -c1-330
