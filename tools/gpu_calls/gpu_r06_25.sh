#!/bin/bash
# round 6, call 25: call 14's candidate once more, now that k_dynamics no longer depends on what the launch before it left behind -- the wide solver class's workers as the first
# workgroups of k_csolve's own grid (16-coordinate templates; MSK_WIDE_IN_CSOLVE=0: the launch of its own) -- parity nodes of the wide class, then A/B inside one library and against 687fa9c's
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_25; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_wide_solver.py tests/test_gpu_parity.py tests/test_contact_trimming.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; S="MSK_LIB=maniskill_amd/csrc/libmsk_physx.so MSK_WIDE_IN_CSOLVE=0"; L=MSK_LIB=maniskill_amd/csrc/libmsk_prev.so
( run merged_1 $N; run separate_1 $S; run prev_1 $L; run merged_2 $N; run separate_2 $S; run prev_2 $L
  STEPS=20 WARM=5 run merged_20steps $N; STEPS=20 WARM=5 run separate_20steps $S; STEPS=20 WARM=5 run prev_20steps $L; STEPS=20 WARM=5 run merged_20steps_b $N; STEPS=20 WARM=5 run prev_20steps_b $L
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_merged $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_prev $L
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_merged $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_prev $L
  STEPS=300 EXTRA="--envs 512" run 512_merged $N; STEPS=300 EXTRA="--envs 512" run 512_prev $L
  STEPS=300 EXTRA="--envs 65536" run 65536_merged $N; STEPS=300 EXTRA="--envs 65536" run 65536_prev $L ) | tee $O/ab_wide_class_inside_csolve_again.log
