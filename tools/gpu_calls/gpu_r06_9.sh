#!/bin/bash
# round 6, call 9: the wide solver class's launch beside k_csolve (a side stream: two branches of the captured graph) -- parity nodes, then A/B on this one box by the
# environment switch MSK_WIDE_CONCURRENT (0 = behind k_csolve, as in the calls before), 1000-step and driver-form passes, PegInsertionSide
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_9; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_wide_solver.py tests/test_step_graph.py tests/test_device_reset.py tests/test_vector_env.py tests/test_fused_step.py tests/test_reference_conformance.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()})
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
( run beside_1 MSK_WIDE_CONCURRENT=1; run behind_1 MSK_WIDE_CONCURRENT=0; run beside_2 MSK_WIDE_CONCURRENT=1; run behind_2 MSK_WIDE_CONCURRENT=0
  STEPS=20 WARM=5 run beside_20steps MSK_WIDE_CONCURRENT=1; STEPS=20 WARM=5 run behind_20steps MSK_WIDE_CONCURRENT=0
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_beside MSK_WIDE_CONCURRENT=1; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_behind MSK_WIDE_CONCURRENT=0
  STEPS=300 EXTRA="--envs 512" run 512_beside MSK_WIDE_CONCURRENT=1; STEPS=300 EXTRA="--envs 512" run 512_behind MSK_WIDE_CONCURRENT=0 ) | tee $O/ab_wide_beside_vs_behind.log
