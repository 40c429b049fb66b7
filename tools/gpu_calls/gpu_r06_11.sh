#!/bin/bash
# round 6, call 11: PegInsertionSide's observation behind a control step as ONE launch with the link frames (k_peg_observe_kin) and with the prepared finger-peg pair lists
# (it was k_kinematics + a lane-per-env kernel scanning the whole pair table twice: 13 + 46 us): the parity nodes that touch the task kernels, then A/B on this box against
# the evidence run's library (libmsk_r06ev.so = HEAD's csrc before the change), config 4 on the fused host and over the reference API
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_11; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_fused_step.py tests/test_device_reset.py tests/test_step_graph.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -3 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --env PegInsertionSide-v1 --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()})
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
leg() { local n=$1; shift
  env "$@" timeout 400 python tools/bench_reference_host.py --env PegInsertionSide-v1 --envs 4096 --steps 100 --accelerate graph > $O/dropin_$n.json 2> $O/dropin_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/dropin_$n.json") if l.startswith("{")][-1]); print("dropin $n: %.3f M  %.3f ms  %s" % (d["value"]/1e6, d["ms_per_step"], d["accelerate"]))
except Exception as e: print("dropin $n failed", e); print(open("$O/dropin_$n.err").read()[-800:])
PY
}
( run new_1 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; run old_1 MSK_LIB=maniskill_amd/csrc/libmsk_r06ev.so; run new_2 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; run old_2 MSK_LIB=maniskill_amd/csrc/libmsk_r06ev.so
  leg new MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; leg old MSK_LIB=maniskill_amd/csrc/libmsk_r06ev.so ) | tee $O/ab_peg_observe_kin.log
