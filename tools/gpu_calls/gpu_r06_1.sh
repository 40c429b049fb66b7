#!/bin/bash
# round 6, call 1: (a) the new -m gpu nodes of the kernel-backed task plugins, (b) the drop-in legs of configs 2, 3, 4 at 4096 envs (eager plugin and graph), (c) the fixed
# chain microbenchmark, (d) A/B on this one box: the shipped library against the same sources built with -fno-slp-vectorize (the compiler packs pairs of the row
# update's fma into v_pk_fma_f32, which sits on the solver's dependent chain)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fused_step.py -m gpu -x -q -k "fused_task_kernels_as_one_hip_graph or torch_plugin" > $O/pytest_fused.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_fused.log
leg() { local n=$1; shift
  timeout 400 python tools/bench_reference_host.py "$@" > $O/dropin_$n.json 2> $O/dropin_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/dropin_$n.json") if l.startswith("{")][-1]); print("$n: %.3f M  %.3f ms  build %.1f s  %s" % (d["value"]/1e6, d["ms_per_step"], d["build_s"], d["accelerate"]))
except Exception as e: print("$n failed", e); print(open("$O/dropin_$n.err").read()[-1500:])
PY
}
leg pickcube_graph --envs 4096 --steps 200 --accelerate graph
leg pickcube_task --envs 4096 --steps 200 --accelerate task
leg peg_graph --env PegInsertionSide-v1 --envs 4096 --steps 100 --accelerate graph
leg pusht_cam_graph --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 100 --accelerate graph
leg pusht_state_graph --env PushT-v1 --envs 4096 --steps 100 --accelerate graph
hipcc --offload-arch=gfx950 -O3 -o $O/chain_microbench tools/chain_microbench.hip 2>/dev/null && $O/chain_microbench > $O/chain_microbench.log; cat $O/chain_microbench.log
run() { local n=$1; shift
  env "$@" timeout 200 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/ab_$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms  late %.3f M  early" % (d["value"]/1e6, d["ms_per_step"], d["step_late"]["value"]/1e6), {k: round(v,1) for k,v in r["kernel_us"].items()}, "late", {k: round(v,1) for k,v in (r.get("kernel_us_late") or {}).items()})
except Exception as e: print("$n failed", e)
PY
}
run physx_1 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
run noslp_1 MSK_LIB=maniskill_amd/csrc/libmsk_noslp.so
run physx_2 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
run noslp_2 MSK_LIB=maniskill_amd/csrc/libmsk_noslp.so
