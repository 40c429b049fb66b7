#!/bin/bash
# round 6, call 3: (a) the ring refills of the device-side reset on a worker thread + side stream: parity nodes, the vector-env probe (where does a wrapper step go?), the soak;
# (b) the PushT camera drop-in leg with Color switched off and the smaller template (MSK_RENDER_TRI_BUDGET 2600 / 5200), phase cuts of a picture of the shim's template;
# (c) the -m gpu nodes of the files this round touched
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_3; mkdir -p $O
cd $R
nproc > $O/nproc.log
timeout 900 python -m pytest tests/test_device_reset.py tests/test_vector_env.py -m gpu -x -q > $O/pytest_device_reset.log 2>&1; echo "pytest device_reset+vector rc $?"; tail -3 $O/pytest_device_reset.log
timeout 600 python tools/gpu_vector_probe.py 4096 300 > $O/vector_probe.log 2>&1; grep -v Warning $O/vector_probe.log | cut -c1-200 | head -60
timeout 300 python tools/gpu_soak_rate.py 6 4096 > $O/soak_device_resets.log 2>&1; grep "vector env\|bare" $O/soak_device_resets.log | cut -c1-90
leg() { local n=$1; shift
  timeout 500 "$@" > $O/dropin_$n.json 2> $O/dropin_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/dropin_$n.json") if l.startswith("{")][-1]); print("$n: %.3f M  %.3f ms  build %.1f s  %s" % (d["value"]/1e6, d["ms_per_step"], d["build_s"], d["accelerate"]))
except Exception as e: print("$n failed", e); print(open("$O/dropin_$n.err").read()[-1500:])
PY
}
leg pusht_cam_2600 python tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 100 --accelerate graph
leg pusht_cam_5200 env MSK_RENDER_TRI_BUDGET=5200 python tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 100 --accelerate graph
leg pusht_cam_1300 env MSK_RENDER_TRI_BUDGET=1300 python tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 100 --accelerate graph
timeout 900 python tools/gpu_render_probe_shim.py PushT-v1 4096 > $O/render_probe_shim_2600.log 2>&1; cat $O/render_probe_shim_2600.log | cut -c1-220
MSK_RENDER_TRI_BUDGET=5200 timeout 900 python tools/gpu_render_probe_shim.py PushT-v1 4096 > $O/render_probe_shim_5200.log 2>&1; tail -4 $O/render_probe_shim_5200.log | cut -c1-220
timeout 1500 python -m pytest tests/test_fused_step.py tests/test_render.py -m gpu -x -q > $O/pytest_fused_render.log 2>&1; echo "pytest fused+render rc $?"; tail -3 $O/pytest_fused_render.log
