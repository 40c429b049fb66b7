#!/bin/bash
# round 6, call 6: the whole -m gpu suite on the sources of call 5 (forward pass, DPP factorisation, wavefront-per-env reset), a 20 000-step soak behind the vector env
# (device-side resets: is the rate flat?), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_6; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -n 4 > $O/pytest_gpu_all.log 2>&1; echo "pytest -m gpu rc $?"; tail -5 $O/pytest_gpu_all.log
timeout 600 python tools/gpu_soak_rate.py 20 4096 > $O/soak_20000.log 2>&1; grep "vector env\|bare" $O/soak_20000.log | cut -c1-90
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("value %.3f M (%.3f ms)  value_1000 %.3f M  late %.3f M  vector_env_steady %.3f M (%.3f ms)" % (d["value"]/1e6, d["ms_per_step"], d.get("value_1000",0)/1e6, d["step_late"]["value"]/1e6, d["vector_env_steady"]["value"]/1e6, d["vector_env_steady"]["ms_per_step"]))
print("roofline", {k: (round(v,4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k in ("frac","substep","measured_hbm_frac")}, {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()})
for k in ("dropin","dropin_fused_graph","config3_pusht_camera_4096_dropin","config4_peg_insertion_side_4096_dropin","config5_open_cabinet_drawer_1024","config3_pusht_camera_4096","config4_peg_insertion_side_4096"):
    v=d.get(k); print(k, json.dumps(v)[:260])
PY
