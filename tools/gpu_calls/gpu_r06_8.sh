#!/bin/bash
# round 6, call 8 (host-side Python only, the kernels are those of the final call): the ring of prepared episodes rebuilt by a worker after a seeded reset (the default line's
# step_reset leg showed 2.9 s inside every seeded reset), the capture path's new rewrites on hardware: the parity nodes of the files touched, the default bench line, the soak
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_8; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_device_reset.py tests/test_vector_env.py tests/test_fused_step.py tests/test_reference_conformance.py -m gpu -x -q > $O/pytest_host_side.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_host_side.log
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1_driver_form.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$O/bench_n1_driver_form.json"))
print("value %.3f M (%.3f ms)  value_1000 %.3f M  late %.3f M  vector_env_steady %.3f M (%.3f ms)" % (d["value"]/1e6, d["ms_per_step"], d.get("value_1000",0)/1e6, d["step_late"]["value"]/1e6, d["vector_env_steady"]["value"]/1e6, d["vector_env_steady"]["ms_per_step"]))
r=d["roofline"]; print("roofline frac %.4f substep %.4f measured %s traffic %s" % (r["frac"], r["substep"], r["measured_hbm_frac"], r["traffic"]))
for k in ("step_reset","vector_env_steady","dropin","dropin_fused_graph","config3_pusht_camera_4096_dropin","config4_peg_insertion_side_4096_dropin","config5_open_cabinet_drawer_1024"):
    print(k, json.dumps(d.get(k))[:300])
PY
timeout 300 python tools/gpu_soak_rate.py 6 4096 > $O/soak.log 2>&1; grep "vector env\|bare" $O/soak.log | cut -c1-80
timeout 300 python tools/gpu_soak.py 3000 4096 > $O/soak_checks.log 2>&1; tail -4 $O/soak_checks.log | cut -c1-200
