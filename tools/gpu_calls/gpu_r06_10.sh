#!/bin/bash
# round 6, call 10: the kernels are the evidence run's again (the side-stream candidate of call 9 is reverted: csrc_digest 1adba498ecf94226); the whole -m gpu suite on the final
# host-side Python (ring rebuilds on a worker, the capture path's rewrites, attributes adopted before a replay) and the driver-form bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_10; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_driver_form.log 2>&1; tail -6 $O/gpu_tests_driver_form.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1_driver_form.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$O/bench_n1_driver_form.json"))
print("value %.3f M (%.3f ms)  value_1000 %.3f M  late %.3f M  vector_env_steady %.3f M (%.3f ms)" % (d["value"]/1e6, d["ms_per_step"], d.get("value_1000",0)/1e6, d["step_late"]["value"]/1e6, d["vector_env_steady"]["value"]/1e6, d["vector_env_steady"]["ms_per_step"]))
r=d["roofline"]; print("roofline frac %.4f substep %.4f measured %s traffic %s" % (r["frac"], r["substep"], r["measured_hbm_frac"], r["traffic"]))
for k in ("step_reset","dropin","dropin_fused_graph","config3_pusht_camera_4096_dropin","config4_peg_insertion_side_4096_dropin","config5_open_cabinet_drawer_1024","config3_pusht_camera_4096","config4_peg_insertion_side_4096"):
    print(k, json.dumps(d.get(k))[:200])
PY
