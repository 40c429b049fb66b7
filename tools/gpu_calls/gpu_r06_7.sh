#!/bin/bash
# round 6, call 7: the vector wrapper's book-keeping as one launch (msk_episode_book_step) on hardware: parity nodes, the probe and the soak again, a kernel trace of the
# wrapper's loop (what is left behind the step graph?), the default bench line (its vector_env_steady leg died in call 6: inference-mode tensors in the refill thread)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_7; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_vector_env.py tests/test_device_reset.py -m gpu -x -q > $O/pytest_vector.log 2>&1; echo "pytest vector rc $?"; tail -3 $O/pytest_vector.log
timeout 600 python tools/gpu_vector_probe.py 4096 300 > $O/vector_probe.log 2>&1; grep -v Warning $O/vector_probe.log | cut -c1-200 | sed -n 3,14p
timeout 300 python tools/gpu_soak_rate.py 6 4096 > $O/soak_device_resets.log 2>&1; grep "vector env\|bare" $O/soak_device_resets.log | cut -c1-90
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vector -- python $R/tools/gpu_soak_rate.py 2 4096 > $O/prof_vector.log 2>&1
cd $R
f=$(ls $O/prof_vector/*/*kernel_stats.csv 2>/dev/null | head -1); echo "kernel stats: $f"; head -30 "$f" | cut -c1-180
find $O -name '*kernel_trace.csv' -size +8M -delete
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("value %.3f M (%.3f ms)  value_1000 %.3f M  late %.3f M  vector_env_steady %.3f M (%.3f ms)" % (d["value"]/1e6, d["ms_per_step"], d.get("value_1000",0)/1e6, d["step_late"]["value"]/1e6, d["vector_env_steady"]["value"]/1e6, d["vector_env_steady"]["ms_per_step"]))
print("roofline", {k: (round(v,4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k in ("frac","substep","measured_hbm_frac")}, {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()})
for k in ("dropin","dropin_fused_graph","config3_pusht_camera_4096_dropin","config4_peg_insertion_side_4096_dropin","config5_open_cabinet_drawer_1024","config3_pusht_camera_4096","config4_peg_insertion_side_4096"):
    v=d.get(k); print(k, json.dumps(v)[:260])
PY
