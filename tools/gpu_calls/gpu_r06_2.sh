#!/bin/bash
# round 6, call 2: device-side resets on hardware (parity nodes, soak behind the vector env against host-side resets), the default bench line with its new legs,
# a kernel trace of the PushT camera drop-in leg (3.9 ms per step in call 1 against 2.0 ms on the fused host: where?), the PickCube phase probe, the microbenchmark again
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_device_reset.py -m gpu -x -q > $O/pytest_device_reset.log 2>&1; echo "pytest device_reset rc $?"; tail -4 $O/pytest_device_reset.log
timeout 300 python tools/gpu_soak_rate.py 6 4096 > $O/soak_device_resets.log 2>&1; grep -v Warning $O/soak_device_resets.log | cut -c1-140
SOAK_HOST_RESETS=1 timeout 300 python tools/gpu_soak_rate.py 3 4096 > $O/soak_host_resets.log 2>&1; grep "vector env" $O/soak_host_resets.log | cut -c1-140
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("value %.3f M (%.3f ms)  value_1000 %.3f M  late %.3f M  vector_env_steady %.3f M (%.3f ms, %d steps with a reset)" % (d["value"]/1e6, d["ms_per_step"], d.get("value_1000",0)/1e6, d["step_late"]["value"]/1e6, d["vector_env_steady"]["value"]/1e6, d["vector_env_steady"]["ms_per_step"], d["vector_env_steady"]["steps_with_a_reset"]))
print("roofline substep %.4f measured %s kernel_us %s" % (d["roofline"]["substep"], d["roofline"]["measured_hbm_frac"], {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()}))
for k in ("step_reset","dropin","dropin_fused_graph","config3_pusht_camera_4096_dropin","config4_peg_insertion_side_4096_dropin","config5_open_cabinet_drawer_1024","config3_pusht_camera_4096","config4_peg_insertion_side_4096","cpu_baseline"):
    v=d.get(k); print(k, json.dumps(v)[:300])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pusht_dropin -- python $R/tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 50 --accelerate graph > $O/prof_pusht_dropin.log 2>&1
cd $R
f=$(ls $O/prof_pusht_dropin/*/*kernel_stats.csv 2>/dev/null | head -1); echo "kernel stats: $f"; head -14 "$f" | cut -c1-200
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; tail -12 $O/phase_probe_pickcube.log | cut -c1-400
hipcc --offload-arch=gfx950 -O3 -o $O/chain_microbench tools/chain_microbench.hip 2>/dev/null && $O/chain_microbench > $O/chain_microbench.log; head -8 $O/chain_microbench.log | cut -c1-210
