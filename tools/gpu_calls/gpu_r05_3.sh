#!/bin/bash
# round 5, GPU call 3: the cabinet graph test with matched histories, the planes-only camera (parity + what it buys + phase cuts), the sharded drop-in bench
# leg, and where PegInsertionSide's solver time goes (class histogram, per-class phase cycles on the -DMSK_PROFILE_PHASES build)
#   gpurun --timeout 1500 -- 'bash tools/gpu_calls/gpu_r05_3.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_3; mkdir -p $O
cd $R
timeout 600 python -m pytest -q -m gpu -n 4 -p no:cacheprovider tests/test_fused_step.py tests/test_render.py tests/test_push_t.py tests/test_vector_env.py tests/test_env_api.py > $O/gpu_tests_subset.log 2>&1; tail -12 $O/gpu_tests_subset.log
timeout 200 python bench.py --env OpenCabinetDrawer-v1 --envs 1024 --steps 50 --warmup 5 > $O/bench_dropin_sharded_cabinet_1024.json 2> $O/bench_dropin_sharded_cabinet.err; tail -c 700 $O/bench_dropin_sharded_cabinet_1024.json; tail -3 $O/bench_dropin_sharded_cabinet.err
timeout 200 python bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 100 --no-cpu-baseline --no-extras > $O/bench_pusht_camera_4096.json 2> $O/bench_pusht.err
python - <<PY
import json
d=json.load(open("$O/bench_pusht_camera_4096.json")); print("PushT camera:", d["value"], d["ms_per_step"], d["camera"], d["roofline"]["kernel_us"])
PY
timeout 300 python tools/gpu_render_probe.py PushT > $O/render_probe_pusht.log 2>&1; cat $O/render_probe_pusht.log | grep "us per picture"
timeout 120 python tools/gpu_peg_probe.py > $O/peg_probe.log 2>&1; tail -4 $O/peg_probe.log
PROBE_ENV=Peg PROBE_STEPS=60 timeout 200 python tools/gpu_phase_probe.py > $O/peg_phase_probe.log 2>&1; grep -v Warning $O/peg_phase_probe.log | tail -16
