#!/bin/bash
# round 5, GPU call 4: the two-pass triangle set-up of k_render_splat against the build before it (one box), its phase cuts, the render parity tests; what the
# wide contact capacity costs on the headline and buys on the Allegro task; the 512-env point
#   gpurun --timeout 1200 -- 'bash tools/gpu_calls/gpu_r05_4.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_4; mkdir -p $O
cd $R
for rep in 1 2; do for n in prev physx; do
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so MSK_RENDER_CUT=0 timeout 100 python tools/gpu_render_probe.py PushT child 2>/dev/null | tail -1
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so MSK_RENDER_CUT=0 timeout 100 python tools/gpu_render_probe.py PickCube child 2>/dev/null | tail -1
done; done | tee $O/render_ab.log
timeout 300 python tools/gpu_render_probe.py PushT 2>/dev/null | grep "us per picture" | tee $O/render_probe_pusht.log
timeout 400 python -m pytest -q -m gpu -n 4 -p no:cacheprovider tests/test_render.py tests/test_push_t.py tests/test_trajectory.py > $O/gpu_tests_render.log 2>&1; tail -3 $O/gpu_tests_render.log
for cap in 0 1; do
  timeout 120 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras --contact-capacity $cap > $O/bench_capacity_$cap.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_capacity_$cap.json")); print("PickCube capacity $cap: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()})
PY
  MSK_CONTACT_CAPACITY=$cap timeout 200 python tools/bench_reference_host.py --env RotateSingleObjectInHandLevel1-v1 --envs 1024 --steps 50 > $O/allegro_capacity_$cap.log 2>&1; tail -1 $O/allegro_capacity_$cap.log | cut -c1-400
done
timeout 120 python bench.py --envs 512 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_n1_512envs.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$O/bench_n1_512envs.json")); print("512 envs: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()})
PY
