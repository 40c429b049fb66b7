#!/bin/bash
# round 6, call 27: k_render_splat's segments taken from a counter in LDS instead of dealt (wavefront w owned segments w, w + 4, ...: every other tile row of ONE half of the picture) --
# picture parity nodes, then config 3 (PushT, 128 x 128 depth + segmentation, fused host and over the reference API) against the library before (libmsk_prev.so = 87aec42's kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_27; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_render.py tests/test_gpu_parity.py tests/test_step_graph.py -m gpu -x -q > $O/pytest_render.log 2>&1; echo "pytest render parity rc $?"; tail -3 $O/pytest_render.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-200} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_prev.so
( EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new_1 $N; EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_prev_1 $L
  EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new_2 $N; EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_prev_2 $L
  EXTRA="--env PushT-v1 --obs-mode rgbd" run pusht_rgbd_new $N; EXTRA="--env PushT-v1 --obs-mode rgbd" run pusht_rgbd_prev $L
  EXTRA="--obs-mode rgbd" run pickcube_rgbd_new $N; EXTRA="--obs-mode rgbd" run pickcube_rgbd_prev $L
  EXTRA="--env PushT-v1 --obs-mode depth+segmentation --envs 16384" STEPS=50 run pusht16384_new $N; EXTRA="--env PushT-v1 --obs-mode depth+segmentation --envs 16384" STEPS=50 run pusht16384_prev $L ) | tee $O/ab_render_segments_taken.log
for l in libmsk_physx.so libmsk_prev.so; do MSK_LIB=maniskill_amd/csrc/$l timeout 600 python tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 50 --accelerate graph 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l over the reference API: %.3f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; done | tee -a $O/ab_render_segments_taken.log
