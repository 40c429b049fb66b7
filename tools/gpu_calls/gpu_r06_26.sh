#!/bin/bash
# round 6, GPU call 26 (call 12 again on the FINAL kernels of the round: digest 5be5eec68e5d30c5): the named env counts of BASELINE configs 4 and 5 on ONE GPU (16384 PegInsertionSide envs, 8192 OpenCabinetDrawer envs), and what more envs per GPU
# buy on PickCube (the metric's 4096 envs leave every SIMD with one wavefront: the kernels are chain-bound, not throughput-bound)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_26; mkdir -p $O
cd $R
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); r=d.get("roofline",{}); print("$2: %.3f M env-steps/s  %.3f ms/step" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r.get("kernel_us",{}).items()}, d.get("config",{}).get("build_s",""))
except Exception as e: print("$2 failed", e, open("$1.err").read()[-600:] if __import__("os").path.exists("$1.err") else "")
PY
}
for n in 4096 16384 65536; do
  timeout 300 python bench.py --envs $n --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_pickcube_$n.json 2> $O/bench_pickcube_$n.json.err; show $O/bench_pickcube_$n.json "PickCube $n envs"
done
timeout 300 python bench.py --env PegInsertionSide-v1 --envs 16384 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_peg_16384.json 2> $O/bench_peg_16384.json.err; show $O/bench_peg_16384.json "PegInsertionSide 16384 envs (config 4's env count, one GPU)"
timeout 600 python bench.py --env OpenCabinetDrawer-v1 --envs 8192 --steps 30 --warmup 5 > $O/bench_cabinet_8192.json 2> $O/bench_cabinet_8192.json.err; show $O/bench_cabinet_8192.json "OpenCabinetDrawer 8192 envs (config 5's env count, one GPU, drop-in graph)"
timeout 300 python bench.py --env PushT-v1 --obs-mode depth+segmentation --envs 16384 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_pusht_16384.json 2> $O/bench_pusht_16384.json.err; show $O/bench_pusht_16384.json "PushT camera 16384 envs"
timeout 400 python tools/bench_reference_host.py --env PegInsertionSide-v1 --envs 16384 --steps 50 --accelerate graph > $O/dropin_peg_16384.json 2> $O/dropin_peg_16384.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/dropin_peg_16384.json") if l.startswith("{")][-1]); print("PegInsertionSide 16384 envs over the reference API (graph): %.3f M  %.3f ms  build %.1f s" % (d["value"]/1e6, d["ms_per_step"], d["build_s"]))
except Exception as e: print("dropin peg 16384 failed", e); print(open("$O/dropin_peg_16384.err").read()[-800:])
PY
