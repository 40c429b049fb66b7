#!/bin/bash
# round 5: PickCube env-steps/s against the env count on ONE GPU (the per-GPU shares of the metric's 2-, 4- and 8-GPU points: 2048, 1024, 512; and beyond 4096)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_8; mkdir -p $O
cd $R
for n in 512 1024 2048 4096 8192 32768; do
  timeout 300 python bench.py --envs $n --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_pickcube_$n.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/bench_pickcube_$n.json")); r=d["roofline"]; print("PickCube %6d envs: %.3f M env-steps/s  %.3f ms/step" % ($n, d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()})
PY
done
