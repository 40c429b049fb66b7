#!/bin/bash
# round 5: candidate -- the box-box list through the 16-lanes-per-pair manifold code (libmsk_bbg.so: -DMSK_BOXBOX_GROUP) with 4 / 6 / 8 / 12 box blocks per env group, against the default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_9; mkdir -p $O
cd $R
run() { local n=$1; shift
  env "$@" timeout 150 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/ab_$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms  late %.3f M  early" % (d["value"]/1e6, d["ms_per_step"], d["step_late"]["value"]/1e6), {k: round(v,1) for k,v in r["kernel_us"].items()}, "late", {k: round(v,1) for k,v in (r.get("kernel_us_late") or {}).items()})
except Exception as e: print("$n failed", e)
PY
}
run physx_1 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
for nb in 4 6 8 12; do run bbg_nbox$nb MSK_LIB=maniskill_amd/csrc/libmsk_bbg.so MSK_NP_NBOX=$nb; done
run physx_2 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
MSK_LIB=maniskill_amd/csrc/libmsk_bbg.so MSK_NP_NBOX=6 timeout 300 python tools/gpu_fuzz_parity.py 128 200 1 PickCube,Peg,StackCube 2>/dev/null
