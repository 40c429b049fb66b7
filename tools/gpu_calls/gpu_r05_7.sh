#!/bin/bash
# round 5, GPU call: candidate E2 (msk_solve.h MSK_OWNER_LOCAL_ROWS: a block's rows carried forward in the owner lane) against the default, ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_7; mkdir -p $O
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
for rep in 1 2 3; do
  run physx_$rep MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
  run e2_$rep MSK_LIB=maniskill_amd/csrc/libmsk_e2.so
done
for n in physx e2; do
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so timeout 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --env PegInsertionSide-v1 > $O/ab_peg_$n.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/ab_peg_$n.json'));print('peg $n', round(d['value']), {k:round(v,1) for k,v in d['roofline']['kernel_us'].items()})"
done
MSK_LIB=maniskill_amd/csrc/libmsk_e2.so timeout 300 python tools/gpu_fuzz_parity.py 128 200 1 PickCube,Peg,StackCube 2>/dev/null
