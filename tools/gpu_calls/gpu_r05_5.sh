#!/bin/bash
# round 5, GPU call 5: scheduling-only candidates on ONE box -- wave priorities for the longest chains (libmsk_prio.so: -DMSK_SETPRIO=3) and a re-check of the
# launch-shape tuning aids on the final kernels; bench.py --steps 1000 (early + late kernel_us)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_5; mkdir -p $O
cd $R
run() { # name, env assignments...
  local n=$1; shift
  env "$@" timeout 150 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/ab_$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms  late %.3f M  early" % (d["value"]/1e6, d["ms_per_step"], d["step_late"]["value"]/1e6), {k: round(v,1) for k,v in r["kernel_us"].items()}, "late", {k: round(v,1) for k,v in (r.get("kernel_us_late") or {}).items()})
except Exception as e: print("$n failed", e)
PY
}
for rep in 1 2; do
  run physx_$rep MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
  run prio_$rep MSK_LIB=maniskill_amd/csrc/libmsk_prio.so
done
run dyn128 MSK_DYN_THREADS=128
run dyn64 MSK_DYN_THREADS=64
run nhull8 MSK_NP_NHULL=8
run nhull2 MSK_NP_NHULL=2
run nbox3 MSK_NP_NBOX=3
for n in physx prio; do
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so timeout 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --env PegInsertionSide-v1 > $O/ab_peg_$n.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/ab_peg_$n.json'));print('peg $n', round(d['value']), {k:round(v,1) for k,v in d['roofline']['kernel_us'].items()})"
done
