#!/bin/bash
# round 6, call 5: k_dynamics with the CRBA rows from the path table, the factorisation and substitutions on DPP row broadcasts (16-row form), the path walks in chunks; k_csolve without the owner selects on the friction bounds: parity, A/B against 18e0277 on one box, phase probe
# the library of the commit before (libmsk_r06c3.so = 18e0277's csrc), the phase probe, the vector-env probe and soak
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_5; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twins.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc $?"; tail -4 $O/pytest_parity.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()})
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
run new_1 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
run old_1 MSK_LIB=maniskill_amd/csrc/libmsk_r06c3.so
run new_2 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
run old_2 MSK_LIB=maniskill_amd/csrc/libmsk_r06c3.so
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; tail -3 $O/phase_probe_pickcube.log | cut -c1-400
for t in PegInsertionSide-v1; do
  timeout 300 python bench.py --env $t --no-extras --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$t.json 2>$O/bench_$t.err; python -c "
import json; d=json.load(open('$O/bench_$t.json')); print('$t', round(d['value']/1e6,3), 'M', {k: round(v,1) for k,v in d['roofline']['kernel_us'].items()})" 2>&1 | tail -1
done
