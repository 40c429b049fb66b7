#!/bin/bash
# round 6, call 21: the two register budgets of the narrowphase in one library (k_narrowphase / k_narrowphase_w2, chosen by env count: >= 8192 -> two wavefronts per SIMD):
# parity nodes with the second form forced (MSK_NP_W2=1), then the default choice against the one-per-SIMD form forced (MSK_NP_W2=0 = the fourth evidence run's behaviour)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_21; mkdir -p $O
cd $R
MSK_NP_W2=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_hull_heaps.py tests/test_wide_solver.py tests/test_contact_trimming.py -m gpu -x -q > $O/pytest_parity_w2.log 2>&1; echo "pytest parity (w2 forced) rc $?"; tail -3 $O/pytest_parity_w2.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_w1.so
( run new_1 $N; run w1_1 $N MSK_NP_W2=0; run new_2 $N; run w1_2 $N MSK_NP_W2=0
  STEPS=20 WARM=5 run new_20steps $N; STEPS=20 WARM=5 run w1_20steps $N MSK_NP_W2=0
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_new $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_w2 $N MSK_NP_W2=1; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_w1 $N MSK_NP_W2=0
  STEPS=300 EXTRA="--envs 8192" run 8192_new $N; STEPS=300 EXTRA="--envs 8192" run 8192_w1 $N MSK_NP_W2=0
  STEPS=300 EXTRA="--envs 6144" run 6144_w2 $N MSK_NP_W2=1; STEPS=300 EXTRA="--envs 6144" run 6144_w1 $N
  STEPS=300 EXTRA="--envs 16384" run 16384_new $N; STEPS=300 EXTRA="--envs 16384" run 16384_w1 $N MSK_NP_W2=0
  STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_new $N; STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_w1 $N MSK_NP_W2=0
  STEPS=100 EXTRA="--envs 16384 --env PushT-v1 --obs-mode depth+segmentation" run pusht16384_new $N; STEPS=100 EXTRA="--envs 16384 --env PushT-v1 --obs-mode depth+segmentation" run pusht16384_w1 $N MSK_NP_W2=0
  STEPS=300 EXTRA="--envs 65536" run 65536_new $N; STEPS=300 EXTRA="--envs 65536" run 65536_w1 $N MSK_NP_W2=0 ) | tee $O/ab_narrowphase_by_env_count.log
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; grep "k_narrowphase:" $O/phase_probe_pickcube.log | cut -c1-420
