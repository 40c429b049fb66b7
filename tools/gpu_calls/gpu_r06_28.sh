#!/bin/bash
# round 6, call 28: the two register budgets of the narrowphase at 4096 envs, task by task, on the final row order (MSK_NP_W2 = 0 / 1): is there a template property that should pick the second one below 8192 envs?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_28; mkdir -p $O
cd $R
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
( for rep in 1 2; do
  EXTRA="--env PegInsertionSide-v1" run peg_w1_$rep MSK_NP_W2=0; EXTRA="--env PegInsertionSide-v1" run peg_w2_$rep MSK_NP_W2=1
  EXTRA="--env PushT-v1" run pusht_state_w1_$rep MSK_NP_W2=0; EXTRA="--env PushT-v1" run pusht_state_w2_$rep MSK_NP_W2=1
  STEPS=1000 run pickcube_w1_$rep MSK_NP_W2=0; STEPS=1000 run pickcube_w2_$rep MSK_NP_W2=1
  done
  EXTRA="--env PegInsertionSide-v1 --envs 6144" run peg6144_w1 MSK_NP_W2=0; EXTRA="--env PegInsertionSide-v1 --envs 6144" run peg6144_w2 MSK_NP_W2=1
  EXTRA="--env PegInsertionSide-v1 --envs 2048" run peg2048_w1 MSK_NP_W2=0; EXTRA="--env PegInsertionSide-v1 --envs 2048" run peg2048_w2 MSK_NP_W2=1 ) | tee $O/ab_narrowphase_budgets_at_4096.log
