#!/bin/bash
# round 6, call 18: the library as it will ship -- 256-thread k_dynamics by default at 4096 envs, the broadphase's template constants once per wavefront, a culled pair's contact
# count asked for ahead of the shape pass -- against call 17's library (without the prefetch, MSK_DYN_THREADS=256) and the third evidence run's (libmsk_base.so = 72d5aae's csrc)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_18; mkdir -p $O
cd $R
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, "gap/substep %.1f" % r.get("launch_gap_us_per_substep", -1), (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so; C=MSK_LIB=maniskill_amd/csrc/libmsk_call17.so; T=MSK_LIB=maniskill_amd/csrc/libmsk_base.so; W=MSK_DYN_THREADS=256
( run new_1 $N; run call17_1 $C $W; run base_1 $T; run new_2 $N; run call17_2 $C $W; run base_2 $T
  STEPS=20 WARM=5 run new_20steps $N; STEPS=20 WARM=5 run call17_20steps $C $W; STEPS=20 WARM=5 run base_20steps $T; STEPS=20 WARM=5 run new_20steps_b $N; STEPS=20 WARM=5 run base_20steps_b $T
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_new $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_base $T
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_new $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_base $T
  STEPS=300 EXTRA="--envs 512" run 512_new $N; STEPS=300 EXTRA="--envs 512" run 512_base $T
  STEPS=300 EXTRA="--envs 2816" run 2816_new $N; STEPS=300 EXTRA="--envs 2816" run 2816_base $T
  STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_new $N; STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_base $T
  STEPS=300 EXTRA="--envs 65536" run 65536_new $N; STEPS=300 EXTRA="--envs 65536" run 65536_base $T ) | tee $O/ab_shipping_library.log
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; grep "first wave\|k_dynamics phases" $O/phase_probe_pickcube.log | cut -c1-520
