#!/bin/bash
# round 6, call 20: k_narrowphase at two wavefronts per SIMD (amdgpu_waves_per_eu(2, 2): 256 VGPRs, no AGPRs, 183 scratch instructions instead of 42 + 45 AGPR moves) -- call 19's
# stamps showed its 1792 workgroups queueing for the 1024 slots of one wavefront per SIMD (PegInsertionSide: hull rows starting 17-90 us late, the launch 134 us against workgroups of at most 111)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_20; mkdir -p $O
cd $R
MSK_LIB=maniskill_amd/csrc/libmsk_np2.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_hull_heaps.py -m gpu -x -q > $O/pytest_parity_np2.log 2>&1; echo "pytest parity (np2) rc $?"; tail -3 $O/pytest_parity_np2.log
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-1000} --warmup ${WARM:-20} --no-cpu-baseline --no-extras ${EXTRA:-} > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, (d.get("camera") or {}).get("us_per_frame", ""))
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
N=MSK_LIB=maniskill_amd/csrc/libmsk_np2.so; L=MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
( run np2_1 $N; run head_1 $L; run np2_2 $N; run head_2 $L
  STEPS=20 WARM=5 run np2_20steps $N; STEPS=20 WARM=5 run head_20steps $L
  STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_np2 $N; STEPS=300 EXTRA="--env PegInsertionSide-v1" run peg_head $L
  STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_np2 $N; STEPS=200 EXTRA="--env PushT-v1 --obs-mode depth+segmentation" run pusht_head $L
  STEPS=300 EXTRA="--envs 512" run 512_np2 $N; STEPS=300 EXTRA="--envs 512" run 512_head $L
  STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_np2 $N; STEPS=300 EXTRA="--envs 16384 --env PegInsertionSide-v1" run peg16384_head $L
  STEPS=300 EXTRA="--envs 65536" run 65536_np2 $N; STEPS=300 EXTRA="--envs 65536" run 65536_head $L ) | tee $O/ab_narrowphase_two_waves_per_simd.log
