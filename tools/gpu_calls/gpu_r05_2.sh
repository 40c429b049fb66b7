#!/bin/bash
# round 5, GPU call 2: the five tests call 1 failed (fixed on the host side), the new paths of bench.py, and the A/B of kernel builds on ONE box:
# HEAD, HEAD with MSK_AREG_ROWBITS (libmsk_e1.so), and round 4's mid-round commit c73dab6 (the 13-20 % question of the round-4 review)
#   gpurun --timeout 1500 -- 'bash tools/gpu_calls/gpu_r05_2.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_2; mkdir -p $O
cd $R
timeout 600 python -m pytest -q -m gpu -n 4 -p no:cacheprovider tests/test_fused_step.py tests/test_wide_solver.py tests/test_many_coordinates.py tests/test_gpu_parity.py > $O/gpu_tests_subset.log 2>&1; tail -25 $O/gpu_tests_subset.log
for rep in 1 2; do for n in physx e1 c73dab6; do
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so timeout 120 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_${n}_$rep.json 2> $O/ab_${n}_$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_${n}_$rep.json")); r=d["roofline"]; print("$n rep $rep: %.3f M  %.3f ms  early" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()}, "late", {k: round(v,1) for k,v in (r.get("kernel_us_late") or {}).items()})
except Exception as e: print("$n rep $rep failed", e)
PY
done; done
for n in physx e1; do
  MSK_LIB=maniskill_amd/csrc/libmsk_$n.so timeout 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --env PegInsertionSide-v1 > $O/ab_peg_${n}.json 2> $O/ab_peg_${n}.err; tail -c 300 $O/ab_peg_${n}.json | head -c 300; echo
done
timeout 200 python bench.py --env OpenCabinetDrawer-v1 --envs 1024 --steps 50 --warmup 5 > $O/bench_dropin_sharded_cabinet_1024.json 2> $O/bench_dropin_sharded_cabinet.err; tail -c 600 $O/bench_dropin_sharded_cabinet_1024.json; tail -3 $O/bench_dropin_sharded_cabinet.err
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1.err; python - <<PY
import json
d=json.load(open("$O/bench_n1_driver_form.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"].get("kernel_us_late"))
for k in ("step_late","step_reset","dropin","dropin_fused_graph","config5_open_cabinet_drawer_1024","config3_pusht_camera_1024_dropin","config4_peg_insertion_side_4096_dropin","config3_pusht_camera_4096","config4_peg_insertion_side_4096"):
    print(k, json.dumps(d.get(k))[:700])
PY
