#!/bin/bash
# round 6, the round's evidence on ONE commit: the driver's own commands (pytest -m gpu -x -q, smoke, bench.py --gpus 1 --steps 20 --warmup 5), the default bench line,
# the bench forms, a same-box A/B against the library of the evidence run before this one (libmsk_base.so = 72d5aae's csrc: before this round's launch-position work: the 256-thread k_dynamics, the broadphase's constants once per wavefront, the narrowphase's hull rows dispatched first), the phase probes, the vector-env probe and soak,
# rocprofv3 kernel statistics and the PMC passes (summarised here: only the summaries travel back), the MFMA question's microbenchmark
#   gpurun --timeout 3000 -- 'bash tools/gpu_calls/gpu_r06_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${FINAL_DIR:-r06_final}; mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_driver_form.log 2>&1; tail -6 $O/gpu_tests_driver_form.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# the PMC passes first: the bench lines below quote their bytes (roofline.traffic) only when the committed summary is of THESE kernel sources, so the summary goes into profiles/ of this copy before they run
rm -rf $R/gpurun_out/pmc; timeout 500 bash tools/pmc_collect.sh > /dev/null 2>&1
python tools/pmc_summarise.py $R/gpurun_out/pmc $O/pmc_counters_4096.json "python bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras" r06-final > $O/pmc_summary.log 2>&1; tail -12 $O/pmc_summary.log
rm -rf $R/gpurun_out/pmc; timeout 500 bash tools/pmc_collect.sh --env PushT-v1 --obs-mode depth+segmentation > /dev/null 2>&1
python tools/pmc_summarise.py $R/gpurun_out/pmc $O/pmc_counters_camera_4096.json "python bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras --env PushT-v1 --obs-mode depth+segmentation" r06-final > $O/pmc_camera_summary.log 2>&1; tail -6 $O/pmc_camera_summary.log
rm -rf $R/gpurun_out/pmc
cp $O/pmc_counters_4096.json $O/pmc_counters_camera_4096.json $R/profiles/ 2>/dev/null; for f in pmc_counters_4096 pmc_counters_camera_4096; do mv $R/profiles/$f.json $R/profiles/r06_$f.json; done
MSK_BENCH_EXTRA_S=500 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench_n1_driver_form.err; tail -c 300 $O/bench_n1_driver_form.json; echo
timeout 300 python bench.py --no-cpu-baseline > $O/bench_n1_default_1000.json 2> $O/bench_n1_default.err; tail -c 200 $O/bench_n1_default_1000.json; echo
timeout 200 python bench.py --envs 512 --no-cpu-baseline --no-extras > $O/bench_n1_512envs.json 2>/dev/null
timeout 200 python bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 200 --no-cpu-baseline --no-extras > $O/bench_pusht_camera_4096.json 2>/dev/null
timeout 200 python bench.py --env PegInsertionSide-v1 --steps 300 --no-cpu-baseline --no-extras > $O/bench_peg_insertion_4096.json 2>/dev/null
run() { local n=$1; shift
  env "$@" timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_$n.json 2>$O/ab_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/ab_$n.json")); r=d["roofline"]; print("$n: %.3f M  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in r["kernel_us"].items()})
except Exception as e: print("$n failed", e); print(open("$O/ab_$n.err").read()[-800:])
PY
}
( run new_1 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
  run old_1 MSK_LIB=maniskill_amd/csrc/libmsk_base.so
  run new_2 MSK_LIB=maniskill_amd/csrc/libmsk_physx.so
  run old_2 MSK_LIB=maniskill_amd/csrc/libmsk_base.so ) | tee $O/ab_head_vs_previous_evidence_run.log
PROBE_STEPS=100 timeout 300 python tools/gpu_phase_probe.py > $O/phase_probe_pickcube.log 2>&1; tail -3 $O/phase_probe_pickcube.log | cut -c1-300
timeout 600 python tools/gpu_vector_probe.py 4096 300 > $O/vector_probe.log 2>&1; grep -v Warning $O/vector_probe.log | cut -c1-160 | sed -n 3,14p
timeout 600 python tools/gpu_soak_rate.py 20 4096 > $O/soak_20000.log 2>&1; grep "vector env\|bare" $O/soak_20000.log | cut -c1-80 | tail -22
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o $O/mfma_ab tools/mfma_ab.hip 2>/dev/null && $O/mfma_ab > $O/mfma_ab.log; hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -Wno-unused-value -o $O/mfma_ab_noslp tools/mfma_ab.hip 2>/dev/null && (echo "--- the VALU form built with -fno-slp-vectorize (the product's flags: no v_pk_fma_f32)"; $O/mfma_ab_noslp) >> $O/mfma_ab.log; cat $O/mfma_ab.log; rm -f $O/mfma_ab $O/mfma_ab_noslp
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_graph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_late -- python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_late.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pusht_cam -- python $R/bench.py --env PushT-v1 --obs-mode depth+segmentation --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_pusht_cam.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_peg -- python $R/bench.py --env PegInsertionSide-v1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_peg.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vector -- python $R/tools/gpu_soak_rate.py 2 4096 > $O/prof_vector.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pusht_dropin -- python $R/tools/bench_reference_host.py --env PushT-v1 --obs-mode depth+segmentation --envs 4096 --steps 50 --accelerate graph > $O/prof_pusht_dropin.log 2>&1
find $O -name '*kernel_trace.csv' -delete
cd $R
find $O -name "*kernel_stats.csv"
