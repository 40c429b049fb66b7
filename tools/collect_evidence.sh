#!/bin/bash
# copies what tools/gpu_calls/gpu_r06_final.sh pulled back (gpurun_out/<dir>) into profiles/ under the round's names:   bash tools/collect_evidence.sh r06_final2 r06
set -e
F=gpurun_out/$1; P=profiles; R=$2
cp $F/gpu_tests_driver_form.log $P/${R}_gpu_tests_final_driver_form.log; cp $F/smoke.log $P/${R}_smoke_final.log
cp $F/bench_n1_driver_form.json $P/${R}_bench_n1_driver_form.json; cp $F/bench_n1_default_1000.json $P/${R}_bench_n1_default.json; cp $F/bench_n1_512envs.json $P/${R}_bench_n1_512envs.json
cp $F/bench_pusht_camera_4096.json $P/${R}_bench_pusht_camera_4096.json; cp $F/bench_peg_insertion_4096.json $P/${R}_bench_peg_insertion_4096.json
cp $F/prof_graph/runc/*kernel_stats.csv $P/${R}_kernel_stats_bench_4096_graph.csv; cp $F/prof_late/runc/*kernel_stats.csv $P/${R}_kernel_stats_bench_4096_1000steps.csv
cp $F/prof_pusht_cam/runc/*kernel_stats.csv $P/${R}_kernel_stats_pusht_camera_4096.csv; cp $F/prof_peg/runc/*kernel_stats.csv $P/${R}_kernel_stats_peg_insertion_4096.csv
cp $F/prof_vector/runc/*kernel_stats.csv $P/${R}_kernel_stats_vector_env_loop.csv; cp $F/prof_pusht_dropin/runc/*kernel_stats.csv $P/${R}_kernel_stats_pusht_camera_dropin_4096.csv
cp $F/pmc_counters_4096.json $P/${R}_pmc_counters_4096.json; cp $F/pmc_counters_camera_4096.json $P/${R}_pmc_counters_camera_4096.json
cp $F/phase_probe_pickcube.log $P/${R}_phase_probe_pickcube.log; cp $F/mfma_ab.log $P/${R}_mfma_ab.log
grep -v "Warning\|gpu_init\|amdgpu.ids" $F/vector_probe.log | cut -c1-220 > $P/${R}_vector_probe.log; grep "bare env\|vector env" $F/soak_20000.log | cut -c1-220 > $P/${R}_soak_20000_steps_rate.log
ls $P | grep "^${R}_" | wc -l
