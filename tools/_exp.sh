mkdir -p gpurun_out/r02
for args in "-e FrankaPickCubeBenchmark-v1 -n=4096 -o=state --sim-freq=100 --control-freq=50" "-e FrankaPickCubeBenchmark-v1 -n=1024 -o=state --sim-freq=100 --control-freq=50" "-e FrankaPickCubeBenchmark-v1 -n=8192 -o=state --sim-freq=100 --control-freq=50" "-e FrankaMoveBenchmark-v1 -n=4096 -o=state --sim-freq=100 --control-freq=50" "-e PickCube-v1 -n=1024 -o=rgb --num-cams=1 --cam-width=128 --cam-height=128 --sim-freq=100 --control-freq=50" "-e CartpoleBalanceBenchmark-v1 -n=1024 -o=rgb --num-cams=1 --cam-width=128 --cam-height=128" "-e CartpoleBalanceBenchmark-v1 -n=1024 -o=rgb+depth --num-cams=1 --cam-width=128 --cam-height=128" "-e CartpoleBalanceBenchmark-v1 -n=4096 -o=state"; do
  echo "== python gpu_sim.py $args"
  python tools/bench_reference_harness.py $args 2>&1 | grep -v "Warning\|WARNING\|warn" | grep "steps/s\|Task ID\|sim_freq\|Error\|error" | cut -c1-200
done 2>&1 | tee gpurun_out/r02/reference_harness.log
