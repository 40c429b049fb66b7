mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r02/gpu_tests_final.log
cat gpurun_out/r02/gpu_tests_final.log
