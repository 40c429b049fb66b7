timeout 600 python -m pytest tests/test_bound_buffers.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python tools/gpu_cabinet_probe.py parity 2>&1 | tail -2
python tools/gpu_cabinet_probe.py bench 1024 4096 2>&1 | tail -1
MSK_BATCH_MERGED=0 python tools/gpu_cabinet_probe.py bench 1024 2>&1 | tail -1
