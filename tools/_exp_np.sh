cp ab/new3.so maniskill_amd/csrc/libmsk_physx.so
for nb in 2 3 4; do MSK_NP_NBOX=$nb python tools/gpu_kernel_probe.py 4096 600 2>&1 | tail -1; done
MSK_NP_NBOX=2 MSK_NP_NHULL=2 python tools/gpu_kernel_probe.py 4096 600 2>&1 | tail -1
