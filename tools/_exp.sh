timeout 900 python -m pytest tests/test_floating_base.py tests/test_gpu_parity.py tests/test_bound_buffers.py -x -q -m gpu 2>&1 | tail -5
cd tests && ZOO_ENVS=8 python ref_env_zoo.py hip 30 MS-AntWalk-v1 MS-AntRun-v1 MS-HumanoidStand-v1 MS-HumanoidWalk-v1 MS-HumanoidRun-v1 PickCube-v1 2>&1 | grep "^ZOO"
