cd tests
python ref_run_node.py hip "tests/test_wrappers.py::test_recordepisode_wrapper_gpu[env_id=PickCube-v1,obs_mode=state]" "tests/test_wrappers.py::test_recordepisode_wrapper[env_id=StackCube-v1,obs_mode=rgb]" 2>&1 | grep -v "Warning\|WARNING\|warn" | tail -4 | cut -c1-250
cd ..
timeout 900 python -m pytest tests/test_render.py tests/test_push_t.py -x -q -m gpu 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
