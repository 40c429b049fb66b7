timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_reference_conformance.py 2>&1 | tail -4
python bench.py --steps 1000 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'])"
python bench.py --steps 1000 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'])"
