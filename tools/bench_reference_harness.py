#!/usr/bin/env python3
"""The REFERENCE'S OWN benchmark harness (mani_skill/examples/benchmarking/gpu_sim.py: SURVEY.md §8(d), the protocol behind the published
numbers: 1000 random-action steps after a warm-up step, its Profiler's "steps/s") run unmodified on this backend through the sapien shim.
Needs a reference build (a checkout, or oracle/_ref/maniskill).  Arguments are the harness's own:

    python tools/bench_reference_harness.py -e PickCube-v1 -n 4096 -o state [--control-freq 50 --sim-freq 100]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness  # noqa: E402

if ref_harness.find_reference() is None:
    print("no reference build present")
    sys.exit(0)
ref_harness.setup("hip")
import tyro  # noqa: E402
from mani_skill.examples.benchmarking import gpu_sim  # noqa: E402

gpu_sim.main(tyro.cli(gpu_sim.Args))
