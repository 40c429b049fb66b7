"""bench.py's own rank bootstrap (round-3 review: `python bench.py --gpus N` without a launcher around it must start its N ranks itself).
The bootstrap is driven here without a GPU: `--bootstrap-selftest` stops after the rendezvous (gloo), one all-reduce and a barrier -- no
physics runs, the product path has no CPU fallback."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(OMP_NUM_THREADS="1", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)


def test_plain_command_spawns_its_two_ranks_and_rank_zero_prints_one_line():
    r = _run(["--gpus", "2", "--bootstrap-selftest"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"bootstrap": "ok", "world": 2, "sum": 3.0}


def test_under_a_launcher_the_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--bootstrap-selftest"], extra_env=dict(WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_one_rank_needs_no_rendezvous():
    r = _run(["--gpus", "1", "--bootstrap-selftest"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["world"] == 1


HOOK = os.path.join(ROOT, "tests", "ref_bench_hook.py")


def _json_line(r):
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads(lines[0])


def test_weak_scaling_form_runs_the_fused_host_on_two_ranks(built):
    """`bench.py --gpus 2 --scaling weak --envs 6`: 6 envs PER rank; bench's own bootstrap starts the ranks, bench's own timed_rollout steps a fused-host
    shard per rank (on the CPU checker, handed in by the self-test's hook), one pipelined gather per step, and the gathered observation has all 12 rows"""
    r = _run(["--gpus", "2", "--scaling", "weak", "--envs", "6", "--steps", "3", "--warmup", "1", "--bootstrap-selftest"],
             extra_env=dict(MSK_BENCH_SELFTEST_HOOK=HOOK, MSK_BENCH_HOOK_CASE="fused_weak"))
    d = _json_line(r)
    assert d["world"] == 2 and d["total_envs"] == 12 and d["envs_per_rank"] == 6 and d["scaling"] == "weak" and d["gathered_rows"] == 12 and d["seconds"] > 0, d


def test_sharded_drop_in_legs_of_configs_four_and_five_run_on_two_ranks(built, tmp_path):
    """BASELINE config 4's task (PegInsertionSide-v1) and config 5's (OpenCabinetDrawer-v1) over the drop-in path, sharded over two ranks started by bench.py
    itself: bench.dropin_sharded_run times the reference's own env per rank (fused_step at the task level: no HIP graphs on the checker) and rank 0 prints
    the contract's line"""
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_harness
    if ref_harness.find_reference() is None:
        pytest.skip("no reference build")
    for env_id, envs in (("PegInsertionSide-v1", "4"), ("OpenCabinetDrawer-v1", "4")):
        r = _run(["--gpus", "2", "--env", env_id, "--envs", envs, "--steps", "2", "--warmup", "1", "--accelerate", "task", "--bootstrap-selftest"],
                 extra_env=dict(MSK_BENCH_SELFTEST_HOOK=HOOK, MSK_BENCH_HOOK_CASE="dropin", MS_ASSET_DIR=str(tmp_path / "assets")))
        d = _json_line(r)
        assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == int(envs) // 2 and d["unit"] == "env-steps/s" and d["value"] > 0 and env_id in d["metric"], d
        assert d["scaling"] == "strong" and "drop-in" in d["path"], d
