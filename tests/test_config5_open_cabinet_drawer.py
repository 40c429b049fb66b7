"""BASELINE.json config 5: the reference's own OpenCabinetDrawer-v1 (Fetch, 15 dofs, 29 links, velocity-controlled base + one of 25
cabinets per sub-scene, padded max_dof) on this backend through the sapien shim.  The PartNet-Mobility cabinets are a download;
tools/make_synthetic_partnet.py writes the substitute SURVEY.md §8(d) prescribes (1-4 prismatic drawers, 3-8 hulls of 16-64 vertices
per link, seeded by the model id) in the dataset's own file layout, so nothing of the task code changes."""
import json
import os
import subprocess
import sys

import pytest

import ref_harness

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
needs_ref = pytest.mark.skipif(ref_harness.find_reference() is None, reason="no ManiSkill checkout (reference) available")

_CPU = r'''
import os, subprocess, sys
sys.path.insert(0, %(here)r); sys.path.insert(0, %(root)r)
import ref_harness
ref = ref_harness.find_reference()
meta = os.path.join(ref, "mani_skill", "assets", "partnet_mobility", "meta")
subprocess.check_call([sys.executable, os.path.join(%(root)r, "tools", "make_synthetic_partnet.py"), "--out", %(assets)r, "--max-drawers", "4",
                       "--ids-from", os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from",
                       os.path.join(meta, "info_cabinet_door_train.json")], stdout=subprocess.DEVNULL)
os.environ["MS_ASSET_DIR"] = %(assets)r
gym = ref_harness.setup("oracle")
import torch
env = gym.make("OpenCabinetDrawer-v1", num_envs=12, obs_mode="state_dict")
obs, _ = env.reset(seed=0)
base = env.unwrapped
sd = base.get_state_dict()
md = base.cabinet.max_dof
assert sd["articulations"]["cabinet"].shape == (12, 13 + 2 * md) and sd["articulations"]["fetch"].shape == (12, 13 + 15 * 2)   # tests/test_sim_state.py:73-103
assert len(base.scene.px._groups) > 1 and 1 <= md <= 4
assert obs["agent"]["qpos"].shape == (12, 15)
for _ in range(4):
    obs, rew, term, trunc, info = env.step(env.action_space.sample())
assert torch.isfinite(rew).all() and torch.isfinite(base.get_state()).all()
# state round trip across structurally different sub-scenes
s0 = base.get_state().clone()
for _ in range(3):
    env.step(env.action_space.sample())
base.set_state(s0)
assert torch.allclose(base.get_state(), s0, atol=1e-5)
# pulling a drawer: a force on the handle link's joint opens it and the task's own evaluate sees it
q = base.cabinet.get_qpos().clone()
lim = base.cabinet.get_qlimits()
q[:] = lim[..., 1] * 0.9
base.cabinet.set_qpos(q)
base.scene._gpu_apply_all(); base.scene.px.gpu_update_articulation_kinematics(); base.scene._gpu_fetch_all()
assert base.evaluate()["open_enough"].all()
print("CONFIG5_OK", md, len(base.scene.px._groups))
'''


@needs_ref
def test_reference_open_cabinet_drawer_on_cpu_checker(built, tmp_path):
    code = _CPU % dict(here=HERE, root=ROOT, assets=str(tmp_path / "assets"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "CONFIG5_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@needs_ref
@pytest.mark.gpu
def test_reference_open_cabinet_drawer_hip_matches_oracle(built):
    """32 sub-scenes (17 structural groups), 50 control steps of random actions from a common initial state: fp32 states within 1e-4
    relative (tools/gpu_cabinet_probe.py parity)."""
    env = dict(os.environ, MSK_CABINET_DRAWERS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_cabinet_probe.py"), "parity"], capture_output=True, text=True, timeout=1800, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(line[-1])
    assert res["groups"] > 1 and res["max_rel_err"] < 1e-4, res


@needs_ref
@pytest.mark.gpu
def test_reference_open_cabinet_drawer_at_scale_hip_matches_oracle(built):
    """Config 5 at scale: 1024 sub-scenes (all 25 structural groups of the synthetic cabinet set, shared launches), 100 control steps of random
    actions, HIP against the oracle on ALL 1024.  Physics only: the oracle side is fed the drive targets the HIP side's controllers wrote --
    the reference's controller code runs in torch on the GPU there and on the CPU here, and their sin / cos differ by an ulp (9 of 96 envs
    differ by 6e-7 after ONE step when both sides run their own controllers; with the targets handed over every state of every step is
    bit-equal, and no group raises a solver scheduling flag)."""
    env = dict(os.environ, MSK_CABINET_DRAWERS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_cabinet_probe.py"), "parity_physics", "1024", "100"], capture_output=True,
                       text=True, timeout=3000, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(line[-1])
    assert res["groups"] >= 20 and res["max_rel_err"] < 1e-4 and res["bit_equal_steps"] >= 90 and all(f & 6 == 0 for f in res["flags"]), res
