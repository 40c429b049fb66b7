"""PhysxRigidDynamicComponent.set_locked_motion_axes (mani_skill/utils/structs/base.py:340-354, 455-468 -> msk_set_locked_axes): a locked
world axis carries no velocity and answers a constraint with infinite mass.  Known answers on the CPU oracle, HIP == oracle on the GPU."""
import numpy as np
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig

H = 0.02   # half size of the cube


def _world(factory, n, lock, z=0.3, v0=(0, 0, 0), w0=(0, 0, 0), half=(H, H, H), q=(1, 0, 0, 0), friction=0.5):
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    m = 0.1
    b = tpl.add_actor("cube", N.BODY_DYNAMIC, p=(0, 0, z), q=q, mass=m, inertia6=tuple(m / 3 * np.array([half[1] ** 2 + half[2] ** 2, half[0] ** 2 + half[2] ** 2,
                                                                                                     half[0] ** 2 + half[1] ** 2])) + (0, 0, 0), angular_damping=0.0)
    tpl.add_shape(b, N.SHAPE_BOX, params=half, static_friction=friction, dynamic_friction=friction)
    tpl.set_locked_axes(b, lock)
    px = factory(tpl, n, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((n, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(n, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[:, b, :3] = torch.tensor([0.0, 0.0, z])
    rbd[:, b, 3:7] = torch.tensor(q, dtype=torch.float32)
    rbd[:, b, 7:10] = torch.tensor(v0, dtype=torch.float32)
    rbd[:, b, 10:13] = torch.tensor(w0, dtype=torch.float32)
    px.gpu_apply_all()
    return px, b, rbd


def _run(px, steps):
    for _ in range(steps):
        px.step()
    px.gpu_fetch_all()


def test_a_cube_with_its_vertical_axis_locked_does_not_fall_and_still_moves_sideways(oracle_factory):
    px, b, rbd = _world(oracle_factory, 1, [0, 0, 1, 0, 0, 0], z=0.3, v0=(0.1, 0, 0))
    _run(px, 100)
    assert abs(rbd[0, b, 2].item() - 0.3) < 1e-6 and abs(rbd[0, b, 9].item()) < 1e-7          # hangs in the air
    assert abs(rbd[0, b, 0].item() - 0.1) < 1e-4                                               # 1 s at 0.1 m/s, nothing to slow it


def test_locked_rotation_turns_an_off_centre_landing_into_a_flat_one(oracle_factory):
    """A cube tilted by 20 degrees about y dropped on the table: free, it falls onto a face (the tilt goes to zero or a quarter turn); with the
    three angular axes locked it comes to rest standing on its edge, still tilted, still without spin."""
    th = np.deg2rad(20.0)
    q = (np.cos(th / 2), 0.0, np.sin(th / 2), 0.0)
    zc = H * (np.cos(th) + np.sin(th)) + 0.01
    free = _world(oracle_factory, 1, [0] * 6, z=zc, q=q)
    lock = _world(oracle_factory, 1, [0, 0, 0, 1, 1, 1], z=zc, q=q)
    _run(free[0], 150); _run(lock[0], 150)
    qf, ql = free[2][0, free[1], 3:7].numpy(), lock[2][0, lock[1], 3:7].numpy()
    assert np.allclose(np.abs(ql), np.abs(np.asarray(q)), atol=1e-6) and lock[2][0, lock[1], 10:13].abs().max() < 1e-7
    tilt = 2 * np.arccos(min(1.0, abs(qf[0])))
    assert min(abs(tilt), abs(tilt - np.pi / 2)) < 0.02                                         # the free cube lies on a face
    assert abs(lock[2][0, lock[1], 2].item() - H * (np.cos(th) + np.sin(th))) < 1e-3            # the locked one stands on its edge
    assert lock[2][0, lock[1], 7:10].abs().max() < 5e-3


def test_a_push_against_a_locked_direction_meets_a_wall(oracle_factory):
    """Two cubes in a row on the table, the first one sliding into the second: if the second is locked along x it does not move and the
    first one stops at it."""
    tpl = SceneTemplate()
    sb.add_table_scene(tpl)
    m = 0.1
    ids = []
    for k, x in enumerate((-0.06, 0.0)):
        b = tpl.add_actor(f"c{k}", N.BODY_DYNAMIC, p=(x, 0, H), mass=m, inertia6=(m / 6 * (2 * H) ** 2,) * 3 + (0, 0, 0))
        tpl.add_shape(b, N.SHAPE_BOX, params=(H, H, H), static_friction=0.1, dynamic_friction=0.1)
        ids.append(b)
    tpl.set_locked_axes(ids[1], [1, 0, 0, 0, 0, 0])
    px = oracle_factory(tpl, 1, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(1, px.bodies_per_env, 13)
    rbd[:, tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[:, ids[0], 7] = 0.5
    px.gpu_apply_all()
    _run(px, 60)
    assert abs(rbd[0, ids[1], 0].item()) < 1e-6                                                 # the locked cube did not give way
    assert rbd[0, ids[0], 0].item() < -2 * H + 2e-3 and abs(rbd[0, ids[0], 7].item()) < 2e-2     # the other one stopped in front of it


@pytest.mark.gpu
def test_locked_axes_hip_equals_oracle(built, oracle_factory):
    from maniskill_amd.physx import PhysxGpuSystem
    th = np.deg2rad(20.0)
    q = (np.cos(th / 2), 0.0, np.sin(th / 2), 0.0)
    zc = H * (np.cos(th) + np.sin(th)) + 0.01
    for lock in ([0, 0, 0, 1, 1, 1], [1, 0, 0, 0, 1, 0], [0, 0, 1, 0, 0, 1]):
        a = _world(oracle_factory, 4, lock, z=zc, q=q, v0=(0.2, 0.1, 0), w0=(1.0, 2.0, 3.0))
        g = _world(lambda t, n, c: PhysxGpuSystem("cuda:0", t, n, c), 4, lock, z=zc, q=q, v0=(0.2, 0.1, 0), w0=(1.0, 2.0, 3.0))
        for _ in range(10):
            _run(a[0], 10); _run(g[0], 10)
            assert torch.equal(a[2], g[2].cpu()), lock


_SHIM = r'''
import sys
sys.path.insert(0, %(here)r); sys.path.insert(0, %(root)r)
import ref_harness
gym = ref_harness.setup("oracle")
import torch
from mani_skill.envs.tasks.tabletop.push_cube import PushCubeEnv
from mani_skill.utils.registration import register_env


@register_env("PushCubeNoSpin-v0", max_episode_steps=50)
class PushCubeNoSpin(PushCubeEnv):
    def _load_scene(self, options):
        super()._load_scene(options)
        self.obj.set_locked_motion_axes([False, False, False, True, True, True])   # utils/structs/base.py:340-354


env = gym.make("PushCubeNoSpin-v0", num_envs=4, render_backend="none")
env.reset(seed=0)
base = env.unwrapped
assert torch.as_tensor(base.obj.get_locked_motion_axes())[0].tolist() == [False, False, False, True, True, True]
q0, p0 = base.obj.pose.q.clone(), base.obj.pose.p.clone()
base.obj.set_linear_velocity(torch.tensor([[0.3, 0.0, 0.0]]).repeat(4, 1))        # a shove and a spin, written into the simulator's state
base.obj.set_angular_velocity(torch.tensor([[1.0, 2.0, 3.0]]).repeat(4, 1))
base.scene._gpu_apply_all()
for _ in range(10):
    env.step(torch.zeros(4, env.action_space.shape[-1]))
    assert base.obj.angular_velocity.abs().max() < 1e-6                          # the spin is gone after the first substep
assert torch.allclose(base.obj.pose.q, q0, atol=1e-6)                             # ... and never turned the cube
assert (base.obj.pose.p[:, 0] - p0[:, 0]).min() > 5e-3                            # the shove moved it (friction stops it after ~1.5 cm)
print("LOCKED_OK")
'''


def test_locked_axes_through_the_shim(built, tmp_path):
    """The reference's own Actor.set_locked_motion_axes on a task's cube, over the shim on the CPU checker: given a shove and a spin,
    the cube translates and does not turn."""
    import os
    import subprocess
    import sys
    import ref_harness
    if ref_harness.find_reference() is None:
        pytest.skip("no ManiSkill checkout (reference) available")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", _SHIM % dict(here=here, root=os.path.dirname(here))], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "LOCKED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
