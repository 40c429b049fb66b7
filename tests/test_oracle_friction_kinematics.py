"""Sliding, spinning and the slide-to-roll transition as known answers for the CPU oracle's friction rows (two tangential rows per
contact point, cone mu * normal impulse): a cube sliding to a stop, a cube spinning on its four corner contacts, a ball that is pushed
off sliding and ends up rolling at 5/7 of its speed."""
import numpy as np
import torch

from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig

G, MU, H = 9.81, 0.3, 0.02


def _world(factory, ball=False):
    tpl = SceneTemplate()
    sb.add_table_scene(tpl, material=(MU, MU, 0.0))
    if ball:
        m = 1000.0 * 4.0 / 3.0 * np.pi * H ** 3
        I = 0.4 * m * H * H
        a = tpl.add_actor("a", N.BODY_DYNAMIC, p=(0, 0, H), mass=m, inertia6=(I, I, I, 0, 0, 0), angular_damping=0.0)
        tpl.add_shape(a, N.SHAPE_SPHERE, params=(H, 0, 0), static_friction=MU, dynamic_friction=MU)
    else:
        a = sb.add_cube(tpl, "a", H, (0, 0, H), material=(MU, MU, 0.0))
    px = factory(tpl, 1, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    rbd[a, :3] = torch.tensor([-0.4, 0.0, H])
    rbd[a, 3:7] = torch.tensor([1.0, 0, 0, 0])
    rbd[a, 7:13] = 0.0
    return px, rbd, a


def test_a_sliding_cube_stops_after_v_squared_over_two_mu_g(oracle_factory):
    """... minus v0 dt / 2: semi-implicit Euler takes the first friction impulse before it moves"""
    px, rbd, a = _world(oracle_factory)
    rbd[a, 7] = 1.0
    px.gpu_apply_all()
    for _ in range(80):
        px.step()
    px.gpu_fetch_all()
    want = 1.0 / (2 * MU * G) - 0.5 * px.timestep
    assert abs(rbd[a, 0].item() + 0.4 - want) < 1e-3 and abs(rbd[a, 7].item()) < 1e-4
    assert rbd[a, 10:13].abs().max().item() < 5e-3 and abs(rbd[a, 2].item() - H) < 1e-4      # it slides, it does not trip or hop


def test_a_spinning_cube_is_braked_by_its_four_corner_contacts(oracle_factory):
    """the manifold's points are the corners, each carrying m g / 4 at h sqrt(2) from the axis: alpha = mu m g h sqrt(2) / (2/3 m h^2)"""
    px, rbd, a = _world(oracle_factory)
    rbd[a, 12] = 20.0
    px.gpu_apply_all()
    px.step()
    px.gpu_fetch_all()
    w1 = rbd[a, 12].item()
    for _ in range(5):
        px.step()
    px.gpu_fetch_all()
    alpha = (w1 - rbd[a, 12].item()) / (5 * px.timestep)
    want = 1.5 * np.sqrt(2) * MU * G / H + 0.05 * w1          # + the actor's default angular damping
    assert abs(alpha - want) < 0.01 * want


def test_a_ball_pushed_off_sliding_ends_up_rolling_at_five_sevenths(oracle_factory):
    px, rbd, a = _world(oracle_factory, ball=True)
    rbd[a, 7] = 1.0
    px.gpu_apply_all()
    for _ in range(60):
        px.step()
    px.gpu_fetch_all()
    assert abs(rbd[a, 7].item() - 5.0 / 7.0) < 1e-4 and abs(rbd[a, 11].item() * H - rbd[a, 7].item()) < 1e-4
