"""Impacts as known answers for the CPU oracle: whatever the contact rows do, a collision between bodies that nothing else touches conserves
linear momentum and angular momentum (about any fixed point, here the origin / the hinge), and an inelastic one does not create energy.
None of this solver's constants enters: independent checks of the row Jacobians (free bodies and articulation links in one system), of
the impulse application and of the integration."""
import numpy as np
import torch

from maniskill_amd import _native as N
from maniskill_amd.physx import SceneTemplate, SimConfig

H = 0.02


def _start(factory, tpl):
    cfg = SimConfig()
    cfg.scene_config.gravity = (0.0, 0.0, 0.0)
    px = factory(tpl, 1, cfg)
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    return px, px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)


def test_an_off_centre_collision_of_two_free_cubes_conserves_linear_and_angular_momentum(oracle_factory):
    masses = (0.2, 0.5)
    for offset in (0.0, 0.015, 0.03):
        tpl = SceneTemplate()
        ids = []
        for k, (p, m) in enumerate(zip(((-0.1, 0.0, 0.5), (0.0, offset, 0.5)), masses)):
            b = tpl.add_actor(f"c{k}", N.BODY_DYNAMIC, p=p, mass=m, inertia6=(m / 6 * (2 * H) ** 2,) * 3 + (0, 0, 0), angular_damping=0.0, linear_damping=0.0)
            tpl.add_shape(b, N.SHAPE_BOX, params=(H, H, H), static_friction=0.3, dynamic_friction=0.3)
            ids.append(b)
        px, rbd = _start(oracle_factory, tpl)
        rbd[ids[0], 7] = 1.0
        px.gpu_apply_all()

        def momenta():
            px.gpu_fetch_all()
            P, L, E = np.zeros(3), np.zeros(3), 0.0
            for b, m in zip(ids, masses):
                x, v, w = (rbd[b, a:a + 3].numpy().astype(np.float64) for a in (0, 7, 10))
                I = m / 6 * (2 * H) ** 2
                P += m * v
                L += m * np.cross(x, v) + I * w
                E += 0.5 * m * v @ v + 0.5 * I * w @ w
            return P, L, E
        P0, L0, E0 = momenta()
        for _ in range(30):
            px.step()
        P1, L1, E1 = momenta()
        # (angular momentum to 0.5 %: the positions advance with the sub-steps' velocities, not with the final one, so r x m v carries an
        #  O(dt * delta v) error of the steps in contact; the impulses themselves are equal and opposite at one point)
        assert np.abs(P1 - P0).max() < 1e-5 and np.abs(L1 - L0).max() < 5e-3 * np.abs(L0).max(), (offset, P1 - P0, L1 - L0)
        assert E1 < E0 and rbd[ids[1], 7].item() > 0.2                               # it was hit, and nothing was gained
        if offset > 0:
            assert abs(rbd[ids[1], 12].item()) > 1.0                                # off centre: both leave spinning


def test_a_bat_on_a_hinge_hitting_a_free_cube_conserves_angular_momentum_about_the_hinge(oracle_factory):
    """An articulation link (revolute about x, a box 0.3 m below the hinge) swinging at 2 rad/s into a free cube: I_hinge w + (r x m v)_x + I_c w_c,x
    stays what it was (the hinge can only transmit a force through its axis and moments about the other two), and the first exchange is the
    inelastic one: w' = I_h w / (I_h + m L^2) to a few per cent."""
    L, mb, Ib, mc = 0.3, 1.0, 0.02, 0.2
    tpl = SceneTemplate()
    art = tpl.add_articulation("bat", root_p=(0, 0, 1.0))
    base = tpl.add_link(art, "base", -1, N.JOINT_FIXED, mass=1.0, inertia6=(1e-2,) * 3 + (0, 0, 0))
    bob = tpl.add_link(art, "bob", base, N.JOINT_REVOLUTE, joint_name="hinge", mass=mb, com=(0, 0, -L), inertia6=(Ib, Ib, Ib, 0, 0, 0), disable_gravity=True)
    tpl.add_shape(bob, N.SHAPE_BOX, p=(0, 0, -L), params=(H, H, H), static_friction=0.0, dynamic_friction=0.0)
    Ic = mc / 6 * (2 * H) ** 2
    cube = tpl.add_actor("cube", N.BODY_DYNAMIC, p=(0, 0.06, 1.0 - L), mass=mc, inertia6=(Ic, Ic, Ic, 0, 0, 0), angular_damping=0.0, linear_damping=0.0,
                         disable_gravity=True)
    tpl.add_shape(cube, N.SHAPE_BOX, params=(H, H, H), static_friction=0.0, dynamic_friction=0.0)
    px, rbd = _start(oracle_factory, tpl)
    rbd[base, :7] = torch.tensor([0, 0, 1.0, 1, 0, 0, 0])
    rbd[cube, :7] = torch.tensor([0, 0.06, 1.0 - L, 1, 0, 0, 0])
    rbd[cube, 7:13] = 0.0
    px.cuda_articulation_qvel.torch()[0, 0] = 2.0
    px.gpu_apply_all()
    Ih = Ib + mb * L * L

    def state():
        px.gpu_fetch_all()
        w = px.cuda_articulation_qvel.torch()[0, 0].item()
        x = rbd[cube, :3].numpy().astype(np.float64) - np.array([0.0, 0.0, 1.0])
        v, wc = rbd[cube, 7:10].numpy().astype(np.float64), rbd[cube, 10:13].numpy().astype(np.float64)
        return Ih * w + mc * np.cross(x, v)[0] + Ic * wc[0], 0.5 * Ih * w * w + 0.5 * mc * v @ v + 0.5 * Ic * wc @ wc, w, v
    L0, E0, _, _ = state()
    seen = []
    for k in range(40):
        px.step()
        if k % 8 == 7:
            seen.append(state())
    for Lk, Ek, w, v in seen:
        assert abs(Lk - L0) < 2e-4 * L0 and Ek < E0
    w1 = seen[0][2]
    assert abs(w1 - 2.0 * Ih / (Ih + mc * L * L)) < 0.02 * 2.0            # first exchange: common speed at the point of contact
    assert seen[-1][3][1] > 0.45                                            # the cube flies off along +y
