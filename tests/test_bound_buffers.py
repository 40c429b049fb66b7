"""msk_bind_buffers / msk_batch (include/msk_physx.h): several contexts behind ONE set of sapien tensors -- what the shim does for scenes
whose sub-scenes differ in structure.  Two contexts bound to row ranges of shared tensors must compute what two stand-alone contexts
compute, and the batched boundary calls must equal the per-context ones."""
import pytest
import torch

from maniskill_amd import _native as N
from maniskill_amd.physx import batch_call


def _make(factory, n):
    from maniskill_amd.envs.pick_cube import PickCubeEnv
    env = PickCubeEnv(num_envs=n, px_factory=factory, fused=False) if factory is not None else PickCubeEnv(num_envs=n, device="cuda:0", fused=False)
    env.reset(seed=5)
    return env


def _run(factory, device):
    a, b = _make(factory, 3), _make(factory, 2)          # stand-alone
    c, d = _make(factory, 3), _make(factory, 2)          # bound to shared tensors
    pa, pb, pc, pd = a.px, b.px, c.px, d.px
    nb, na, md = pc.bodies_per_env, pc.arts_per_env, pc.max_dof
    pitch = md + 3                                       # wider rows than either context needs, as SAPIEN pads to the largest articulation
    z = lambda r, w: torch.zeros(r, w, dtype=torch.float32, device=device)   # noqa: E731
    shared = {"rigid_body_data": z(5 * nb, 13), "rigid_body_force": z(5 * nb, 4), "rigid_body_torque": z(5 * nb, 4)}
    for nm in ("qpos", "qvel", "qacc", "qf", "target_qpos", "target_qvel"):
        shared["articulation_" + nm] = z(5 * na, pitch)
    pc.bind_buffers(shared, 0, 0)
    pd.bind_buffers(shared, 3 * nb, 3 * na)
    full = N.FETCH_RIGID_DATA | N.FETCH_ART_QPOS | N.FETCH_ART_QVEL | N.FETCH_ART_QACC | N.FETCH_ART_TARGETS
    batch_call([pc, pd], N.BATCH_FETCH, full)
    gen = torch.Generator().manual_seed(1)
    for k in range(12):
        tq = 0.2 * torch.rand(5 * na, md, generator=gen).to(device)
        pa.cuda_articulation_target_qpos.torch()[:] = pa.cuda_articulation_qpos.torch() + tq[:3 * na]
        pb.cuda_articulation_target_qpos.torch()[:] = pb.cuda_articulation_qpos.torch() + tq[3 * na:]
        shared["articulation_target_qpos"][:, :md] = shared["articulation_qpos"][:, :md] + tq
        pa.gpu_apply_articulation_target_position(); pb.gpu_apply_articulation_target_position()
        batch_call([pc, pd], N.BATCH_APPLY, N.APPLY_ART_TARGET_QPOS)
        for _ in range(2):
            pa.step(); pb.step()
            batch_call([pc, pd], N.BATCH_STEP)
        pa.gpu_fetch_all(); pb.gpu_fetch_all()
        batch_call([pc, pd], N.BATCH_FETCH, full)
    ref_rb = torch.cat([pa.cuda_rigid_body_data.torch(), pb.cuda_rigid_body_data.torch()])
    ref_q = torch.cat([pa.cuda_articulation_qpos.torch(), pb.cuda_articulation_qpos.torch()])
    assert torch.equal(shared["rigid_body_data"], ref_rb)
    assert torch.equal(shared["articulation_qpos"][:, :md], ref_q) and (shared["articulation_qpos"][:, md:] == 0).all()
    assert tuple(pc.cuda_articulation_qpos.torch().shape) == (3 * na, pitch)
    assert (ref_q - torch.cat([a.px.cuda_articulation_target_qpos.torch(), b.px.cuda_articulation_target_qpos.torch()])).abs().max() < 0.3


def test_bound_contexts_equal_standalone_on_cpu_checker(oracle_factory):
    _run(oracle_factory, "cpu")


@pytest.mark.gpu
def test_bound_contexts_equal_standalone_on_hip(built):
    _run(None, "cuda:0")
