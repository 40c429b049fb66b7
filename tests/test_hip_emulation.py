"""The PRODUCT's HIP kernels in the GPU-less container: maniskill_amd/csrc/msk_physx.hip and its headers, as they are, compiled for the
CPU against the programming-model emulation of tests/hipemu (one fiber per work-item, wavefront rendezvous for cross-lane operations, host
memory) and run against the CPU oracle -- the comparisons the -m gpu tests make on hardware, at sizes the emulation finishes in seconds.

What this can and cannot show is said in tests/hipemu/hip/hip_runtime.h: arithmetic, indexing, lane mappings, lists / scans / masks are
checked bit for bit; races, fences and timing are not -- the -m gpu tests stay the parity tests proper."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.push_t import PushTEnv
from maniskill_amd.envs.stack_cube import StackCubeEnv
from maniskill_amd.physx import SimConfig

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu_factory(built):
    from emu_backend import EmuPhysxSystem
    return lambda tpl, n, cfg: EmuPhysxSystem(tpl, n, cfg)


def test_the_emulated_library_is_the_product_source(emu_factory):
    """libmsk_emu.so is built from maniskill_amd/csrc/msk_physx.hip itself (tests/hipemu/Makefile) and exports the product's C ABI"""
    from emu_backend import EMU_LIB, emu_lib
    from maniskill_amd import _native as N
    lib = emu_lib()
    assert all(hasattr(lib, n) for n in N.EXPORTS) and lib.has_task_kernels
    mk = open(os.path.join(HERE, "hipemu", "Makefile")).read()
    assert "$(CSRC)/msk_physx.hip" in mk and os.path.getmtime(EMU_LIB) >= os.path.getmtime(os.path.join(HERE, "..", "maniskill_amd", "csrc", "msk_physx.hip"))


@pytest.mark.parametrize("cls,adim,n,steps", [(PickCubeEnv, 8, 9, 40), (PegInsertionSideEnv, 8, 6, 40), (PushTEnv, 7, 5, 40), (StackCubeEnv, 8, 4, 30)])
def test_physics_rollout_matches_oracle_bit_for_bit(emu_factory, oracle_factory, cls, adim, n, steps):
    """k_dynamics / k_narrowphase / k_csolve (+ apply / fetch / kinematics / queries) against orc_sim.c: same host code on both sides, so the
    observations, rewards and the contact lists with their pair ids must be the same bits (tests/test_gpu_parity.py on hardware)"""
    emu = cls(num_envs=n, px_factory=emu_factory, fused=False)
    cpu = cls(num_envs=n, px_factory=oracle_factory)
    oe, _ = emu.reset(seed=2022); oc, _ = cpu.reset(seed=2022)
    assert torch.equal(oe, oc)
    gen = torch.Generator().manual_seed(0)
    for t in range(steps):
        a = 2 * torch.rand(n, adim, generator=gen) - 1
        oe, re, te, ue, _ = emu.step(a)
        oc, rc, tc, uc, _ = cpu.step(a)
        assert torch.equal(oe, oc) and torch.equal(re, rc) and torch.equal(te, tc) and torch.equal(ue, uc), (cls.__name__, t, (oe - oc).abs().max().item())
        if t % 8 == 7:
            for e in range(n):
                gi, gv = emu.px.get_contacts(e)
                ci, cv = cpu.px.get_contacts(e)
                assert gi.shape == ci.shape and (gi == ci).all() and np.array_equal(gv, cv), (cls.__name__, t, e)
    assert torch.equal(emu.get_state(), cpu.get_state()) and emu.px.get_overflow() & 6 == 0


@pytest.mark.parametrize("cls,adim,kw", [(PickCubeEnv, 8, {}), (PegInsertionSideEnv, 8, {}), (PushTEnv, 7, {}), (PickCubeEnv, 7, dict(control_mode="pd_ee_delta_pose"))])
def test_fused_task_kernels_match_the_torch_path(emu_factory, oracle_factory, cls, adim, kw):
    """k_*_set_action(_ee) / msk_control_step / k_*_observe against the torch mirror of the reference's task code on the oracle"""
    n = 7
    emu = cls(num_envs=n, px_factory=emu_factory, fused=True, **kw)
    cpu = cls(num_envs=n, px_factory=oracle_factory, **kw)
    assert emu.fused and not cpu.fused
    emu.reset(seed=5); cpu.reset(seed=5)
    gen = torch.Generator().manual_seed(3)
    for t in range(25):
        a = 2 * torch.rand(n, adim, generator=gen) - 1
        oe, re, te, ue, ie = emu.step(a)
        oc, rc, tc, uc, ic = cpu.step(a)
        assert torch.allclose(oe, oc, atol=1e-4) and torch.allclose(re, rc, atol=1e-5) and torch.equal(te, tc) and torch.equal(ue, uc), (cls.__name__, t)
        assert torch.equal(ie["success"], ic["success"])


@pytest.mark.parametrize("mode", ["1", "0"])
def test_pictures_match_oracle_bit_for_bit(emu_factory, oracle_factory, monkeypatch, mode):
    """k_render_splat (and k_render_env, MSK_RENDER_MODE=0) + k_render_texture against orc_render.c: PositionSegmentation, depth, segmentation
    and Color of PushT and PickCube rollouts"""
    monkeypatch.setenv("MSK_RENDER_MODE", mode)
    for cls, adim in ((PushTEnv, 7), (PickCubeEnv, 8)):
        n = 3
        emu = cls(num_envs=n, px_factory=emu_factory, fused=False, obs_mode="rgb+depth+segmentation")
        cpu = cls(num_envs=n, px_factory=oracle_factory, obs_mode="rgb+depth+segmentation")
        emu.reset(seed=6); cpu.reset(seed=6)
        gen = torch.Generator().manual_seed(2)
        for _ in range(6):
            a = 2 * torch.rand(n, adim, generator=gen) - 1
            oe = emu.step(a)[0]; oc = cpu.step(a)[0]
        ce, cc = oe["sensor_data"]["base_camera"], oc["sensor_data"]["base_camera"]
        for k in cc:
            assert torch.equal(ce[k], cc[k]), (cls.__name__, mode, k, int((ce[k] != cc[k]).sum()))
        assert torch.equal(emu.camera.get_picture_cuda().torch(), cpu.camera.get_picture_cuda().torch())
        assert torch.equal(emu.camera.get_picture_cuda("Color").torch(), cpu.camera.get_picture_cuda("Color").torch())
        assert emu.px.get_overflow() == 0


def test_planes_only_pictures_leave_the_texture_alone_and_keep_the_planes_bits(emu_factory, oracle_factory):
    """msk_camera_set_outputs(position_texture = 0), what the fused envs ask for (no obs mode of theirs hands out `position`): the rasteriser neither computes
    camera-space x, y nor stores the int16 x 4 texture -- a sentinel in it survives the picture -- and depth / segmentation are the oracle's bits; asking for
    the texture all the same renders it for that request"""
    n = 2
    emu = PushTEnv(num_envs=n, px_factory=emu_factory, fused=False, obs_mode="depth+segmentation")
    cpu = PushTEnv(num_envs=n, px_factory=oracle_factory, obs_mode="depth+segmentation")
    emu.reset(seed=4); cpu.reset(seed=4)
    a = torch.zeros(n, 7)
    emu.step(a); cpu.step(a)
    assert emu.camera._position_texture is False
    emu.camera._tex.fill_(-12345)
    emu.camera.take_picture(); cpu.camera.take_picture()
    assert bool((emu.camera._tex == -12345).all())
    oe, oc = emu.camera.get_obs(), cpu.camera.get_obs()
    assert torch.equal(oe["depth"], oc["depth"]) and torch.equal(oe["segmentation"], oc["segmentation"]) and int((oc["segmentation"] != 0).sum()) > 1000
    assert torch.equal(emu.camera.get_picture_cuda().torch(), cpu.camera.get_picture_cuda().torch())      # rendered on request
    emu.camera._tex.fill_(-12345)
    emu.camera.take_picture()
    assert bool((emu.camera._tex == -12345).all())      # ... and the configured mode is back in force


def test_lights_textures_and_picture_sizes_match_oracle(emu_factory, oracle_factory):
    """the scenes of tests/test_render.py's -m gpu tests: point / spot lights, a textured wall (perspective, mip levels), pictures that are not
    128 x 128 with splatted, medium and large triangles"""
    import test_render as R
    lights = dict(ambient=(0.05, 0.1, 0.15), directional=[((1, 1, -1), (0.2, 0.2, 0.2))], point=[((-1.5, -1.0, 0.7), (0.5, 0.4, 0.3))],
                  spot=[((-1.875, -1.875, 1.0), (0.2, 0.3, -1), 0.5, 0.8, (0.5, 0.5, 0.5))])
    a, b = R._lit_floor(emu_factory, lights, n=2), R._lit_floor(oracle_factory, lights, n=2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for tilt, size in ((0.0, (128, 128)), (0.9, (64, 64)), (-0.6, (256, 128))):
        wl = dict(ambient=(0.3, 0.4, 0.5), directional=[((1, 0.3, -0.5), (0.6, 0.5, 0.4))], point=[((1.0, 0.5, 0.5), (0.5, 0.5, 0.5))])
        a, b = R._textured_wall(emu_factory, n=2, tilt=tilt, size=size, lights=wl), R._textured_wall(oracle_factory, n=2, tilt=tilt, size=size, lights=wl)
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]), (tilt, size, int((a[0] != b[0]).sum()))
    rng = np.random.default_rng(5)
    boxes = [((2.0 + 0.5 * rng.random(), float(y), float(z)), (0.02 + 0.03 * rng.random(),) * 3) for y, z in rng.uniform(-1.2, 1.2, size=(40, 2))]
    boxes += [((4.0, 0.0, 0.0), (0.1, 1.5, 1.5)), ((3.2, 0.8, 0.3), (0.3, 0.3, 0.3)), ((1.0, 0.05, -0.1), (0.2, 0.2, 0.2))]
    for W, H in ((256, 128), (64, 64)):
        cam = R.CameraConfig("c", (0, 0, 0), (1, 0, 0, 0), W, H, np.pi / 2, 0.01, 100.0)
        pics = []
        for fac in (emu_factory, oracle_factory):
            px, grp, ids = R._boxes_scene(fac, boxes, n=2, cam=cam, ground=True)
            grp.get_picture_cuda("Color"); grp.take_picture()
            pics.append((grp.get_picture_cuda("PositionSegmentation").torch().clone(), grp.get_picture_cuda("Color").torch().clone()))
        assert torch.equal(pics[0][0], pics[1][0]) and torch.equal(pics[0][1], pics[1][1]), (W, H)


def test_generic_ik_kernel_matches_its_torch_mirror(emu_factory):
    """k_ik_delta (msk_compute_ik_delta) on chains of 5 (primal form), 6, 7 and 8 (dual form) joints with prismatic ones"""
    import test_ik_generic as G
    from maniskill_amd.agents.ik import SerialChain
    for kinds, root in (("rrprr", "mount"), ("rrrrrr", "base"), ("rprrrrr", "mount"), ("prrrrrrr", "base")):
        tpl, base, mount, links, tool = G._arm(kinds, seed=2 + len(kinds))
        n = 40
        px = emu_factory(tpl, n, SimConfig()); px.gpu_init()
        rb = mount if root == "mount" else base
        chain = SerialChain(tpl, tool, rb, links)
        gen = torch.Generator().manual_seed(8)
        q0 = 0.8 * (2 * torch.rand(n, len(kinds), generator=gen) - 1) * torch.tensor([0.3 if k == "p" else 1.0 for k in kinds])
        P = G._poses(px, n, q0)
        delta = 0.05 * (2 * torch.rand(n, 6, generator=gen) - 1)
        want = chain.ik_delta(P, q0, delta)
        got = px.compute_ik_delta(tool, rb, links, delta, commit_targets=True)
        J = chain.jacobian(P)
        assert torch.allclose(torch.bmm(J, (got - q0).unsqueeze(-1)), torch.bmm(J, (want - q0).unsqueeze(-1)), atol=1e-4), kinds
        smin = torch.linalg.svdvals(J.double())[:, -1]
        err = (got - want).abs().amax(dim=1)
        assert (err[smin > 0.1] < 1e-4).all() and (err < 2e-3).all(), (kinds, err.max().item())
        px.gpu_fetch_all()
        assert torch.equal(px.cuda_articulation_target_qpos.torch()[:, :len(kinds)], got)
        with pytest.raises(RuntimeError, match="between the root"):
            px.compute_ik_delta(links[1], rb, links, delta)


def test_substeps_in_one_call_and_env_partitions(emu_factory, oracle_factory):
    """msk_step_n and the partitioned launch chains (msk_set_step_parts) give the plain loop's bits"""
    n = 128           # partitions are whole 64-env chunks
    envs = [PickCubeEnv(num_envs=n, px_factory=emu_factory, fused=False) for _ in range(2)]
    for e in envs:
        e.reset(seed=11)
    assert envs[1].px.set_step_parts(2) == 2 and envs[1].px.step_parts == 2
    gen = torch.Generator().manual_seed(4)
    for t in range(6):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        o0 = envs[0].step(a)[0]; o1 = envs[1].step(a)[0]
        assert torch.equal(o0, o1), t
    cpu = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    cpu.reset(seed=11)
    gen = torch.Generator().manual_seed(4)
    for t in range(6):
        oc = cpu.step(2 * torch.rand(n, 8, generator=gen) - 1)[0]
    assert torch.equal(o0, oc)


def _ref_run(backend, env_id, n, steps, obs_mode="state"):
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_emu_vs_oracle.py"), backend, env_id, str(n), str(steps), obs_mode], cwd=HERE, capture_output=True,
                       text=True, timeout=3000)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("EVO ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(line[-1][4:])


@pytest.mark.parametrize("env_id,n,steps,obs_mode", [("PickCube-v1", 4, 6, "rgb+depth+segmentation"), ("RotateValveLevel1-v1", 4, 4, "state"),
                                                     ("OpenCabinetDrawer-v1", 3, 3, "state"), ("UnitreeG1Stand-v1", 2, 10, "state")])
def test_the_references_own_envs_over_the_shim(built, env_id, n, steps, obs_mode):
    """the reference's unmodified task code over the sapien shim on the emulated HIP library and on the oracle: the same bits in every buffer
    the reference reads (several structural groups -> msk_bind_buffers / msk_batch / the k_multi_* kernels for the valve and the cabinets; 43
    coordinates -> k_dynamics<64, 64> and k_csolve<64, 64> for the humanoid)"""
    import ref_harness
    if ref_harness.find_reference() is None:
        pytest.skip("no reference checkout / build")
    if env_id == "OpenCabinetDrawer-v1":
        assets = "/tmp/ms_assets_synth_emu"
        ref = ref_harness.find_reference()
        meta = os.path.join(ref, "mani_skill", "assets", "partnet_mobility", "meta")
        subprocess.check_call([sys.executable, os.path.join(HERE, "..", "tools", "make_synthetic_partnet.py"), "--out", assets, "--max-drawers", "1", "--ids-from",
                               os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from", os.path.join(meta, "info_cabinet_door_train.json")],
                              stdout=subprocess.DEVNULL)
        os.environ["MS_ASSET_DIR"] = assets
    a, b = _ref_run("emu", env_id, n, steps, obs_mode), _ref_run("oracle", env_id, n, steps, obs_mode)
    assert a["finite"] and b["finite"] and a["sha"] == b["sha"], (a, b)


def test_floating_bases_locked_axes_rounded_shapes_and_bounces(emu_factory, oracle_factory):
    """the scenes of the -m gpu twins in tests/test_floating_base.py, test_locked_axes.py and test_rounded_shapes.py (free-floating roots,
    locked world axes of dynamic actors, sphere / capsule / cylinder hulls through GJK / EPA, restitution): emulated HIP == oracle, bit for bit"""
    import test_floating_base as F
    import test_locked_axes as L
    import test_rounded_shapes as RS
    from maniskill_amd import _native as N
    a = F._rollout(*F._box_world(emu_factory, 3, True), 120)
    b = F._rollout(*F._box_world(oracle_factory, 3, True), 120)
    assert torch.equal(a, b)
    pa, ba, ra, _ = F._chain(emu_factory, 3, (0, 0, -9.81))
    pb, bb, rb, _ = F._chain(oracle_factory, 3, (0, 0, -9.81))
    for _ in range(80):
        pa.step(); pb.step()
    pa.gpu_fetch_all(); pb.gpu_fetch_all()
    assert torch.equal(ra, rb) and torch.equal(pa.cuda_articulation_qpos.torch(), pb.cuda_articulation_qpos.torch())
    th = np.deg2rad(20.0)
    q = (np.cos(th / 2), 0.0, np.sin(th / 2), 0.0)
    zc = L.H * (np.cos(th) + np.sin(th)) + 0.01
    for lock in ([0, 0, 0, 1, 1, 1], [1, 0, 0, 0, 1, 0]):
        x = L._world(oracle_factory, 3, lock, z=zc, q=q, v0=(0.2, 0.1, 0), w0=(1.0, 2.0, 3.0))
        y = L._world(emu_factory, 3, lock, z=zc, q=q, v0=(0.2, 0.1, 0), w0=(1.0, 2.0, 3.0))
        for _ in range(5):
            L._run(x[0], 10); L._run(y[0], 10)
            assert torch.equal(x[2], y[2]), lock
    cases = [(N.SHAPE_SPHERE, (0.03, 0, 0), 0.05, (1, 0, 0, 0), dict(v0=(0.3, 0.1, 0), friction=1.0)),
             (N.SHAPE_CAPSULE, (0.02, 0.05, 0), 0.04, (0.9659258, 0, 0.2588190, 0), dict(w0=(0, 0, 2.0))),
             (N.SHAPE_CYLINDER, (0.03, 0.02, 0), 0.05, (0.8660254, 0.5, 0, 0), dict(v0=(0.1, 0, 0)))]
    for shape, params, z, qq, kw in cases:
        x = RS._world(emu_factory, 3, shape, params, z, q=qq, **kw)
        y = RS._world(oracle_factory, 3, shape, params, z, q=qq, **kw)
        for k in range(50):
            x[0].step(); y[0].step()
        x[0].gpu_fetch_all(); y[0].gpu_fetch_all()
        assert torch.equal(x[2], y[2]), (shape, float((x[2] - y[2]).abs().max()))
    x, y = RS._bouncer(emu_factory, 2, 0.8), RS._bouncer(oracle_factory, 2, 0.8)
    for k in range(150):
        x[0].step(); y[0].step()
    x[0].gpu_fetch_all(); y[0].gpu_fetch_all()
    assert torch.equal(x[2], y[2])


def test_every_solver_class_and_the_link_joint_forces(emu_factory, oracle_factory):
    """the same PickCube rollout with every env forced through solver class 1, 2 and 3 (A in LDS / in global memory) gives class 0's bits;
    link incoming joint forces (k_link_forces) equal the oracle's"""
    n = 5
    ref = PickCubeEnv(num_envs=n, px_factory=oracle_factory)
    ref.reset(seed=21)
    envs = []
    for caps in ((64, 64, 64), (-1, 64, 64), (-1, -1, 64), (-1, -1, -1)):
        e = PickCubeEnv(num_envs=n, px_factory=emu_factory, fused=False)
        e.reset(seed=21)
        e.px.set_solver_classes(caps)
        envs.append(e)
    gen = torch.Generator().manual_seed(9)
    for t in range(15):
        a = 2 * torch.rand(n, 8, generator=gen) - 1
        oc = ref.step(a)[0]
        outs = [e.step(a)[0] for e in envs]
        for k, o in enumerate(outs):
            assert torch.equal(o, oc), (t, k)
    counts = [e.px.get_solver_class_counts() for e in envs]
    assert counts[1][0] == 0 and counts[2][0] == 0 and counts[2][1] == 0 and counts[3][3] == n, counts
    envs[0].px.gpu_fetch_articulation_link_incoming_joint_forces(); ref.px.gpu_fetch_articulation_link_incoming_joint_forces()
    assert torch.equal(envs[0].px.cuda_articulation_link_incoming_joint_forces.torch(), ref.px.cuda_articulation_link_incoming_joint_forces.torch())


def test_the_whole_reference_task_zoo_on_the_emulated_library(built):
    """every registered task of the reference that needs no download (tests/ref_env_zoo.py: Panda, SO100, two-robot, Allegro, D'Claw, TriFinger,
    Unitree G1, ANYmal-like MJCF ants and humanoids, the Draw tasks' pose-only actors, FMB's 39 coordinates ...) built by the reference's own code
    over the shim, reset and stepped with the same seeded actions on the emulated HIP library and on the oracle: every buffer the reference reads
    has the same bits (FMBAssembly1Easy-v1 too: it starts interpenetrating, with many hull-queue items per wavefront, and runs over the contact capacity)."""
    import ref_harness
    import test_reference_conformance as T
    if ref_harness.find_reference() is None:
        pytest.skip("no reference checkout / build")
    os.environ["ZOO_HASH"] = "1"
    try:
        a, b = T._run_zoo("emu", "4", (), "2"), T._run_zoo("oracle", "4", (), "2")
    finally:
        os.environ.pop("ZOO_HASH", None)
    assert len(a) >= 45 and all(v.startswith("ok") for v in a.values()) and all(v.startswith("ok") for v in b.values()), ({k: v for k, v in a.items() if not v.startswith("ok")})
    differ = sorted(k for k in a if a[k] != b[k])
    assert not differ, differ
