"""One table for judging a solver change on the CPU oracle before it goes anywhere near the kernels ("a hard squeeze ... leaks"):

  press     a saturated prismatic drive (0.3 kg ram) on the 64 g cube on the table: how far the cube dips into the table top and the ram
            into the cube on the way, and whether the chain ends at m_ram g + f_max
  heavy     a cube of `ratio` times the mass on the 64 g cube: the light cube's dip, the forces at the end
  stacks    3 / 5 cubes, a ten times denser top cube, a leaning stack: the worst linear / angular speed over the last half second
            (StackCube's is_static thresholds are 1e-2 m/s and 0.5 rad/s)
  soak      PickCube-v1, uniform random actions: env-steps with the cube more than 3 mm inside the table, envs ever deeper than 1 cm

    python tools/oracle_solver_quality.py [press] [heavy] [stacks] [soak]          (all four without arguments)

Environment variables reach the oracle (liborc.so is loaded in this process), so an experimental switch read with getenv() in
oracle/orc_sim.c can be compared against the committed solver without rebuilding:  ORC_X=1 python tools/oracle_solver_quality.py"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle_backend import OraclePhysxSystem  # noqa: E402

from maniskill_amd import _native as N  # noqa: E402
from maniskill_amd.envs import scene_builders as sb  # noqa: E402
from maniskill_amd.physx import SceneTemplate, SimConfig  # noqa: E402

G, H = 9.81, 0.02
M_CUBE = 1000.0 * (2 * H) ** 3


def factory(tpl, n, cfg):
    return OraclePhysxSystem(tpl, n, cfg)


def start(tpl):
    px = factory(tpl, 1, SimConfig())
    px.gpu_init()
    px.set_scene_offsets(np.zeros((1, 3)))
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    return px, rbd


def press():
    from test_oracle_press import M_RAM, _press
    print("press: f_max | cube dips into the table | ram sinks into the cube | ram->cube force at the end (want)")
    for fmax in (20.0, 30.0, 50.0, 100.0, 200.0):
        px, rbd, cube, query = _press(factory, fmax, q0=0.24)
        dip = sink = 0.0
        f = 0.0
        for k in range(300):
            px.step()
            px.gpu_fetch_all()
            z, q = rbd[cube, 2].item(), px.cuda_articulation_qpos.torch()[0, 0].item()
            dip, sink = max(dip, H - z), max(sink, (z + H) - (0.3 - q - H))
            if k >= 200:
                px.gpu_query_contact_pair_impulses(query)
                f += query.cuda_impulses.torch().view(2, 3)[1, 2].item() / px.timestep / 100.0
        print(f"  {fmax:6.0f} N | {1e3 * dip:6.2f} mm | {1e3 * sink:6.2f} mm | {f:8.3f} ({M_RAM * G + fmax:.3f})")


def heavy():
    print("heavy: mass ratio | light cube dips | table->light, light->heavy at the end (want)")
    for ratio in (1, 10, 30, 100):
        m2 = ratio * M_CUBE
        tpl = SceneTemplate()
        sb.add_table_scene(tpl)
        a = tpl.add_actor("light", N.BODY_DYNAMIC, p=(0, 0, H), mass=M_CUBE, inertia6=(M_CUBE / 6 * (2 * H) ** 2,) * 3 + (0, 0, 0))
        tpl.add_shape(a, N.SHAPE_BOX, params=(H, H, H))
        b = tpl.add_actor("heavy", N.BODY_DYNAMIC, p=(0, 0, 3 * H), mass=m2, inertia6=(m2 / 6 * (2 * H) ** 2,) * 3 + (0, 0, 0))
        tpl.add_shape(b, N.SHAPE_BOX, params=(H, H, H))
        px, rbd = start(tpl)
        rbd[a, :7] = torch.tensor([0, 0, H, 1, 0, 0, 0])
        rbd[b, :7] = torch.tensor([0, 0, 3 * H, 1, 0, 0, 0])
        rbd[[a, b], 7:13] = 0
        px.gpu_apply_all()
        query = px.gpu_create_contact_pair_impulse_query([(a, tpl.body_id("table-workspace")), (b, a)])
        dip = 0.0
        for _ in range(300):
            px.step()
            px.gpu_fetch_all()
            dip = max(dip, H - rbd[a, 2].item())
        px.gpu_query_contact_pair_impulses(query)
        f = query.cuda_impulses.torch().view(2, 3)[:, 2] / px.timestep
        print(f"  {ratio:4d} | {1e3 * dip:6.2f} mm | {f[0].item():8.3f} ({(M_CUBE + m2) * G:.3f})  {f[1].item():8.3f} ({m2 * G:.3f})")


def stacks():
    print("stacks: cubes, density of the top cube, lean per layer | worst linear, angular speed over the last half second of four")
    for n, dens, off in ((3, 1000.0, 0.0), (4, 1000.0, 0.0), (5, 1000.0, 0.0), (3, 10000.0, 0.0), (3, 1000.0, 0.005)):
        tpl = SceneTemplate()
        sb.add_table_scene(tpl)
        cubes = [sb.add_cube(tpl, f"cube{k}", H, (0, 0, H + 2 * H * k), density=(dens if k == n - 1 else 1000.0)) for k in range(n)]
        px, rbd = start(tpl)
        for k, c in enumerate(cubes):
            rbd[c, :7] = torch.tensor([off * k, 0.0, H + 2 * H * k, 1, 0, 0, 0])
            rbd[c, 7:13] = 0
        px.gpu_apply_all()
        lin = ang = 0.0
        for t in range(400):
            px.step()
            if t >= 350:
                px.gpu_fetch_all()
                lin = max(lin, rbd[cubes, 7:10].norm(dim=1).max().item())
                ang = max(ang, rbd[cubes, 10:13].norm(dim=1).max().item())
        print(f"  {n} {dens:7.0f} {off:5.3f} | {lin:.4f} m/s  {ang:.4f} rad/s")


def soak(n=512, steps=1000):
    from maniskill_amd.envs.pick_cube import PickCubeEnv
    env = PickCubeEnv(num_envs=n, px_factory=factory)
    env.reset(seed=7)
    gen = torch.Generator().manual_seed(1)
    deep = torch.zeros(n)
    zmin = torch.full((n,), 9.0)
    for _ in range(steps):
        env.step(2 * torch.rand(n, env.action_dim, generator=gen) - 1)
        z = env._rbd[:, env._b_cube, 2]
        zmin = torch.minimum(zmin, z)
        deep += (z < H - 0.003).float()
    pen = H - zmin
    print(f"soak: PickCube-v1, {n} envs x {steps} random actions: {int(deep.sum())} of {n * steps} env-steps with the cube more than 3 mm inside the "
          f"table; envs ever deeper than 1 / 3 / 5 / 10 mm: {[int((pen > x).sum()) for x in (1e-3, 3e-3, 5e-3, 1e-2)]}; state finite: "
          f"{bool(torch.isfinite(env._rbd).all())}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["press", "heavy", "stacks", "soak"]
    for name in which:
        dict(press=press, heavy=heavy, stacks=stacks, soak=soak)[name]()
