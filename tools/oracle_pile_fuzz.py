"""Random piles on the CPU oracle: 2-5 random boxes dropped on one another with random spin; flags a seed when, after four seconds, a box
that is still over the table is inside the table top or not at rest (tests/test_oracle_embedded.py pins the seeds this found in round 3).
    python tools/oracle_pile_fuzz.py [seeds=40]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import warnings; warnings.simplefilter("ignore")
import numpy as np, torch
from oracle_backend import OraclePhysxSystem
from maniskill_amd import _native as N
from maniskill_amd.envs import scene_builders as sb
from maniskill_amd.physx import SceneTemplate, SimConfig
def quat_rand(rng):
    q = rng.normal(size=4); return q/np.linalg.norm(q)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv)>1 else 40):
    rng = np.random.default_rng(seed)
    tpl = SceneTemplate(); sb.add_table_scene(tpl)
    nb = rng.integers(2, 6); bodies=[]; kinds=[]
    for k in range(nb):
        kind = "box"
        dens = rng.uniform(300, 3000)
        if kind=="box":
            hs = rng.uniform(0.01, 0.035, size=3); m = dens*8*hs.prod(); I = m/3*np.array([hs[1]**2+hs[2]**2, hs[0]**2+hs[2]**2, hs[0]**2+hs[1]**2])
            b = tpl.add_actor(f"b{k}", N.BODY_DYNAMIC, p=(0,0,1), mass=m, inertia6=tuple(I)+(0,0,0)); tpl.add_shape(b, N.SHAPE_BOX, params=tuple(hs))
        elif kind=="sphere":
            r = rng.uniform(0.01,0.03); m = dens*4/3*np.pi*r**3
            b = tpl.add_actor(f"b{k}", N.BODY_DYNAMIC, p=(0,0,1), mass=m, inertia6=(0.4*m*r*r,)*3+(0,0,0)); tpl.add_shape(b, N.SHAPE_SPHERE, params=(r,0,0))
        else:
            r = rng.uniform(0.008,0.02); hl = rng.uniform(0.01,0.04); m = dens*(np.pi*r*r*2*hl+4/3*np.pi*r**3)
            I = (0.5*m*r*r, m*(r*r/4+hl*hl/3+0.0), m*(r*r/4+hl*hl/3))
            b = tpl.add_actor(f"b{k}", N.BODY_DYNAMIC, p=(0,0,1), mass=m, inertia6=I+(0,0,0)); tpl.add_shape(b, N.SHAPE_CAPSULE, params=(r,hl,0))
        bodies.append(b); kinds.append(kind)
    px = OraclePhysxSystem(tpl, 1, SimConfig()); px.gpu_init(); px.set_scene_offsets(np.zeros((1,3)))
    rbd = px.cuda_rigid_body_data.torch().view(px.bodies_per_env, 13)
    rbd[tpl.body_id("table-workspace"), :7] = torch.tensor([-0.12, 0.0, -sb.TABLE_HEIGHT, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    for k,b in enumerate(bodies):
        rbd[b,:3] = torch.tensor([rng.uniform(-0.04,0.04), rng.uniform(-0.04,0.04), 0.06+0.07*k], dtype=torch.float32)
        rbd[b,3:7] = torch.tensor(quat_rand(rng), dtype=torch.float32); rbd[b,7:13] = torch.tensor(rng.normal(size=6)*np.array([0.2,0.2,0.2,2,2,2]), dtype=torch.float32)
    px.gpu_apply_all()
    vmax=0; zmin=9; E=[]
    for t in range(400):
        px.step(); px.gpu_fetch_all()
        vmax=max(vmax, rbd[bodies,7:10].norm(dim=1).max().item()); zmin=min(zmin, rbd[bodies,2].min().item())
    fin = bool(torch.isfinite(rbd).all()); vend = rbd[bodies,7:10].norm(dim=1).max().item(); wend = rbd[bodies,10:13].norm(dim=1).max().item()
    ontable = (rbd[bodies,0].abs() < 0.5) & (rbd[bodies,1].abs() < 0.5) & (rbd[bodies,2] > -0.1); flag = (not fin) or vmax>6 or (ontable & (rbd[bodies,2] < 0.005)).any().item() or (rbd[bodies,7:10].norm(dim=1)[ontable] > 0.02).any().item() or (rbd[bodies,10:13].norm(dim=1)[ontable] > 0.3).any().item()
    bad += flag
    if flag or seed<3: print(seed, nb, kinds, "finite",fin,"vmax %.2f zmin %.4f vend %.4f wend %.3f ovf %d"%(vmax,zmin,vend,wend,px.get_overflow()), "<<<" if flag else "")
print("flagged", bad)
