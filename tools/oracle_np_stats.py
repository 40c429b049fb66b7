#!/usr/bin/env python3
"""How much GJK / EPA work does the narrowphase do per substep?  (CPU only: counts in the oracle, whose per-pair algorithm the HIP
lane-group kernels mirror step for step, so the iteration counts ARE the length of the GPU's dependent chain per hull pair.)

    python tools/oracle_np_stats.py [--env PickCube-v1] [--envs 64] [--steps 200]

Builds oracle/*.c with -DORC_STATS into a temporary library, runs a random-action rollout and prints, per phase of the rollout, hull
pairs tested / culled by the oriented-box test / reaching GJK / reaching EPA and the mean iteration counts.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="PickCube-v1")
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    lib_path = os.path.join(tempfile.mkdtemp(), "liborc_stats.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("orc_api.c", "orc_sim.c", "orc_collide.c", "orc_render.c")]
    subprocess.check_call(["cc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp", "-fPIC", "-fvisibility=hidden",
                           "-DORC_STATS", "-I" + os.path.join(ROOT, "include"), "-shared", "-o", lib_path, *src, "-lm"])
    import oracle_backend
    from maniskill_amd import _native as N
    from maniskill_amd.envs import registered as _registry

    oracle_backend._lib = N.NativeLib(lib_path, "orc_")          # the instrumented build instead of oracle/liborc.so
    raw = C.CDLL(lib_path)
    raw.orc_stats_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    env = _registry()[args.env](num_envs=args.envs, px_factory=lambda t, n, c: oracle_backend.OraclePhysxSystem(t, n, c))
    env.reset(seed=2022)
    gen = torch.Generator().manual_seed(0)
    buf = (C.c_longlong * 16)()
    raw.orc_stats_read(buf, 1)
    sub = env._sim_steps_per_control
    print(f"{args.env}, {args.envs} envs; per env and substep:")
    print(f"{'control steps':>14} {'pairs':>6} {'plane':>6} {'boxbox':>6} {'sat hit':>7} {'hull':>6} {'obb out':>7} {'gjk':>5} {'gjk it':>6} {'epa/k':>5} {'epa it':>6} {'epa deg':>7} {'manif':>6} {'points':>6}")
    chunk = max(args.steps // 5, 1)
    counts = []
    for k in range(args.steps):
        env.step(2 * torch.rand(args.envs, env.action_dim, generator=gen) - 1)
        counts.append(env.px.get_env_contact_counts().copy())
        if (k + 1) % chunk == 0:
            raw.orc_stats_read(buf, 1)
            g, gi, mo, e, ei, ed, oc, ot, pl, bb, sat, mf, pts, vis = list(buf)[:14]
            per = args.envs * sub * chunk
            print(f"{k + 1 - chunk:>6}..{k + 1:<6} {vis / per:>6.1f} {pl / per:>6.2f} {bb / per:>6.2f} {sat / per:>7.2f} {ot / per:>6.2f} {oc / per:>7.2f} {g / per:>5.2f} "
                  f"{gi / max(g, 1):>6.2f} {1000 * e / per:>5.2f} {ei / max(e, 1):>6.2f} {ed / per:>7.2f} {mf / per:>6.2f} {pts / per:>6.2f}")


    return counts


if __name__ == "__main__":
    import numpy as np
    c = np.stack(main())
    print("contact points per env (last substep of each control step): mean %.2f, median %d, p90 %d, p99 %d, max %d; batch maximum per step: mean %.1f"
          % (c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), c.max(axis=1).mean()))
