"""world_size-2 gloo test of the sharded path (CPU, oracle backend): the gathered observations of two
ranks equal the single-process run of the same global env set."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, steps, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle_backend import OraclePhysxSystem
    from maniskill_amd.dist import make_sharded_pick_cube

    env, gather, r, w = make_sharded_pick_cube(total, device_type="cpu",
                                               px_factory=lambda tpl, n, cfg: OraclePhysxSystem(tpl, n, cfg))
    obs, _ = env.reset(seed=2022)
    gen = torch.Generator().manual_seed(0)
    from maniskill_amd.dist import ObservationGather
    pipe = ObservationGather(env.num_envs, env.obs_dim, w, env.device)   # its own buffers: the two forms are not mixed on one object
    out, piped = None, []
    for _ in range(steps):
        a = torch.rand(total, 8, generator=gen) * 2 - 1      # same global action stream on every rank
        n = env.num_envs
        o, rew, term, trunc, _ = env.step(a[r * n:(r + 1) * n])
        out = gather(o, rew, term, trunc)
        prev = pipe.pipelined(o, rew, term, trunc)         # the overlapped form: hands back the previous step's gather
        piped.append(None if prev is None else [t.clone() for t in prev])
        piped_now = [t.clone() for t in out]
        if len(piped) > 1:
            assert all(torch.equal(x, y) for x, y in zip(piped[-1], last_sync))
        last_sync = piped_now
    final = pipe.flush()
    assert piped[0] is None and all(torch.equal(x, y) for x, y in zip(final, out)) and pipe.flush() is None
    if rank == 0:
        q.put([t.clone().numpy() for t in out])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(built):
    from oracle_backend import OraclePhysxSystem
    from maniskill_amd.envs.pick_cube import PickCubeEnv

    total, steps = 8, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    env = PickCubeEnv(num_envs=total, px_factory=lambda tpl, n, cfg: OraclePhysxSystem(tpl, n, cfg))
    env.reset(seed=2022)
    gen = torch.Generator().manual_seed(0)
    for _ in range(steps):
        o, rew, term, trunc, _ = env.step(torch.rand(total, 8, generator=gen) * 2 - 1)
    assert (got[0] == o.numpy()).all()
    assert (got[1] == rew.numpy()).all()
    assert (got[2] == term.numpy()).all() and (got[3] == trunc.numpy()).all()


def test_shard_range_covers_everything():
    from maniskill_amd.dist import shard_range

    for total, world in ((4096, 8), (4096, 2), (10, 3)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
