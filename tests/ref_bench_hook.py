"""Hook of bench.py's bootstrap self-test (MSK_BENCH_SELFTEST_HOOK): on the ranks bench.py started itself (gloo, no GPU), build env shards on the CPU checker and
drive bench's OWN rollout code over them -- `timed_rollout` for a fused-host shard in the weak-scaling form, `dropin_sharded_run` for the reference's env of
BASELINE config 4 / 5's task.  Test infrastructure: the oracle is loaded here, never by bench.py."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(bench, args, rank, world):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    what = os.environ.get("MSK_BENCH_HOOK_CASE", "fused_weak")
    if what == "fused_weak":       # --scaling weak: --envs per rank; the fused host of --env over make_sharded_env, bench.timed_rollout, the MAX over ranks
        from oracle_backend import OraclePhysxSystem
        from maniskill_amd.dist import make_sharded_env
        env, gather, r, w = make_sharded_env(args.env, args.envs, device_type="cpu", px_factory=lambda t, n, c: OraclePhysxSystem(t, n, c))
        assert (r, w) == (rank, world) and env.num_envs * world == args.envs
        env.reset(seed=2022)
        n_local = env.num_envs

        def sync():
            if world > 1:
                dist.barrier()
        dt = bench.timed_rollout(env.step, lambda: 2 * torch.rand(n_local, env.action_dim) - 1, gather.pipelined, gather.flush, sync, args.steps, args.warmup)
        t = torch.tensor([dt], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        obs = gather(*env.step(torch.zeros(n_local, env.action_dim))[:4])[0]      # the gathered observation covers the GLOBAL env set
        if rank == 0:
            print(json.dumps({"case": what, "world": world, "total_envs": args.envs, "envs_per_rank": n_local, "scaling": args.scaling, "steps": args.steps,
                              "gathered_rows": int(obs.shape[0]), "seconds": float(t.item())}), flush=True)
        return 0
    # the drop-in path: the reference's own env per rank over the shim (on the checker), bench.dropin_sharded_run
    import ref_harness
    ref = ref_harness.find_reference()
    if ref is None:
        if rank == 0:
            print(json.dumps({"skipped": "no reference build"}), flush=True)
        return 0
    if args.env.startswith("OpenCabinet"):
        import subprocess
        assets = os.environ["MS_ASSET_DIR"]
        if rank == 0 and not os.path.isdir(os.path.join(assets, "data")):
            meta = os.path.join(ref, "mani_skill", "assets", "partnet_mobility", "meta")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_partnet.py"), "--out", assets, "--max-drawers", "2", "--ids-from",
                                   os.path.join(meta, "info_cabinet_drawer_train.json"), "--placeholder-ids-from",
                                   os.path.join(meta, "info_cabinet_door_train.json")], stdout=subprocess.DEVNULL)
        if world > 1:
            dist.barrier()
    ref_harness.setup("oracle")
    from oracle_backend import oracle_lib
    from maniskill_amd.dist import make_sharded_gym_env
    kw = {} if args.env.startswith("OpenCabinet") else dict(render_backend="none")
    acc = None if args.accelerate == "none" else ("task" if args.accelerate == "graph" else args.accelerate)      # (no HIP graphs on the checker)
    shard = make_sharded_gym_env(args.env, args.envs, device_type="cpu", reference_root=ref, backend=oracle_lib(), accelerate=acc, obs_mode=args.obs_mode, **kw)
    shard.reset(seed=2022)
    return bench.dropin_sharded_run(args, shard)
