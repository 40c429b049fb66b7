"""Multi-GPU sharding of independent sub-scenes: one process per GPU, RCCL all-gather of observations only.

The reference runtime is single-device (mani_skill/envs/sapien_env.py:95-100,240-245); envs never
interact (sub-scene spacing, structs/types.py:77-79), so the path shards trivially:
rank r owns the contiguous global env range [r*n, (r+1)*n) with seeds 2022 + global index, steps
it with NO data-path collective, and publishes (obs | reward | terminated | truncated) through ONE
``all_gather_into_tensor`` per control step.  Messages are ~45 floats/env (PickCube: 512 envs/rank
-> 92 KB), i.e. latency-bound on xGMI: one fused collective, no bucketing.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(device_type: str = "cuda") -> Tuple[int, int, int]:
    """Returns (rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1.
    backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU test-suite."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if device_type == "cuda" else "gloo"
        if device_type == "cuda":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env range of `rank` (strong scaling: total fixed; sizes differ by at most one)."""
    base, rem = divmod(total_envs, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class ObservationGather:
    """Packs (obs, reward, terminated, truncated) of the local shard into one tensor and all-gathers it."""

    def __init__(self, local_envs: int, obs_dim: int, world: int, device, equal_shards: bool = True):
        self.world, self.local_envs, self.obs_dim = world, local_envs, obs_dim
        self.width = obs_dim + 3
        self.send = torch.empty(local_envs, self.width, dtype=torch.float32, device=device)
        self.recv = torch.empty(world * local_envs, self.width, dtype=torch.float32, device=device) if world > 1 else self.send
        assert equal_shards, "all_gather_into_tensor needs equal shard sizes (pad the env count to a multiple of world)"
        # second buffer pair + the collective in flight, for pipelined()
        self._send2 = torch.empty_like(self.send) if world > 1 else None
        self._recv2 = torch.empty_like(self.recv) if world > 1 else None
        self._inflight = None      # (work handle, recv buffer) of the newest collective
        self._parity = 0
        self._last = None          # world 1: the newest step's tensors

    def _pack(self, s, obs, reward, terminated, truncated):
        s[:, : self.obs_dim] = obs
        s[:, self.obs_dim] = reward
        s[:, self.obs_dim + 1] = terminated.to(torch.float32)
        s[:, self.obs_dim + 2] = truncated.to(torch.float32)

    def _unpack(self, r):
        return r[:, : self.obs_dim], r[:, self.obs_dim], r[:, self.obs_dim + 1] > 0.5, r[:, self.obs_dim + 2] > 0.5

    def pipelined(self, obs, reward, terminated, truncated):
        """One all-gather per control step, overlapped with the next step's physics: the collective of step k is issued
        on RCCL's own stream as soon as the step's outputs exist and is only waited for when step k + 1 hands in its
        outputs -- the call returns the gathered tensors of the PREVIOUS step (None the first time); ``flush()`` returns the
        newest.  Nothing on the physics stream waits for xGMI; a rank that is ahead runs at most one step ahead."""
        if self.world == 1:
            prev, self._last = self._last, (obs, reward, terminated, truncated)
            return prev
        prev = self.flush()
        send, recv = (self.send, self.recv) if self._parity == 0 else (self._send2, self._recv2)
        self._parity ^= 1
        self._pack(send, obs, reward, terminated, truncated)
        self._inflight = (dist.all_gather_into_tensor(recv, send, async_op=True), recv)
        return prev

    def flush(self):
        """Waits for the collective in flight and returns its tensors (None if there is none)."""
        if self.world == 1:
            last, self._last = self._last, None
            return last
        if self._inflight is None:
            return None
        work, recv = self._inflight
        self._inflight = None
        work.wait()
        return self._unpack(recv)

    def __call__(self, obs, reward, terminated, truncated):
        if self.world == 1:   # one shard: nothing to exchange, nothing to pack
            return obs, reward, terminated, truncated
        self._pack(self.send, obs, reward, terminated, truncated)
        dist.all_gather_into_tensor(self.recv, self.send)
        return self._unpack(self.recv)


def make_sharded_env(env_id: str, total_envs: int, device_type: str = "cuda", px_factory=None, **kw):
    """One env shard per process (env_id: "PickCube-v1", "PushT-v1" or "PegInsertionSide-v1").
    Returns (env, gather, rank, world)."""
    from .envs.peg_insertion_side import PegInsertionSideEnv
    from .envs.pick_cube import PickCubeEnv
    from .envs.push_t import PushTEnv

    cls = {"PickCube-v1": PickCubeEnv, "PushT-v1": PushTEnv, "PegInsertionSide-v1": PegInsertionSideEnv}[env_id]
    rank, world, local = init_distributed(device_type)
    start, count = shard_range(total_envs, rank, world)
    assert total_envs % world == 0, "num_envs must divide evenly over the ranks"
    device: Optional[str] = f"cuda:{local}" if device_type == "cuda" else None
    env = cls(num_envs=count, device=device, env_index_offset=start, total_envs=total_envs, px_factory=px_factory, **kw)
    obs_dim = env.obs_dim if kw.get("obs_mode", "state") == "state" else (21 if env_id == "PushT-v1" else env.obs_dim)
    gather = ObservationGather(count, obs_dim, world, env.device)
    return env, gather, rank, world


class ShardedGymEnv:
    """One shard of a registered ManiSkill task over the sapien shim (the drop-in path: the reference's own BaseEnv / controllers / task
    code).  The reference runtime is single-device (mani_skill/envs/sapien_env.py:95-100); here every rank builds `gym.make(env_id,
    num_envs=total/world)` for its contiguous range of the global env set, seeds env i of the GLOBAL set with seed + i (ManiSkill takes a
    seed per env: BaseEnv.reset(seed=[...])) and lays its sub-scenes out on the global grid (shim: set_shard), so a rollout does not
    depend on the number of ranks.  `step` returns what the wrapped env returns; `gather` all-gathers the flat state observation."""

    def __init__(self, env, start, count, total, gather, rank, world):
        self.env, self.start, self.num_envs, self.total_envs = env, start, count, total
        self.gather, self.rank, self.world = gather, rank, world
        self.unwrapped = env.unwrapped
        self.device = env.unwrapped.device
        self.action_space = env.action_space

    def reset(self, seed=None, options=None):
        """`seed=None` (a plain reset, or a partial one through options["env_idx"]) is forwarded as None: the env's RNG streams go on, as
        BaseEnv.reset(seed=None) means (sapien_env.py:907-918) -- successive episodes differ.  An int seeds env i of the GLOBAL set with
        seed + i; a list is taken as the seeds of this shard's envs.  The reference replaces ALL episode RNGs whenever it gets a seed list
        (`_set_episode_rng`, sapien_env.py:999-1016), so a seeded partial reset re-seeds the whole shard there too: it gets the full list."""
        if seed is None:
            return self._reset(None, options)
        if isinstance(seed, (list, tuple)) or hasattr(seed, "__len__"):
            seeds = [int(s) for s in seed]
            assert len(seeds) == self.num_envs, f"{len(seeds)} seeds for a shard of {self.num_envs} envs"
        else:
            seeds = [int(seed) + self.start + i for i in range(self.num_envs)]
        return self._reset(seeds, options)

    def _reset(self, seed, options):
        # the scene is rebuilt -- a new PhysxSystem, under this shard again -- when the caller asks for it or when BaseEnv.reset reconfigures on its own
        # (reconfiguration_freq != 0, sapien_env.py:898: the default of some tasks when a rank holds one env)
        if (options and options.get("reconfigure")) or getattr(self.unwrapped, "reconfiguration_freq", 0):
            from sapien import _system
            _system.set_shard(self.start, self.num_envs, self.total_envs)
            try:
                return self.env.reset(seed=seed, options=options)
            finally:
                _system.set_shard(0, None, None)
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        self.env.close()


def make_sharded_gym_env(env_id: str, total_envs: int, device_type: str = "cuda", reference_root: Optional[str] = None, backend=None,
                         accelerate: Optional[str] = None, **gym_kw):
    """Any registered ManiSkill task, sharded: -> ShardedGymEnv.  `backend`: a NativeLib to run the shim on instead of libmsk_physx.so
    (the test-suite hands in the CPU oracle); `reference_root`: where the `mani_skill` package lives (default: importable already, or
    MANISKILL_ROOT); `accelerate`: "control" | "task" | "graph" -- maniskill_amd.fused_step.accelerate on the shard's env (same results, fewer
    launches; raises fused_step.Unsupported where the task's step is not restated)."""
    rank, world, local = init_distributed(device_type)
    start, count = shard_range(total_envs, rank, world)
    assert total_envs % world == 0, "num_envs must divide evenly over the ranks"
    from . import shim
    shim.install(reference_root or os.environ.get("MANISKILL_ROOT"))
    import sapien.physx as physx
    from sapien import _system
    if backend is not None:
        physx._set_backend(backend, host_memory=True)
    import gymnasium as gym
    import mani_skill.envs  # noqa: F401  (registers the tasks)
    if device_type == "cuda":
        gym_kw.setdefault("sim_backend", f"physx_cuda:{local}" if local else "physx_cuda")
    # the shard is in force only while THIS env builds its scene (every PhysxSystem keeps the one it was created under): an unsharded
    # gym.make later in the same process -- an eval env -- lays its sub-scenes out on its own local grid
    _system.set_shard(start, count, total_envs)
    try:
        env = gym.make(env_id, num_envs=count, **gym_kw)
        if start > 0:   # BaseEnv.__init__ builds the scene under the seeds 2022 + LOCAL index (sapien_env.py:327); tasks that draw per-env assets
            #             at build time (a cabinet per sub-scene) get the ones of their GLOBAL index by reconfiguring once under those seeds
            env.reset(seed=[2022 + start + i for i in range(count)], options=dict(reconfigure=True))
    finally:
        _system.set_shard(0, None, None)
    if accelerate:
        from .fused_step import accelerate as _accelerate
        _accelerate(env, graph=accelerate == "graph", task=accelerate != "control")
    obs_dim = int(env.observation_space.shape[-1]) if getattr(env.observation_space, "shape", None) else 0
    gather = ObservationGather(count, obs_dim, world, env.unwrapped.device) if obs_dim else None
    return ShardedGymEnv(env, start, count, total_envs, gather, rank, world)


def make_sharded_pick_cube(total_envs: int, device_type: str = "cuda", px_factory=None, **kw):
    """make_sharded_env("PickCube-v1", ...): kept for callers of round 1."""
    return make_sharded_env("PickCube-v1", total_envs, device_type=device_type, px_factory=px_factory, **kw)
