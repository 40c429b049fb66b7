"""Partial resets of the fused envs without the host on the step path (include/msk_physx.h: msk_reset_masked).

``ManiSkillVectorEnv.step`` resets the sub-scenes that finished inside the same step (mani_skill/vector/wrappers/gymnasium.py:164-176); once episodes drift out of phase
that happens at almost every step.  The host-side reset of these envs -- the reference's structure: ``BaseEnv.reset`` -> ``_clear_sim_state``,
``_initialize_episode`` with a numpy RNG per sub-scene, ``controller.reset``, masked writes through the torch views, ``_gpu_apply_all``, kinematics, ``_gpu_fetch_all``
(envs/sapien_env.py:857-978,1023-1036; utils/scene_builder/table/scene_builder.py:67-103) -- is ~30 torch launches, ~10 uploads and two waits for the device per call,
0.3-0.7 ms whatever the number of envs it names (profiles/r05_soak_where_after_pinned_uploads.log).

Here the SAME reset code produces, ahead of time and off the step path, the rows it would write -- an *episode image* per (sub-scene, episode number): the env's
own ``reset`` is run on a host-memory shadow of the env (same class, same seeds: ``2022 + global index`` main seeds, episode seed = f(main seed, episode counter),
envs/utils/randomization/batched_rng.py:13-70 in spirit), the entries it writes are found by probing with NaNs once, and a ring of ``slots`` images per sub-scene is
kept on the device.  A reset is then one kernel over a device-side mask: the named envs take their next image (fetch of the env, the image's entries, the apply of the
env: bit for bit what the host path leaves -- ``tests/test_device_reset.py``), their episode counters advance on the device.  Every ``slots // 2`` resets the counters
are copied to pinned memory behind the stream (no wait) and a worker thread refills what was consumed -- the shadow's reset on ONE intra-op thread (its ~40 torch
ops over a few thousand rows cost 4 ms that way and 200 ms through an OpenMP team per op: profiles/r06_device_reset_refresh.log), uploaded on a side stream into
slots nothing reads (the slots of episodes already consumed) -- so the step path only launches; the next refill first joins the one before it."""
from __future__ import annotations

import atexit
import ctypes as C
import threading
import weakref

import numpy as np
import torch

RIGID, QPOS, QVEL, TQPOS, TQVEL = 0, 1, 2, 3, 4      # include/msk_physx.h: MSK_RESET_*


_LIVE = weakref.WeakSet()      # the DeviceReset objects of this process (their workers are stopped before the interpreter goes)


def _stop_workers():
    """at exit: a daemon worker still inside torch when the interpreter finalises is unwound by ``pthread_exit`` through C++ frames -- ``terminate called without an
    active exception`` and a core instead of exit code 0 (seen once in four default bench runs: gpurun_out/r06_final3/bench_n1_default.err).  Builds in flight are
    cancelled, refills joined; errors at this point have nobody to go to."""
    for dr in list(_LIVE):
        try:
            dr.close()
        except BaseException:          # noqa: BLE001
            pass


atexit.register(_stop_workers)


class _NullPx:
    """what a shadow env's reset calls on its physics system: nothing happens there (the rows it wrote ARE the product)"""

    host_memory = True

    def __init__(self, px):
        self.lib, self.device = px.lib, torch.device("cpu")
        self.num_envs, self.bodies_per_env, self.timestep = px.num_envs, px.bodies_per_env, px.timestep

    def gpu_apply_all(self): pass
    def gpu_update_articulation_kinematics(self): pass
    def gpu_fetch_all(self): pass
    def _apply(self, mask): pass
    def _fetch(self, mask): pass


class DeviceReset:
    """``env.reset_mask(done)`` for a fused env of maniskill_amd.envs (PickCubeEnv and its subclasses, PushTEnv)."""

    def __init__(self, env, slots: int = 64, threaded=None):
        # every tensor of this object and of the shadow env is an ordinary tensor, whatever mode the caller is in: the refill thread is not in the caller's
        # torch.inference_mode() (thread-local), and an inference tensor refuses in-place writes from outside it
        with torch.inference_mode(False):
            self._init(env, slots, threaded)

    def _init(self, env, slots, threaded):
        if getattr(env, "control_mode", "pd_joint_delta_pos").startswith("pd_ee") and "target" in env.control_mode:
            raise RuntimeError("controllers that keep an end-effector target re-read the link frames inside reset(): host-side resets only")
        if not hasattr(env.px.lib, "reset_masked"):
            raise RuntimeError("this library has no msk_reset_masked")
        self.env, self.px, self.slots = env, env.px, int(slots)
        self.n = env.num_envs
        dev = env.device
        self.dev = dev
        self._shadow = self._make_shadow()
        self._probe_entries()
        self.nent = len(self.codes)
        self.ent = torch.as_tensor(self.codes, dtype=torch.int32, device=dev)
        self.image = torch.zeros(self.n, self.slots, self.nent, dtype=torch.float32, device=dev)
        self.episode = torch.zeros(self.n, dtype=torch.int32, device=dev)        # next episode number per env (the device's copy is the one that counts)
        self.mask = torch.zeros(self.n, dtype=torch.uint8, device=dev)
        self.filled = np.zeros(self.n, dtype=np.int64)                            # episodes [.., filled) are in the ring
        self.since_refresh = 0
        self.resets = self.refreshes = self.images_made = 0
        self._stage = None
        # refills off the step path: a worker thread + a side stream (None: where there is a device to overlap with)
        self.threaded = (dev.type == "cuda") if threaded is None else bool(threaded)
        self._job = None                  # the refill in flight: (thread, [exception])
        self._side = torch.cuda.Stream(dev) if dev.type == "cuda" else None
        self._uploaded = None             # event on the side stream: the last refill's uploads
        self._snap = None                 # pinned host copy of the counters + its event
        self._rebuild_job = None          # a whole-ring build in flight: (thread, [exception], cancel flag, sub-scenes, their first episodes)
        self.rebuilds = 0
        self._unbuilt = None              # sub-scenes whose ring build close() cancelled
        _LIVE.add(self)
        self.rebuild(np.arange(self.n))

    # ---------------------------------------------------------------------------------------------------------------- the shadow env
    def _make_shadow(self):
        env = self.env
        sh = object.__new__(type(env))
        d = dict(env.__dict__)
        cpu = torch.device("cpu")
        for k, v in list(d.items()):          # the env's small device constants (poses, rest positions): host copies
            if isinstance(v, torch.Tensor) and v.device.type != "cpu" and v.numel() <= 64 * max(self.n, 1):
                d[k] = v.detach().cpu().clone()
        sh.__dict__ = d
        n, nb = self.n, env.px.bodies_per_env
        sh.device = cpu
        sh.px = _NullPx(env.px)
        sh._rbd = torch.zeros(n, nb, 13)
        for name in ("_qpos", "_qvel", "_target_qpos_buf", "_target_qvel_buf"):
            t = getattr(env, name, None)
            if t is not None:
                setattr(sh, name, torch.zeros(tuple(t.shape)))
        if isinstance(getattr(env, "_target_qpos", None), torch.Tensor):
            sh._target_qpos = torch.zeros(tuple(env._target_qpos.shape))
        sh._elapsed_steps = torch.zeros(n, dtype=torch.int32)
        sh._offsets = env._offsets.detach().cpu().clone()
        from .pick_cube import BatchedRNG
        sh._rng = BatchedRNG(np.zeros(n, dtype=np.uint64))
        sh._episode_count = np.zeros(n, dtype=np.uint64)
        sh._main_seeds = env._main_seeds                   # (shared: a reseeded env is reseeded here too)
        sh._stage = None
        sh._buffers_stale = False
        sh._step_graph = None
        sh._dev_reset = None
        sh.device_reset = False                            # (the shadow's own resets are the host's, by definition)
        sh.camera, sh.cameras = None, {}
        sh.fused = True                                    # (whatever the env's mode: the shadow's reset ends in the stub below, not in task code)
        sh._fused_observe = lambda advance: (None, None, None, None, {})
        return sh

    def _buffers(self, sh):
        """(code base, tensor viewed [n, words per env]) of the shadow's sapien-style buffers"""
        n = self.n
        out = [(RIGID, sh._rbd.view(n, -1))]
        pitch = sh._qpos.shape[1]
        for which, name in ((QPOS, "_qpos"), (QVEL, "_qvel"), (TQPOS, "_target_qpos_buf"), (TQVEL, "_target_qvel_buf")):
            t = getattr(sh, name, None)
            if t is not None:
                assert t.shape[1] == pitch
                out.append((which, t.view(n, -1)))
        return out

    def _run(self, idx: np.ndarray, episodes: np.ndarray):
        """the env's own reset for sub-scenes ``idx`` at episode numbers ``episodes``, on the shadow"""
        sh = self._shadow
        sh._episode_count[idx] = episodes.astype(np.uint64)
        k = torch.get_num_threads()       # (OpenMP's thread count is the calling thread's own setting: the worker's 1 does not reach the caller's ops)
        if k != 1:
            torch.set_num_threads(1)
        try:
            sh.reset(seed=None, options=dict(env_idx=torch.as_tensor(idx, dtype=torch.long)))
        finally:
            if k != 1:
                torch.set_num_threads(k)

    def _probe_entries(self):
        """which words of the buffers does a reset write?  Two runs over NaN-filled buffers (the same episode twice would hide nothing; two different episodes make sure
        a word is listed even when one of them happens to write a NaN-free value... every written word is non-NaN in both)"""
        sh = self._shadow
        idx = np.arange(self.n)
        written = None
        for ep in (0, 1):
            for _, t in self._buffers(sh):
                t.fill_(float("nan"))
            self._run(idx, np.full(self.n, ep))
            w = [(~torch.isnan(t)).all(dim=0) for _, t in self._buffers(sh)]       # written in EVERY env
            anyw = [(~torch.isnan(t)).any(dim=0) for _, t in self._buffers(sh)]
            for a, b in zip(w, anyw):
                if not torch.equal(a, b):
                    raise RuntimeError("a reset writes different words in different sub-scenes: host-side resets only")
            written = w if written is None else [a | b for a, b in zip(written, w)]
        self.codes, self._gather = [], []
        for (which, t), w in zip(self._buffers(sh), written):
            words = torch.nonzero(w).reshape(-1)
            self._gather.append(words)
            self.codes += [(which << 24) | int(v) for v in words]
        for _, t in self._buffers(sh):
            t.zero_()

    def _rows(self, idx: np.ndarray) -> torch.Tensor:
        sh = self._shadow
        ii = torch.as_tensor(idx, dtype=torch.long)
        return torch.cat([t[ii][:, words] for (_, t), words in zip(self._buffers(sh), self._gather)], dim=1)

    # ---------------------------------------------------------------------------------------------------------------- the ring
    def _upload(self, idx: np.ndarray, slot: np.ndarray, rows: torch.Tensor):
        if self.dev.type != "cuda":
            self.image[torch.as_tensor(idx, dtype=torch.long), torch.as_tensor(slot, dtype=torch.long)] = rows
            return
        if self._stage is None:
            self._stage = [torch.empty(self.n, self.nent).pin_memory(), torch.empty(self.n, 2, dtype=torch.int64).pin_memory(), torch.cuda.Event()]
        host, hidx, ev = self._stage
        ev.synchronize()
        k = len(idx)
        host[:k] = rows
        hidx[:k, 0] = torch.as_tensor(idx)
        hidx[:k, 1] = torch.as_tensor(slot)
        with torch.cuda.stream(self._side if self._job_thread() else torch.cuda.current_stream(self.dev)):
            d = host[:k].to(self.dev, non_blocking=True)
            di = hidx[:k].to(self.dev, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.dev))
            self.image[di[:, 0], di[:, 1]] = d

    def _fill(self, idx: np.ndarray, first: np.ndarray, upto: np.ndarray, cancel=None):
        """images of episodes first[i] .. upto[i] - 1 of sub-scene idx[i]"""
        j = 0
        while True:
            sel = first + j < upto
            if not sel.any() or (cancel is not None and cancel.is_set()):
                break
            ii, ep = idx[sel], (first + j)[sel]
            self._run(ii, ep)
            self._upload(ii, ep % self.slots, self._rows(ii))
            self.images_made += len(ii)
            j += 1

    def rebuild(self, idx: np.ndarray):
        """sub-scenes whose seeds or episode counters the host set (a seeded reset): their ring starts over at the host's counter.  A whole ring is ``slots`` runs of the
        env's reset over these sub-scenes -- 0.4 to 3 s for 4096 x 64 on the GPU boxes' host cores (round 6's first default bench line: 2.9 s inside every seeded
        reset).  Threaded: a worker builds it while the caller goes on; until it is done ``ready()`` is False and resets are the host's (the env's own ``reset``:
        correct, slower), then the device's counters take the host's over and the device path resumes.  A new seeded reset cancels a build in flight."""
        self.join()
        idx = np.asarray(idx, dtype=np.int64)
        pend = self._cancel_rebuild()
        if pend is not None:
            idx = np.union1d(idx, pend)
        if self._unbuilt is not None:
            idx, self._unbuilt = np.union1d(idx, self._unbuilt), None
        if len(idx) == 0:
            return
        ep0 = self.env._episode_count[idx].astype(np.int64)
        if not self.threaded:
            self.episode[torch.as_tensor(idx, dtype=torch.long, device=self.dev)] = torch.as_tensor(ep0, dtype=torch.int32, device=self.dev)
            self._fill(idx, ep0, ep0 + self.slots)
            self.filled[idx] = ep0 + self.slots
            return
        cancel, err = threading.Event(), []

        def work():
            try:
                torch.set_num_threads(1)
                self._fill(idx, ep0, ep0 + self.slots, cancel)
                if self._side is not None:
                    self._uploaded = self._side.record_event()
            except BaseException as e:          # noqa: BLE001 -- re-raised on the caller's thread when the build is taken over
                err.append(e)

        t = threading.Thread(target=work, name="msk-device-reset-rebuild", daemon=True)
        self._rebuild_job = (t, err, cancel, idx, ep0)
        self.rebuilds += 1
        t.start()

    def _cancel_rebuild(self):
        """stops a ring build in flight; -> the sub-scenes it was building (they are not built), or None"""
        job = self._rebuild_job
        if job is None:
            return None
        job[2].set()
        job[0].join()
        self._rebuild_job = None
        return job[3]

    def ready(self) -> bool:
        """the device path may be used: no ring build is in flight (one that has just finished is taken over here)"""
        job = self._rebuild_job
        if job is None:
            return self._unbuilt is None
        if job[0].is_alive():
            return False
        self._finish_rebuild()
        return self._unbuilt is None

    def wait_ready(self):
        job = self._rebuild_job
        if job is not None:
            job[0].join()
            self._finish_rebuild()

    def _finish_rebuild(self):
        t, err, cancel, idx, ep0 = self._rebuild_job
        t.join()
        self._rebuild_job = None
        if err:
            raise err[0]
        if self._uploaded is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._uploaded)
            self._uploaded = None
        self.filled[idx] = ep0 + self.slots
        cnt = self.env._episode_count.astype(np.int64)          # the host reset the envs meanwhile: its counters are the ones that count
        self.episode.copy_(torch.as_tensor(cnt.astype(np.int32)).to(self.dev))
        low = np.nonzero(self.filled - cnt < self.slots // 2 + 1)[0]
        if len(low):                                             # (a long build, many host-side resets meanwhile: topped up before the device path goes on)
            self._fill(low, self.filled[low], cnt[low] + self.slots)
            self.filled[low] = cnt[low] + self.slots
        self.since_refresh = 0

    def pull_counts(self) -> np.ndarray:
        """the device's episode counters (one read-back); the host's copy follows"""
        self.join()
        if self._rebuild_job is not None and not self.ready():      # a ring build in flight: resets are the host's, so are the counters
            return self.env._episode_count.astype(np.int64)
        ep = self.episode.cpu().numpy().astype(np.int64)
        self.env._episode_count[:] = ep.astype(np.uint64)
        return ep

    def _refill(self, ep: np.ndarray):
        idx = np.nonzero(ep + self.slots > self.filled)[0]
        if len(idx):
            self._fill(idx, self.filled[idx], ep[idx] + self.slots)
            self.filled[idx] = ep[idx] + self.slots

    def refresh(self):
        """refill the ring up to ``slots`` episodes past the device's counters.  Threaded: the counters are snapshot behind the stream and a worker does the rest;
        what it writes are the slots of episodes [filled, snapshot + slots) = the slots of episodes consumed before the snapshot, which no later reset reads (an
        env is at most ``slots // 2`` episodes past the previous snapshot, so its reads stay inside [snapshot, filled)); the refill after this one joins it first."""
        self.join()
        self.since_refresh = 0
        self.refreshes += 1
        if not self.threaded:
            self._refill(self.pull_counts())
            return
        if self.dev.type == "cuda":
            if self._snap is None:
                self._snap = (torch.empty(self.n, dtype=torch.int32).pin_memory(), torch.cuda.Event())
            host, ev = self._snap
            host.copy_(self.episode, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.dev))
        else:
            host, ev = self.episode.clone(), None
        err = []

        def work():
            try:
                torch.set_num_threads(1)
                if ev is not None:
                    ev.synchronize()
                ep = host.numpy().astype(np.int64)
                self.env._episode_count[:] = ep.astype(np.uint64)
                self._refill(ep)
                if self._side is not None:
                    self._uploaded = self._side.record_event()
            except BaseException as e:          # noqa: BLE001 -- re-raised on the caller's thread at the join
                err.append(e)

        t = threading.Thread(target=work, name="msk-device-reset-refill", daemon=True)
        self._job = (t, err)
        t.start()

    def close(self):
        """no worker of this object runs after this: a refill in flight is joined, a ring build in flight cancelled -- its sub-scenes stay the host's (``ready()`` is
        False) until the next seeded reset builds their ring again"""
        job = self._job
        if job is not None and threading.current_thread() is not job[0]:
            job[0].join()
        pend = self._cancel_rebuild()
        if pend is not None and len(pend):
            self._unbuilt = pend if self._unbuilt is None else np.union1d(self._unbuilt, pend)

    def _job_thread(self) -> bool:
        cur = threading.current_thread()
        return (self._job is not None and cur is self._job[0]) or (self._rebuild_job is not None and cur is self._rebuild_job[0])

    def join(self):
        """the refill in flight has finished, and the caller's stream is behind its uploads"""
        job = self._job
        if job is None or threading.current_thread() is job[0]:
            return
        job[0].join()
        self._job = None
        if self._uploaded is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._uploaded)
            self._uploaded = None
        if job[1]:
            raise job[1][0]

    # ---------------------------------------------------------------------------------------------------------------- the reset
    def reset_mask(self, done: torch.Tensor):
        """the partial reset of the envs ``done`` names (bool / uint8 [num_envs], on the device); -> (obs, info) of ``env.reset``"""
        env, px, L = self.env, self.px, self.px.lib
        if self.since_refresh >= self.slots // 2:      # an env advances at most one episode per reset: the ring cannot run dry in between
            self.refresh()
        self.since_refresh += 1
        self.resets += 1
        if done.dtype in (torch.bool, torch.uint8) and done.is_contiguous() and done.device == self.mask.device:
            mask = done                    # (a bool tensor is one byte per element, 0 / 1: the kernel reads it as it is)
        else:
            self.mask.copy_(done)
            mask = self.mask
        p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
        L.check(px.ctx, L.reset_masked(px.ctx, p(mask), p(self.image), self.slots, p(self.ent), self.nent, p(self.episode), p(env._elapsed_steps), px._stream()),
                "reset_masked")
        tq = getattr(env, "_target_qpos", None)
        if isinstance(tq, torch.Tensor) and getattr(env, "control_mode", "") != "pd_joint_delta_pos":
            # controller.reset(): the torch-side controllers' own copy of the targets (pd_joint_pos.py:54-69); pd_joint_delta_pos rewrites all of it at every action
            self._controller_targets(mask)
        if getattr(env, "fused", False):      # the first observation of the new episodes: the task kernel (it computes the link frames of the new state itself)
            env._buffers_stale = True
            obs, _, _, _, info = env._fused_observe(False)
            return obs, info
        px.gpu_update_articulation_kinematics()      # the torch task code reads the sapien buffers: the tail of the host-side reset
        px.gpu_fetch_all()
        env._buffers_stale = False
        info = env.get_info()
        obs = env.get_obs(info)
        return (env._with_sensor_data(obs) if hasattr(env, "_with_sensor_data") else obs), info

    def _controller_targets(self, done):
        """env._target_qpos rows of the reset envs = their new qpos: read from the image (no fetch): only the torch-side controllers of the less common control modes
        read it, the fused controller kernels read the simulator's own targets"""
        env = self.env
        if not hasattr(self, "_q_cols"):
            nq = env._target_qpos.shape[1]
            pos = {c: t for t, c in enumerate(self.codes)}
            try:
                self._q_cols = torch.as_tensor([pos[(QPOS << 24) | j] for j in range(nq)], dtype=torch.long, device=self.dev)
            except KeyError:
                self._q_cols = None
        if self._q_cols is None:
            return
        slot = ((self.episode - 1) % self.slots).long()      # (the kernel has advanced the counters of the reset envs)
        rows = self.image[torch.arange(self.n, device=self.dev), slot][:, self._q_cols]
        env._target_qpos.copy_(torch.where(done.bool()[:, None], rows, env._target_qpos))


class DeviceResetMixin:
    """``reset`` of the fused envs: without a seed and without a state to restore it is the device-side reset (all envs, or the ``env_idx`` named); ``reset_mask``
    is the form ManiSkillVectorEnv's same-step auto reset uses (a device-side mask, no index list, no wait).  ``device_reset``: None = on for the fused envs on a
    GPU, True / False = as said (the CPU suite asks for it on its host-memory backends)."""

    _dev_reset = None
    device_reset = None

    def _device_reset_wanted(self) -> bool:
        want = self.device_reset
        if want is None:
            want = bool(getattr(self, "fused", False)) and not getattr(self.px, "host_memory", False)
        if want and self._dev_reset is None:
            try:
                self._dev_reset = DeviceReset(self, slots=int(getattr(self, "device_reset_slots", 64)), threaded=getattr(self, "device_reset_threaded", None))
            except RuntimeError:
                self.device_reset = want = False
        return bool(want) and self._dev_reset.ready()

    def _reset_on_device(self, seed, options):
        """-> (obs, info), or None when this reset is the host's (a seed, a state to restore, device resets switched off)"""
        if seed is not None or (options and any(k != "env_idx" for k in options)) or not getattr(self, "_constructed", False) or not self._device_reset_wanted():
            return None
        if options and "env_idx" in options:
            idx = torch.as_tensor(options["env_idx"], device=self.device, dtype=torch.long)
            mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            mask[idx] = 1
        else:
            mask = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
        return self._dev_reset.reset_mask(mask)

    def reset_mask(self, done: torch.Tensor):
        """partial reset of the envs a device-side mask names (``terminated | truncated`` of a step) -> (obs, info)"""
        if self._device_reset_wanted():
            return self._dev_reset.reset_mask(done)
        rows = torch.nonzero(done.reshape(-1)).reshape(-1)
        return self.reset(options=dict(env_idx=rows))

    def _host_reset_begins(self):
        if self._dev_reset is not None:
            self._dev_reset.pull_counts()

    def _host_reset_ends(self, idx_np, reseeded: bool = True):
        """reseeded: the reset set these sub-scenes' seeds / episode counters (``reset(seed=...)``): their prepared episodes are void.  A host-side reset without a
        seed only consumed one episode each -- the ring stays, the device's counters follow when the device path resumes"""
        dr = self._dev_reset
        if dr is None:
            return
        if reseeded:
            dr.rebuild(idx_np)
        elif dr._rebuild_job is None:      # (device path in use and the host reset some envs itself: options other than env_idx)
            idx = torch.as_tensor(np.asarray(idx_np), dtype=torch.long, device=dr.dev)
            dr.episode[idx] = torch.as_tensor(self._episode_count[idx_np].astype(np.int32)).to(dr.dev)
