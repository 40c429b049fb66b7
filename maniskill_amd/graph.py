"""One control step of an env captured as a HIP graph.

The torch-side task code (struct gathers, evaluate, observations, reward -- SURVEY.md §8a rows A3, A5-A7) is a few
hundred tiny kernels per step for the tasks without a fused task kernel; at 4096 envs the host cannot issue them as
fast as the GPU retires them, so the step is launch-bound.  Everything in the step is already enqueued on torch's
current stream (the C-ABI entry points take the stream as an argument and never synchronise), so the whole step --
controller, S substeps, fetch, contact queries, task code, the camera if there is one -- can be captured once into a
hipGraph and replayed with one launch.

Contract of a captured env step:
  * no host synchronisation and no host->device copy inside ``step`` (all the envs of this package comply for the
    control modes documented in DESIGN.md §6),
  * the action tensor is copied into a static input; the outputs are static tensors that the *next* replay
    overwrites, so ``__call__`` hands out clones of everything: obs (camera textures included), reward, flags and ``info``,
  * anything outside ``step`` (reset, set_state, per-env instance updates) stays eager and needs no re-capture: the
    graph holds pointers into the simulator's persistent state, not copies of it.
"""
from __future__ import annotations

import torch


_CONSTANTS = {}
CAPTURING = False   # True while a StepGraph warms up / captures: the step's outputs are snapshotted by the replay anyway (_clone_tree), so code
                    # inside the step hands out its static buffers (the camera planes) instead of cloning them a first time


def const(values, device, dtype=torch.float32):
    """A small device constant, uploaded once per (values, device, dtype).  Step-path code uses this instead of
    ``torch.tensor(..., device=...)``: a host->device copy per step is wasted work in eager mode and illegal inside a
    stream capture."""
    def freeze(v):
        return tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else float(v)
    key = (freeze(values), str(device), dtype)
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


def alloc_step_outputs(n: int, obs_dim: int, device, extra_floats: int = 0):
    """Fresh output tensors of one fused step -- observations [n, obs_dim] f32, reward [n] f32, flag bytes [n, 8] bool, elapsed steps [n]
    i32, optional extra [n, extra_floats] f32 -- carved out of ONE allocation: every tensor is contiguous on its own, and a step-graph
    replay snapshots all of them with a single copy (_clone_tree) instead of one launch each."""
    words = n * (obs_dim + 1 + 2 + 1 + extra_floats)
    pack = torch.empty(words, dtype=torch.float32, device=device)
    o = n * obs_dim
    obs = pack[:o].view(n, obs_dim)
    rew = pack[o:o + n]
    fl = pack[o + n:o + 3 * n].view(torch.uint8).view(n, 8).view(torch.bool)
    elapsed = pack[o + 3 * n:o + 4 * n].view(torch.int32)
    extra = pack[o + 4 * n:].view(n, extra_floats) if extra_floats else None
    return obs, rew, fl, elapsed, extra


def _tensors_of(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, dict):
        for v in x.values():
            _tensors_of(v, out)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _tensors_of(v, out)


def _clone_tree(x, _st=None):
    """Snapshots of every tensor a replay hands out.  Everything produced inside the capture lives in memory the next replay
    overwrites, so callers get copies made after the replay.  Tensors that share one STORAGE (alloc_step_outputs: observations, reward,
    flag columns and step counter are dtype-views of one allocation -- `_base` does not survive a dtype view, the storage does) are copied
    ONCE, as the whole storage, and re-viewed: each copy is a launch of its own behind the graph (round 4's trace showed ten of them per
    control step, 50 us of a 850 us step, where one does)."""
    if _st is None:      # first call: which storages are worth copying whole (their DISTINCT views cover at least half of them)
        ts = []
        _tensors_of(x, ts)
        cover, seen = {}, set()
        for t in ts:
            st = t.untyped_storage()
            key = (st.data_ptr(), t.storage_offset(), tuple(t.size()), tuple(t.stride()), t.dtype)   # the same slice handed out twice counts once
            if key in seen:
                continue
            seen.add(key)
            cover[st.data_ptr()] = min(cover.get(st.data_ptr(), 0) + t.numel() * t.element_size(), st.nbytes())
        _st = {"cover": cover, "clones": {}}
    if isinstance(x, torch.Tensor):
        st = x.untyped_storage()
        if st.nbytes() > 0 and 2 * _st["cover"].get(st.data_ptr(), 0) >= st.nbytes():
            c = _st["clones"].get(st.data_ptr())
            if c is None:       # the storage as bytes, copied once
                whole = torch.empty(0, dtype=torch.uint8, device=x.device).set_(st, 0, (st.nbytes(),), (1,))
                c = _st["clones"][st.data_ptr()] = whole.clone().untyped_storage()
            return torch.empty(0, dtype=x.dtype, device=x.device).set_(c, x.storage_offset(), x.size(), x.stride())
        return x.clone()
    if isinstance(x, dict):
        return {k: _clone_tree(v, _st) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_tree(v, _st) for v in x)
    return x


def probe_capture(step_fn, action, device) -> str:
    """Why a step cannot be captured: runs ``step_fn`` once more inside a (throw-away) stream capture and asks the HIP runtime after every torch operator
    and every C-ABI call whether the capture is still valid (``hipStreamIsCapturing``); returns the first one after which it is not, with the line of host
    code that issued it -- or "" when the capture survives.  A diagnostic for error messages: a failed capture otherwise only says
    "operation failed due to a previous error during capture"."""
    import ctypes
    import os
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    from . import _native
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return ""
    hip.hipStreamIsCapturing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    found = []

    def site():
        for f in reversed(traceback.extract_stack(limit=80)[:-2]):
            if "/torch/" not in f.filename and not f.filename.endswith(("maniskill_amd/graph.py", "maniskill_amd/fused_step.py")):
                return f"{os.path.basename(os.path.dirname(f.filename))}/{os.path.basename(f.filename)}:{f.lineno}"
        return "?"

    def alive(what):
        if found:
            return
        st = ctypes.c_int(0)
        rc = hip.hipStreamIsCapturing(ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream), ctypes.byref(st))
        if rc != 0 or st.value != 1:       # hipStreamCaptureStatusActive = 1; 2 = invalidated
            found.append(f"{what} @ {site()} (hipStreamIsCapturing: rc {rc}, status {st.value})")

    class Probe(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            alive(str(func))
            return out
    orig_check = _native.NativeLib.check

    def check(self, ctx, code, what):
        alive(f"{self.prefix}{what}")
        return orig_check(self, ctx, code, what)
    _native.NativeLib.check = check
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            alive("capture begin")
            with Probe():
                step_fn(action)
    except Exception as e:   # noqa: BLE001 -- the failure this probe is asked to explain
        if not found:
            found.append(f"no operator or C-ABI call was caught invalidating the capture; it ended with: {str(e).splitlines()[0][:200]}")
    finally:
        _native.NativeLib.check = orig_check
        torch.cuda.synchronize(device)
    return found[0] if found else ""


class StepGraph:
    """Captures ``step_fn(action) -> (obs, reward, terminated, truncated, info)`` of one env shard."""

    def __init__(self, step_fn, num_envs: int, action_dim: int, device, warmup: int = 2):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("StepGraph needs a GPU env: HIP graphs capture device work only")
        self.device = device
        self.action = torch.zeros(num_envs, action_dim, dtype=torch.float32, device=device)
        # eager warm-up on a side stream: lazy initialisation (module loads, allocator pools, lazily created camera
        # planes) must not happen inside the capture
        global CAPTURING
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        CAPTURING = True
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    step_fn(self.action)
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: API calls of other host threads (RCCL's watchdog polling its events, a data loader) must not
            # invalidate this thread's capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = step_fn(self.action)
        finally:
            CAPTURING = False
        self.replays = 0
        self.executed_steps = warmup      # what the env went through while this object was built: the warm-up steps ran, the captured step was only recorded

    def __call__(self, action):
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if action.ndim == 1:
                action = action[None]
            if action.shape != self.action.shape:
                raise AssertionError(f"Received action of shape {tuple(action.shape)} but expected shape {tuple(self.action.shape)}")
            self.action.copy_(action)
        self.graph.replay()
        self.replays += 1
        return _clone_tree(tuple(self.out))
