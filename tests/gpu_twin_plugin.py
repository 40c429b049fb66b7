"""pytest plugin (``-p gpu_twin_plugin``): run ``-m gpu`` test bodies WITHOUT a GPU, on the product's HIP sources under the emulation of tests/hipemu.

What it swaps: ``maniskill_amd._native.default_lib`` hands out the emulated library (tests/emu_backend.py), ``PhysxGpuSystem`` keeps its buffers in host memory,
and a TorchFunctionMode sends every ``device="cuda:0"`` / ``.to("cuda:0")`` / ``.cuda()`` to the CPU.  What it cannot do: HIP graphs, events, streams, 4096-env
rollouts in seconds -- tests that need those are not twinned (tests/test_gpu_twins.py lists the ones that are).  Purpose: a -m gpu test whose BODY is wrong (an
array size that no longer matches the ABI: round 4's red suite) fails here, in the CPU suite, before it costs a GPU minute."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _to_cpu(x):
    if isinstance(x, torch.device):
        return torch.device("cpu") if x.type == "cuda" else x
    if isinstance(x, str) and x.split(":")[0] == "cuda":
        return "cpu"
    return x


class CudaIsCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs:
            kwargs["device"] = _to_cpu(kwargs["device"])
        if func in (torch.Tensor.to, torch.Tensor.cuda):
            if func is torch.Tensor.cuda:
                return args[0]
            args = tuple(_to_cpu(a) for a in args)
        return func(*args, **kwargs)


_mode = None


def pytest_configure(config):
    global _mode
    from maniskill_amd import _native, physx
    from emu_backend import emu_lib
    _native.default_lib = emu_lib
    physx.N.default_lib = emu_lib
    physx.PhysxGpuSystem.host_memory = True
    orig_init = physx.PhysxGpuSystem.__init__

    def init(self, device, *a, **k):
        orig_init(self, _to_cpu(device), *a, **k)
    physx.PhysxGpuSystem.__init__ = init
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.is_available = lambda: False
    _mode = CudaIsCpu()
    _mode.__enter__()


def pytest_collection_modifyitems(config, items):
    for it in items:
        # the first_hardware_run isolation (a subprocess per test) is for kernels that may fault on hardware: not here
        it.own_markers = [mk for mk in it.own_markers if mk.name != "first_hardware_run"]


def pytest_unconfigure(config):
    if _mode is not None:
        _mode.__exit__(None, None, None)
