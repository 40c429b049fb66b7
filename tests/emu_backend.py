"""Test double: the PRODUCT's HIP sources (maniskill_amd/csrc/msk_physx.hip and its headers, unmodified) compiled for the CPU against the
programming-model emulation of tests/hipemu (fibers per work-item, wavefront rendezvous for cross-lane operations, host memory for device
memory) behind maniskill_amd's PhysxGpuSystem class.  What the GPU-less container can check of the kernels themselves: their arithmetic,
indexing, lane mappings, lists, scans and masks -- against the CPU oracle, bit for bit (tests/test_hip_emulation.py).  It is slow (a
cross-lane operation costs 128 fiber switches) and says nothing about races or timing: the -m gpu tests remain the parity tests proper."""
import os
import subprocess

import torch

from maniskill_amd import _native as N
from maniskill_amd.physx import PhysxGpuSystem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_LIB = os.path.join(EMU_DIR, os.environ.get("EMU_LIB_NAME") or "libmsk_emu.so")      # EMU_LIB_NAME: a variant build (tests/hipemu/Makefile), candidate tests only

_lib = None


def emu_lib() -> N.NativeLib:
    global _lib
    if _lib is None:
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:      # (pytest -n: one worker builds, the others find it done)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", EMU_DIR, os.path.basename(EMU_LIB)])      # (make rebuilds it when a kernel source changed)
        _lib = N.NativeLib(EMU_LIB, "msk_")
    return _lib


class EmuPhysxSystem(PhysxGpuSystem):
    """The HIP kernels under emulation: same Python surface, host memory."""

    host_memory = True

    def __init__(self, template, num_envs, sim_config=None):
        super().__init__(torch.device("cpu"), template, num_envs, sim_config, lib=emu_lib())
