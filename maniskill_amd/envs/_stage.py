"""Host -> device uploads of a reset.

The fused envs evaluate an episode's randomness on the host (float64, rounded once: the initial state has the same bits whatever device the env lives on) and
hand the results to the device as float32 rows.  Done as ``torch.as_tensor(array, device=...)`` each of these is a PAGEABLE copy: a staging copy inside the HIP
runtime and a blocking wait -- 2-3 ms apiece on an MI355X box, ten per reset, and the cost of a reset then dominates a 50-step episode (measured in round 5:
profiles/r05_soak_where_before_fix.log, 3 -> 15 ms per auto reset over a long run).  Here every upload of a reset goes through a reusable PINNED buffer of its
own (keyed by its position in the reset and its row width), filled on the host and copied asynchronously on the current stream."""
from __future__ import annotations

import numpy as np
import torch


class HostStage:
    def __init__(self, device, max_rows: int):
        self.device = torch.device(device)
        self.max_rows = int(max_rows)
        self._slots = {}
        self._k = 0

    def begin(self):
        """start of a reset: the uploads that follow are numbered from zero again (the k-th upload of every reset reuses the k-th buffer)"""
        self._k = 0

    def __call__(self, a) -> torch.Tensor:
        a = np.asarray(a)
        if self.device.type != "cuda":
            return torch.as_tensor(a, dtype=torch.float32)
        rows = a.shape[0] if a.ndim > 0 else 1
        if a.ndim == 0 or rows > self.max_rows:      # (a scalar, or more rows than envs: not a per-env table)
            return torch.as_tensor(a, dtype=torch.float32, device=self.device)
        key = (self._k, a.shape[1:])
        self._k += 1
        slot = self._slots.get(key)
        if slot is None:
            slot = self._slots[key] = (torch.empty((self.max_rows,) + tuple(a.shape[1:]), dtype=torch.float32).pin_memory(), torch.cuda.Event())
        else:
            slot[1].synchronize()            # the copy that last read this buffer (a reset ago) has long finished; this makes it a guarantee
        host = slot[0][:rows]
        host.numpy()[...] = a                # float64 -> float32 on the host: the same rounding torch.as_tensor(dtype=float32) applies
        out = host.to(self.device, non_blocking=True)
        slot[1].record(torch.cuda.current_stream(self.device))
        return out
