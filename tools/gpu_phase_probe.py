"""Development aid: builds the library with -DMSK_PROFILE_PHASES into /tmp and prints the mean cycles of the
solver phases (count | stage | rows J | Y | A | sweeps | finish) per contact-count class."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "maniskill_amd", "csrc")
lib = os.path.join(src, os.environ.get("PROBE_LIB", "libmsk_prof.so"))      # libmsk_prof_nostats.so: the same without the narrowphase's atomic counters (they cost a box-box block ~14 us)      # build it where hipcc is (the build container: `make -C maniskill_amd/csrc libmsk_prof.so`); it travels with the snapshot
if "PROBE_LIB" not in os.environ and (not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".h", ".hip")))):
    subprocess.check_call(f"make -C {src} libmsk_prof.so", shell=True)
from maniskill_amd import _native as N
N.DEFAULT_LIB = lib
from maniskill_amd.envs.pick_cube import PickCubeEnv
from maniskill_amd.envs.peg_insertion_side import PegInsertionSideEnv
n = int(os.environ.get("PROBE_ENVS", "4096"))
env = (PegInsertionSideEnv if os.environ.get("PROBE_ENV", "PickCube") == "Peg" else PickCubeEnv)(num_envs=n, device="cuda:0")
env.reset(seed=2022); torch.manual_seed(0)
dll = env.px.lib.dll
dll.msk_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
names = ["count", "stage", "rowsJ", "Y", "A", "sweeps", "finish"]
def report(tag):
    out = np.zeros(n * 16 + 64 + 8192, dtype=np.int64)
    dll.msk_debug_phases(env.px.ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    t = out[:n * 8].reshape(n, 8)
    d = np.diff(t[:, :7], axis=1)
    c = env.px.get_env_contact_counts()
    for lo, hi in ((0, 4), (5, 12), (13, 20), (21, 48)):
        sel = (c >= lo) & (c <= hi)
        if sel.sum():
            tot = t[sel, 6] - t[sel, 0]
            print(tag, f"contacts {lo}-{hi} n={sel.sum()}", {k: int(v) for k, v in zip(names[1:], d[sel].mean(0))}, "total mean", int(tot.mean()), "max", int(tot.max()),
                  "of the slowest env:", {k: int(v) for k, v in zip(names[1:], d[sel][tot.argmax()])}, "its contacts", int(c[sel][tot.argmax()]))
def np_report(tag, launches):
    out = np.zeros(n * 16 + 64 + 8192, dtype=np.int64)
    dll.msk_debug_phases(env.px.ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    d = out[n * 8:n * 8 + 24].reshape(3, 8)
    for t, name in enumerate(("plane", "boxbox", "gjk")):
        it = max(d[t, 0], 1)
        print(tag, name, f"items/launch {d[t,0]/launches:.0f} hits {d[t,5]/launches:.0f} primary mean {d[t,1]/it:.0f} max {d[t,2]} manifold mean {d[t,3]/it:.0f} max {d[t,4]} cycles")
    x = out[n * 8 + 24:n * 8 + 24 + 16]
    items = max(d[2, 0], 1)
    print(tag, f"hull items: through EPA {x[0]/launches:.1f}/launch, EPA cycles mean {x[1]/max(x[0],1):.0f} max {x[2]}; GJK iterations mean {x[3]/items:.2f} max {x[4]}; "
          f"cull stage mean {x[5]/items:.0f} cycles, items into GJK {x[6]/launches:.1f}/launch; primary histogram (<25k <50k <100k <150k <200k <300k <400k more) per launch {[round(float(v)/launches, 1) for v in x[8:16]]}")
def where_in_the_launch(name, rows, cyc):
    """slot 7 of a row: the 100 MHz clock at the wave's first and last stamp (low / high word) -- of the LAST launch: how long the launch is from its first wave's
    start to its last wave's end, how late waves start, what a wave's cycle count is in time (= the clock the chip ran at)"""
    w = rows[:, 7].astype(np.uint64)
    ok = w != 0
    t0 = (w[ok] & np.uint64(0xffffffff)).astype(np.int64)
    t1 = (w[ok] >> np.uint64(32)).astype(np.int64)
    t1 = np.where(t1 < t0, t1 + (1 << 32), t1)
    first = t0.min()
    start, dur = (t0 - first) / 100.0, (t1 - t0) / 100.0
    span = (t1.max() - first) / 100.0
    ghz = cyc[ok] / np.maximum(dur, 1e-3) / 1e3
    q = lambda a, p: float(np.percentile(a, p))      # noqa: E731
    print(f"{name}: first wave start -> last wave end {span:.1f} us; a wave starts {q(start, 50):.1f} us (median) / {q(start, 90):.1f} (90 %) / {q(start, 99):.1f} (99 %) / {start.max():.1f} (last) after the first, {100.0 * float((start > 5.0).mean()):.1f} % later than 5 us; "
          f"a wave lasts {dur.mean():.1f} us mean / {dur.max():.1f} max = {q(ghz, 50):.2f} GHz (median of cycles / time); the last wave to end started at {float(start[np.argmax(t1)]):.1f} us and lasted {float(dur[np.argmax(t1)]):.1f}")



def where_all(tag):
    o = np.zeros(n * 16 + 64 + 8192, dtype=np.int64)
    dll.msk_debug_phases(env.px.ctx, o.ctypes.data_as(C.POINTER(C.c_longlong)))
    r = o[:n * 8].reshape(n, 8)
    where_in_the_launch(tag + " k_csolve  ", r, (r[:, 6] - r[:, 0]).astype(np.float64))
    r = o[n * 8 + 64:n * 16 + 64].reshape(n, 8)
    where_in_the_launch(tag + " k_dynamics", r, (r[:, 6] - r[:, 0]).astype(np.float64))
    # the narrowphase's workgroups (grid: env groups x rows, x fastest; rows in dispatch order: the hull rows (4), the box-box rows (2), the plane row): one word each
    w = o[n * 16 + 64:n * 16 + 64 + 4096].astype(np.uint64)
    w2 = o[n * 16 + 64 + 4096:n * 16 + 64 + 8192].astype(np.uint64)      # the workgroup's phases: five 12-bit offsets from its entry, 10 ns units
    ok = np.nonzero(w)[0]
    if len(ok):
        t0 = (w[ok] & np.uint64(0xffffffff)).astype(np.int64); t1 = (w[ok] >> np.uint64(32)).astype(np.int64)
        t1 = np.where(t1 < t0, t1 + (1 << 32), t1)
        first = t0.min(); start = (t0 - first) / 100.0; dur = (t1 - t0) / 100.0; end = (t1 - first) / 100.0
        gx = (n + 15) // 16
        kinds = ok // gx
        q = lambda a, p: float(np.percentile(a, p))      # noqa: E731
        print(f"{tag} k_narrowphase: {len(ok)} workgroups, first start -> last end {end.max():.1f} us; starts: median {q(start, 50):.1f} / 90 % {q(start, 90):.1f} / last {start.max():.1f} us, {100.0 * float((start > 5.0).mean()):.1f} % later than 5 us; "
              f"a workgroup lasts {dur.mean():.1f} us mean / {q(dur, 99):.1f} (99 %) / {dur.max():.1f} max; the last to end: kind row {int(kinds[np.argmax(end)])}, started at {float(start[np.argmax(end)]):.1f} us, lasted {float(dur[np.argmax(end)]):.1f}")
        for k in np.unique(kinds):
            sel = kinds == k
            ph = np.stack([((w2[ok][sel] >> np.uint64(12 * j)) & np.uint64(4095)).astype(np.float64) / 100.0 for j in range(5)], axis=1)
            big = dur[sel] > 10
            if big.any():
                pm = ph[big].mean(0)
                print(f"      kind row {int(k)}, the {int(big.sum())} workgroups over 10 us, first pass, mean us from entry: operands fetched {pm[1]:.1f}, contacts computed {pm[2]:.1f}, speculative points filtered {pm[0]:.1f}, old record read + count / total updated {pm[4]:.1f}, slots written {pm[3]:.1f}; whole workgroup {dur[sel][big].mean():.1f}")
            print(f"      kind row {int(k)}: {int(sel.sum())} workgroups, start median {q(start[sel], 50):.1f} / last {start[sel].max():.1f} us, lasts mean {dur[sel].mean():.1f} / 99 % {q(dur[sel], 99):.1f} / max {dur[sel].max():.1f} us, last end {end[sel].max():.1f} us, over 10 us: {int((dur[sel] > 10).sum())}")



for _ in range(3): env.step(torch.zeros(n, 8, device="cuda:0"))
report("zero  ")
where_all("zero actions, step 3:")
for _ in range(17): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
where_all("random actions, step 20:")
for _ in range(43): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
report("random")
np_report("random(all steps)", 63 * 5 + 5)
env.px.lib.dll.msk_debug_reset.argtypes = [C.c_void_p]
env.px.lib.dll.msk_debug_reset(env.px.ctx)
for _ in range(int(os.environ.get("PROBE_STEPS", "100"))): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
env.px.lib.dll.msk_debug_reset(env.px.ctx)
for _ in range(40): env.step(2 * torch.rand(n, 8, device="cuda:0") - 1)
report("random, steps 160-200")
np_report("random(steps 160-200)", 40 * 5)
print("solver classes", env.px.get_solver_class_counts())
out = np.zeros(n * 16 + 64 + 8192, dtype=np.int64)
dll.msk_debug_phases(env.px.ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)))
dd = np.diff(out[n * 8 + 64:n * 16 + 64].reshape(n, 8)[:, :7], axis=1)
print("k_dynamics phases (mean cycles): forward", int(dd[:, 0].mean()), "rnea", int(dd[:, 1].mean()), "backward", int(dd[:, 2].mean()),
      "crba", int(dd[:, 3].mean()), "solve", int(dd[:, 4].mean()), "free", int(dd[:, 5].mean()), "total", int(dd.sum(1).mean()))
where_all("random actions, last step:")
