"""Runs test functions of the REFERENCE's own test modules (by node id) on this backend, in this process.

``python tests/ref_run_node.py <backend> <node id> [<node id> ...]`` with node ids as pytest writes them
(``tests/test_gpu_envs.py::test_partial_resets`` or a whole file ``tests/structs/test_pose.py``); a trailing
``[name=value,...]`` keeps only the parametrisations with those values.  The modules are imported by
name, so the byte-compiled build of the reference (oracle/_ref/maniskill, no sources: what travels to the GPU box) works the same
as a checkout; ``pytest.mark.parametrize`` marks are expanded here.  Exit code 0 = every selected test passed.
"""
import importlib
import itertools
import sys
import traceback


def _cases(fn):
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    axes = []
    for m in marks:
        names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
        vals = []
        for v in m.args[1]:
            v = getattr(v, "values", v) if hasattr(v, "values") and hasattr(v, "marks") else v     # pytest.param(...)
            vals.append(dict(zip(names, v if len(names) > 1 else [v])))
        axes.append(vals)
    for combo in itertools.product(*axes) if axes else [()]:
        kw = {}
        for d in combo:
            kw.update(d)
        yield kw


def main():
    backend, nodes = sys.argv[1], sys.argv[2:]
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ref_harness
    if ref_harness.setup(backend) is None:
        print("no ManiSkill checkout found")
        return 3
    failed = ran = 0
    import tempfile
    os.chdir(tempfile.mkdtemp(prefix="ref_run_node_"))   # the reference's tests record videos / trajectories relative to the cwd (GBs for a whole module)
    for node in nodes:
        sel = {}
        if node.endswith("]") and "[" in node:          # tests/x.py::test_y[env_id=PickCube-v1,obs_mode=rgb]: only these parameter values
            node, _, flt = node[:-1].partition("[")
            sel = dict(kv.split("=", 1) for kv in flt.split(","))
        path, _, func = node.partition("::")
        mod = importlib.import_module(path[:-3].replace("/", ".") if path.endswith(".py") else path.replace("/", "."))
        names = [func] if func else [n for n in dir(mod) if n.startswith("test_") and callable(getattr(mod, n))]
        for name in names:
            fn = getattr(mod, name)
            for kw in _cases(fn):
                if any(str(kw.get(k)) != v for k, v in sel.items()):
                    continue
                ran += 1
                label = f"{path}::{name}" + (f"[{'-'.join(str(v) for v in kw.values())}]" if kw else "")
                try:
                    fn(**kw)
                    print("PASSED", label, flush=True)
                except BaseException:
                    failed += 1
                    print("FAILED", label, flush=True)
                    traceback.print_exc()
    print(f"{ran - failed} passed, {failed} failed")
    return 1 if failed or not ran else 0


if __name__ == "__main__":
    sys.exit(main())
