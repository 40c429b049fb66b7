"""pytest plugin (``-p ref_pytest_plugin``) for running the REFERENCE's own test files, unmodified, on this backend: installs the
sapien shim + stand-ins and selects the backend named by MSK_REF_BACKEND ("oracle": CPU checker, "hip": libmsk_physx.so) before
the reference's test modules are imported.  Used by tests/test_reference_conformance.py in a subprocess."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    import ref_harness
    gym = ref_harness.setup(os.environ.get("MSK_REF_BACKEND", "oracle"))
    if gym is None:
        raise RuntimeError("no ManiSkill checkout found")
    for m in ("gpu_sim", "slow"):
        config.addinivalue_line("markers", f"{m}: marker of the reference's test-suite")
