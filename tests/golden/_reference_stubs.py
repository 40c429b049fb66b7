"""Lets the pure-Python / pure-torch parts of the reference (/root/reference/mani_skill) run in this container.

The reference's third-party stack (sapien, gymnasium, trimesh, ...) is not installed and cannot be; its task logic (evaluate,
observations, rewards), controllers' action scaling, pose algebra and rotation conversions, however, are plain torch code that only
*mentions* those packages.  ``install()`` registers an import hook under which every missing package becomes a module whose
Capitalised attributes are fresh empty classes (valid base classes and isinstance targets) and whose other attributes are
MagicMocks; ``Fake`` stands in for ``self`` when a reference method is called unbound: class attributes, properties and methods
come from the reference class, explicitly given attributes win, anything else is a mock.

Used only by tests/golden/make_reference_vectors.py (run HERE, where /root/reference exists); the vectors it writes are what
travels.
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest.mock import MagicMock

STUBS = ("sapien", "gymnasium", "trimesh", "transforms3d", "tyro", "h5py", "dacite", "pytorch_kinematics", "mplib", "imageio", "cv2",
         "IPython", "pynvml", "fast_kinematics", "pymeshlab", "open3d", "gdown", "lxml", "GPUtil", "toppra", "yourdfpy", "coacd")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "Pose":   # sapien.Pose(p, q): the one record the task code builds literals of
            import numpy as np

            class Pose:
                def __init__(self, p=(0, 0, 0), q=(1, 0, 0, 0)):
                    self.p, self.q = np.asarray(p, dtype=np.float32), np.asarray(q, dtype=np.float32)
            setattr(self, name, Pose)
            return Pose
        if name[:1].isupper():
            class _Base:
                def __init__(self, *a, **k):
                    self.args, self.kwargs = a, k

                def __class_getitem__(cls, item):
                    return cls
            v = type(name, (_Base,), {})
        else:
            v = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        m.__version__ = "0.29.1"
        return m

    def exec_module(self, module):
        pass


def install(reference_root="/root/reference"):
    sys.dont_write_bytecode = True      # read-only checkout: no __pycache__ left behind
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


class Fake:
    """``self`` for a reference method called unbound."""

    def __init__(self, cls, **attrs):
        object.__setattr__(self, "_cls", cls)
        for k, v in attrs.items():
            object.__setattr__(self, k, v)

    def __getattr__(self, name):
        cls = object.__getattribute__(self, "_cls")
        for c in cls.__mro__:
            if name in c.__dict__:
                v = c.__dict__[name]
                if isinstance(v, property):
                    return v.fget(self)
                if isinstance(v, (staticmethod, classmethod)):
                    return v.__get__(None, cls)
                if callable(v):
                    return types.MethodType(v, self)
                return v
        m = MagicMock(name=name)
        object.__setattr__(self, name, m)
        return m


def ns(**kw):
    return types.SimpleNamespace(**kw)
