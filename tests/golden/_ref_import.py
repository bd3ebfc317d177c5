"""Load the reference's hot-path modules file-by-file (golden generation ONLY).

Used only by ``make_golden.py`` inside the build container, where
``/root/reference`` is mounted.  Nothing here is imported by the tests, the
product or the bench; the reference's files never travel to the GPU box.

The reference cannot be imported as a package (``mmdet/__init__.py`` needs
mmcv/mmengine, which are absent).  SURVEY.md §8c route: fabricate empty
stand-in *modules* for the absent third-party packages so that
``importlib`` can execute the five fork files directly.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types

import torch.nn as nn

REF = "/root/reference"
_PREFIXES = ("mmengine", "mmcv", "mmdet", "torchvision")


class _Registry:
    def register_module(self, *a, **k):
        def deco(cls):
            return cls
        return deco

    def build(self, cfg, *a, **k):
        raise RuntimeError("registry stub cannot build")


class _RoIHeadStub(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._dummy = nn.Parameter(nn.Parameter.__new__(nn.Parameter).new_zeros(1))


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.isupper():
            v = _Registry()
        elif name.endswith("RoIHead"):
            v = _RoIHeadStub
        elif name[0].isupper():
            v = type(name, (), {})
        else:
            def v(*a, **k):
                return None
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        _installed = True


def load(relpath):
    """Execute one reference file under its dotted in-package name so that its
    relative imports resolve to the stand-in modules."""
    install()
    name = relpath[:-3].replace("/", ".")
    spec = importlib.util.spec_from_file_location(name, f"{REF}/{relpath}")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
