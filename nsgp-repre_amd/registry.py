"""Registries under the reference's names.

The reference registers its classes in MMEngine's ``OPTIMIZERS`` / ``RUNNERS`` /
``MODELS`` (mmdet/engine/optimizers/SGD_NSCL.py:15, nsrunner_roi_replay.py:111,
standard_roi_replay_head.py:30).  When mmengine is importable we register into
those very registries, so ``cl_faster_rcnn_cfgs/*.py`` (``type='SGDNSCL'`` ...)
resolve to this package unchanged.  Without mmengine (this image) a minimal
registry with the same ``register_module`` / ``build`` surface stands in.
"""


class _LocalRegistry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force and self.module_dict[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        cfg.update(default_args)
        typ = cfg.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**cfg)


try:  # pragma: no cover - mmengine is absent in this image
    from mmengine.registry import MODELS, OPTIMIZERS, RUNNERS
    DATASETS = None     # the fork's own host-only dataset classes stay in charge (see datasets/__init__.py)
    HAVE_MMENGINE = True
except Exception:
    OPTIMIZERS = _LocalRegistry("optimizer")
    RUNNERS = _LocalRegistry("runner")
    MODELS = _LocalRegistry("model")
    DATASETS = _LocalRegistry("dataset")
    HAVE_MMENGINE = False


def register(registry, name=None):
    """``force=True`` so that re-registering over the reference fork's own class works."""
    def deco(cls):
        if registry is None:
            return cls
        try:
            registry.register_module(name=name or cls.__name__, force=True, module=cls)
        except TypeError:
            registry.register_module(name=name or cls.__name__, module=cls)
        return cls
    return deco
