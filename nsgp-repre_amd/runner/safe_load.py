"""Reading the task hand-off files (``covariance.pth``, ``rois_etc.pth``, ``mask.pth``, ``ewc_reg_terms_ewc.pth``) whoever wrote them.

The reference writes two of them as pickled ``collections.defaultdict`` objects -- ``torch.save(self.fea_in, ...)`` with
``fea_in = defaultdict(dict)`` after task 1 (nsrunner_roi_replay.py:708,757) and ``{'importance': defaultdict(list), 'task_param':
defaultdict(list)}`` (:957-958,989) -- and reads them back with a plain ``torch.load``.  ``torch.load(weights_only=True)`` (the only
mode this package uses) refuses a ``defaultdict`` even when the class is allow-listed ("Can only SETITEM for dict, OrderedDict,
Counter"), so a task-1 directory written by the reference could not be consumed.  ``load_handoff`` therefore tries the weights-only
loader first and, only when that one stops at a ``defaultdict``, reads the archive with an ALLOW-LIST unpickler: the globals it
resolves are ``collections.defaultdict`` / ``OrderedDict``, the builtin ``dict`` / ``list`` (as default factories), torch's storage
type tags and the tensor rebuild entry -- the last two mapped to local functions that only reinterpret bytes of the archive.  Any
other global raises; nothing named by the file is ever called.  ``defaultdict`` containers come back as plain ``dict``.
"""
import collections
import pickle
import zipfile

import torch

_STORAGE_DTYPES = {"FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
                   "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
                   "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool}


class _StorageTag:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor(flat, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    return torch.as_strided(flat, tuple(size), tuple(stride), int(storage_offset))


_ALLOWED = {("collections", "defaultdict"): collections.defaultdict, ("collections", "OrderedDict"): collections.OrderedDict,
            ("builtins", "dict"): dict, ("builtins", "list"): list, ("__builtin__", "dict"): dict, ("__builtin__", "list"): list,
            ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor}
_ALLOWED.update({("torch", name): _StorageTag(dt) for name, dt in _STORAGE_DTYPES.items()})


class _AllowListUnpickler(pickle.Unpickler):
    def __init__(self, file, archive, prefix):
        super().__init__(file)
        self._zf, self._prefix, self._flat = archive, prefix, {}

    def find_class(self, module, name):
        try:
            return _ALLOWED[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"hand-off file names the global {module}.{name}, which is not on the allow-list") from None

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) == 5 and pid[0] == "storage" and isinstance(pid[1], _StorageTag)):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        _, tag, key, _location, numel = pid
        if key not in self._flat:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            self._flat[key] = torch.frombuffer(bytearray(raw), dtype=tag.dtype) if int(numel) > 0 else torch.empty(0, dtype=tag.dtype)
        return self._flat[key]


def _plain(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device) if device is not None else obj
    if isinstance(obj, dict):
        return {k: _plain(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v, device) for v in obj)
    return obj


def _load_allow_listed(path, map_location):
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise pickle.UnpicklingError(f"{path}: not a torch.save zip archive")
        prefix = pkl[0][:-len("/data.pkl")]
        if f"{prefix}/byteorder" in zf.namelist() and zf.read(f"{prefix}/byteorder").strip() != b"little":
            raise pickle.UnpicklingError(f"{path}: big-endian archives are not supported")
        with zf.open(pkl[0]) as f:
            obj = _AllowListUnpickler(f, zf, prefix).load()
    return _plain(obj, map_location)


def load_handoff(path, map_location=None):
    """``torch.load(path, weights_only=True)``; a file that loader refuses BECAUSE it holds a ``collections.defaultdict`` (what the
    reference's task-1 ``covariance.pth`` and its ``ewc_reg_terms_ewc.pth`` are) is read with the allow-list unpickler above."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as exc:
        if "defaultdict" not in str(exc):
            raise
    return _load_allow_listed(path, map_location)
