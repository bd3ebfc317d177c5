"""NSGP task hand-off machinery: the pieces of ``BRNullSpaceRunner`` that are on the hot path
(mmdet/engine/runner/nsrunner_roi_replay.py), written against plain torch objects so that they run
with or without MMEngine:

* ``wire_param_names``         runner:473-484   param_groups[i]['params'|'names'] from named_parameters()
* ``should_ignore``            runner:643-650   ``re.match(ignore_key, name)`` over the ignore list
* ``CovarianceCollector``      runner:723-729, 876-934   forward hooks -> HIP implicit-im2col SYRK
* ``cal_fea_in``               runner:705-763   one pass over the loader in eval mode, C1 all-reduce,
                                                += previous task's covariance, save ``covariance.pth``
* ``update_optim_transforms``  runner:635-662   load covariance -> get_eigens -> get_transforms
* ``cal_rois``                 runner:777-868   RoI-feature dump, C2 ragged gather, save ``rois_etc.pth``
"""
import logging
import os
import os.path as osp
import re
from collections import defaultdict
from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from . import dist as D
from .safe_load import load_handoff

logger = logging.getLogger("nsgp_repre_amd")

#: appended to the config's ignore_keys by the reference (runner:355)
EXTRA_IGNORE_KEYS = ["roi_head.bbox_head.fc_cls", "roi_head.bbox_head.fc_reg", "teacher"]


def full_ignore_keys(cfg_ignore_keys: Optional[Sequence[str]]) -> list:
    return (list(cfg_ignore_keys) if cfg_ignore_keys else []) + EXTRA_IGNORE_KEYS


def should_ignore(name: str, ignore_keys: Sequence[str]) -> bool:
    """``re.match`` = anchored at the start of the name (runner:646)."""
    return any(bool(re.match(k, name)) for k in ignore_keys)


def unwrap(model: nn.Module) -> nn.Module:
    """DDP / MMDistributedDataParallel expose the real model as ``.module`` (is_model_wrapper)."""
    return model.module if isinstance(model, (nn.parallel.DistributedDataParallel, nn.DataParallel)) else model


def wire_param_names(optimizer, model: nn.Module) -> None:
    """Rebuild every param group's ``params`` in ``named_parameters()`` order and record the
    parallel ``names`` list the NSGP optimizers key their projectors on (runner:473-484).
    Parameters with ``requires_grad=False`` drop out of the optimizer, exactly as in the reference."""
    model = unwrap(model)
    group_of = {}
    for i, group in enumerate(optimizer.param_groups):
        for p in group["params"]:
            group_of[id(p)] = i
        group["params"] = []
        group["names"] = []
    for name, param in model.named_parameters():
        if param.requires_grad:
            i = group_of[id(param)]          # KeyError if the optimizer never saw it, like the reference
            optimizer.param_groups[i]["params"].append(param)
            optimizer.param_groups[i]["names"].append(name)


class CovarianceStreams:
    """Side HIP streams for the covariance accumulation of independent layers.

    One hooked forward of R-50-FPN is 61 accumulations of 4-5 launches each, most of them far too small to fill 256 CUs (a
    1x1 convolution with D = 64 is ONE output tile); on a single stream they run back to back and the forward is bound by
    the sum of their latencies.  The layers are independent of each other and only READ the activations, so slot ``i`` (a
    hooked module) is pinned to side stream ``i % n``: its launches are ordered behind the producer of its input (the
    stream that calls ``run``) and behind the previous accumulations into the same C (same stream, always), and overlap
    the other layers'.  Every stream owns its workspace.  ``join()`` makes the calling stream wait for all of them --
    before the covariances are read (all-reduce, save, eigendecomposition).  Results are bitwise what one stream gives.
    CONSTRAINT: the main stream does not wait for the side streams until ``join()``; the activation must not be written in place
    after the hooked convolution (true of the detectors here; an arbitrary model with an in-place op on a convolution's input would
    race) -- ``n_streams = 1`` removes the concurrency.  Since round 3 only the layers the grouped pass cannot take run this way."""

    def __init__(self, n_streams: int = 4):
        self.n = max(1, int(n_streams))
        self._streams, self._ws = None, None

    def _setup(self, device):
        if self._streams is None or self._streams[0].device != device:
            self._streams = [torch.cuda.Stream(device=device) for _ in range(self.n)]
            self._ws = [None] * self.n

    def workspace(self, k, nbytes, device):
        w = self._ws[k]
        if w is None or w.numel() < nbytes or w.device != device:
            w = self._ws[k] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return w

    def run(self, slot: int, x: torch.Tensor, fn):
        """``fn(workspace_getter)`` on the slot's side stream, ordered behind everything already queued on the current
        stream; ``x`` (the activation the launches read) is kept alive for that stream."""
        self._setup(x.device)
        k = slot % self.n
        side = self._streams[k]
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            out = fn(lambda nbytes: self.workspace(k, nbytes, x.device))
        x.record_stream(side)
        return out

    def join(self):
        if self._streams is not None:
            cur = torch.cuda.current_stream(self._streams[0].device)
            for s in self._streams:
                cur.wait_stream(s)


class CovarianceCollector:
    """Forward hooks that accumulate ``C_k (+)= X^T X`` per hooked Conv2d / Linear.

    Mirrors ``compute_cov`` + ``update_cov`` (runner:876-934) with four changes of *how*: the
    module->name map is built once instead of scanning ``named_modules()`` inside every hook call
    (runner:893-896), X is never materialised (HIP implicit im2col, see csrc/covariance.hip),
    there is no ``empty_cache()`` per call (runner:915), and the accumulations of different layers run on
    ``n_streams`` side HIP streams (``CovarianceStreams``; 1 = everything on the current stream) -- call ``join()``
    (``remove()`` does) before reading ``fea_in``.  ``fea_in`` has the reference's layout:
    ``{module_name + '.weight': [D x D] fp32}``."""

    def __init__(self, model: nn.Module, ignore_keys: Sequence[str], n_streams: int = 4, grouped: bool = True):
        self.model = unwrap(model)
        self.ignore_keys = list(ignore_keys)
        self.fea_in: Dict[str, torch.Tensor] = {}
        self._names = {}
        self._slots = {}
        self._handles = []
        self._workspace = None
        self._streams = CovarianceStreams(n_streams) if n_streams > 1 else None
        #: GROUPED pass (default on the GPU): the hooks of the convolutions whose D = Cin*kh*kw is a multiple of 64 only stash their
        #: input; ``flush()`` -- called by ``cal_fea_in`` after every forward, by ``join()`` and ``remove()`` -- accumulates all of
        #: them in a handful of launches (``ops.CovGroupPlan``: one tile table over all layers, no split-K; 3x3 layers in the correlation form).  The other layers
        #: (the 7x7 stem, Linear) are accumulated at hook time as before, on the side streams.  False = every layer at hook time.
        self.grouped = grouped
        self._pending = []
        self._plans = {}
        self._group_ws = None

    MAX_PLANS = 8      # grouped plans kept alive (one per input geometry)

    def hooked_modules(self):
        # every module that has a `.weight` and is not ignored (runner:723-724); only Conv2d and
        # Linear contribute in compute_cov (runner:901-913), the rest are no-ops
        return [(n, m) for n, m in self.model.named_modules()
                if hasattr(m, "weight") and not should_ignore(n, self.ignore_keys)]

    def register(self):
        for n, m in self.hooked_modules():
            self._names[m] = n + ".weight"
            self._slots[m] = len(self._slots)
            self._handles.append(m.register_forward_hook(self.compute_cov))
        return self

    def join(self):
        """Everything stashed is accumulated and the current stream waits for every side stream: call before reading ``fea_in``."""
        self.flush()
        if self._streams is not None:
            self._streams.join()

    def remove(self):
        self.join()
        for h in self._handles:
            h.remove()
        self._handles = []

    def close(self):
        """Release the grouped plans' device tables (explicitly, on the calling thread)."""
        for plan in self._plans.values():
            plan.close()
        self._plans = {}
        self._group_ws = None

    @torch.no_grad()
    def flush(self):
        """Accumulate the inputs stashed since the last flush: ONE grouped run for the first occurrence of every layer, the
        single-layer path (in stream order) for anything else (a module called twice in one forward; a layer the plan declines)."""
        if not self._pending:
            return
        pending, self._pending = self._pending, []
        first, rest, seen = [], [], set()
        for n, x, k, s, p, version in pending:
            if x._version != version:
                raise RuntimeError(f"{n}: the input of this convolution was modified in place after the convolution ran; the grouped covariance "
                                   "pass reads activations at the end of the forward -- use CovarianceCollector(..., grouped=False)")
            (rest if n in seen else first).append((n, x, k, s, p))
            seen.add(n)
        geoms = tuple((x.shape[0], x.shape[1], x.shape[2], x.shape[3], k, s, p) for _n, x, k, s, p in first)
        dev = first[0][1].device
        # one plan per input geometry (real loaders pad to many sizes): the most recent MAX_PLANS are kept, all of them share ONE
        # workspace (the materialised operands: ~3 GB for R-50-FPN at 800 x 1344) that only ever grows
        plan = self._plans.pop((geoms, dev), None)
        if plan is None:
            plan = ops.CovGroupPlan(geoms, dev)
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans))).close()
        self._plans[(geoms, dev)] = plan                       # (re-)inserted last: dicts keep insertion order -> LRU
        if self._group_ws is None or self._group_ws.numel() < plan.workspace_bytes or self._group_ws.device != dev:
            self._group_ws = None                                # release before growing
            self._group_ws = torch.empty(max(plan.workspace_bytes, 16), dtype=torch.uint8, device=dev)
        covs = plan.run([x for _n, x, _k, _s, _p in first], [self.fea_in.get(n) for n, *_ in first], workspace=self._group_ws)
        for (n, x, k, s, p), c, ok in zip(first, covs, plan.routes):
            if ok:
                self.fea_in[n] = c
            else:
                rest.insert(0, (n, x, k, s, p))
        for n, x, k, s, p in rest:
            nbytes = ops.cov_workspace_bytes(x.shape[1], x.shape[2], x.shape[3], k, s, p)
            self.fea_in[n] = ops.cov_accumulate_conv2d(x, k, s, p, self.fea_in.get(n), self._ws(nbytes, x.device))

    def _ws(self, nbytes, device):
        if self._workspace is None or self._workspace.numel() < nbytes or self._workspace.device != device:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._workspace

    @torch.no_grad()
    def compute_cov(self, module, fea_in, fea_out):
        name = self._names[module]
        x = fea_in[0]
        if isinstance(module, nn.Linear):
            x = x.detach().float().contiguous()
            self.fea_in[name] = ops.cov_accumulate_linear(x, self.fea_in.get(name))
        elif isinstance(module, nn.Conv2d):
            x = x.detach().float().contiguous()
            k, s, p = module.kernel_size, module.stride, module.padding
            if self.grouped and x.is_cuda and (x.shape[1] * k[0] * k[1]) % 64 == 0:
                # accumulated by flush(), with all the others.  The activation is read AFTER the rest of the forward: an in-place op on
                # a convolution's input behind the convolution would be seen -- the tensor's version counter is checked at flush
                self._pending.append((name, x, tuple(k), tuple(s), tuple(p), x._version))
                return None
            nbytes = ops.cov_workspace_bytes(x.shape[1], x.shape[2], x.shape[3], k, s, p)
            if self._streams is None or not x.is_cuda:
                self.fea_in[name] = ops.cov_accumulate_conv2d(x, k, s, p, self.fea_in.get(name), self._ws(nbytes, x.device))
            else:
                self.fea_in[name] = self._streams.run(
                    self._slots[module], x, lambda ws: ops.cov_accumulate_conv2d(x, k, s, p, self.fea_in.get(name), ws(nbytes)))
        return None


@torch.no_grad()
def cal_fea_in(model: nn.Module, batches: Iterable, ignore_keys: Sequence[str], save_path: Optional[str] = None,
               previous_path: Optional[str] = None, task_id: int = 1, forward=None, grouped: bool = True,
               n_streams: int = 4) -> Dict[str, torch.Tensor]:
    """One hooked pass over ``batches`` in eval mode (runner:705-763).  ``forward(model, batch)`` runs
    one batch through the model (the reference calls ``model(inputs, data_samples, mode='nullspace')``
    after ``data_preprocessor``); the default calls ``model(batch)``.  ``grouped`` / ``n_streams``: see ``CovarianceCollector``
    (``grouped=False, n_streams=1`` is the reference's own order of operations: every hook accumulates on the current stream
    before the next layer runs -- the setting for a model that modifies a convolution's input in place)."""
    collector = CovarianceCollector(model, ignore_keys, n_streams=n_streams, grouped=grouped).register()
    net = unwrap(model)
    was_training = net.training
    net.eval()
    try:
        for batch in batches:
            if forward is not None:
                forward(net, batch)
            else:
                net(batch)
            collector.flush()           # one grouped run per forward: the stashed activations are released here
    finally:
        collector.remove()
        collector.close()
        net.train(was_training)
    fea_in = collector.fea_in
    D.barrier()
    D.all_reduce_dict(fea_in)                                   # C1
    D.barrier()
    if task_id != 1 and previous_path is not None:
        dev = next(net.parameters()).device
        old = load_handoff(previous_path, dev)
        fea_in = {k: v + old[k].to(v.device) for k, v in fea_in.items() if not should_ignore(k, ignore_keys)}
    if save_path is not None and D.get_rank() == 0:
        torch.save(fea_in, save_path)
    return fea_in


@torch.no_grad()
def update_optim_transforms(optimizer, covariance, ignore_keys: Sequence[str], offset: float = 0.0, device=None):
    """runner:635-662: load ``covariance.pth`` (or take a dict), drop ignored keys, then
    ``get_eigens`` + ``get_transforms(offset)``.  (The reference runs the identical
    ``update_model_transforms`` right after, i.e. the decomposition twice; once is enough.)"""
    if isinstance(covariance, (str, os.PathLike)):
        covariance = load_handoff(covariance, device)       # the reference's task-1 file is a pickled defaultdict
    fea_in = {k: (v.to(device) if device is not None else v) for k, v in covariance.items()
              if not should_ignore(k, ignore_keys)}
    D.sharded_eigens(optimizer, fea_in)          # = get_eigens on one GPU; layers spread over the ranks under DDP (C4)
    optimizer.get_transforms(offset=offset)
    return fea_in


@torch.no_grad()
def cal_rois(model: nn.Module, batches: Iterable, save_path: Optional[str] = None, previous_path: Optional[str] = None,
             task_id: int = 1, reserve_per_class: int = 0, num_classes: int = 20, forward=None):
    """RoI-feature dump for RePRE (runner:777-868).  ``forward(model, batch)`` must return the
    6-tuple of ``get_bbox_stuff`` (feats, cls_target, cls_weight, bbox_target, bbox_weight, rois);
    the reference calls ``model(inputs, data_samples, mode='roi_replay')``.  Returns / saves the list
    of six concatenated tensors (``rois_etc.pth``)."""
    net = unwrap(model)
    net.eval()
    cols = [[] for _ in range(6)]
    for batch in batches:
        res = forward(net, batch) if forward is not None else net(batch)
        for c, r in zip(cols, res):
            c.append(r)
    gathered = [torch.cat(D.all_gather_different_shape(torch.cat(c, dim=0))) for c in cols]   # C2
    if reserve_per_class != 0:
        cls_targets = gathered[1]
        picks, res = {}, []
        for tns in gathered:
            tmp = []
            for cls_idx in range(num_classes):
                cls_mask = cls_targets == cls_idx
                if cls_idx not in picks:
                    picks[cls_idx] = torch.randperm(int(cls_mask.sum()))[:reserve_per_class]
                tmp.append(tns[cls_mask][picks[cls_idx].to(tns.device)])
            res.append(torch.cat(tmp, dim=0))
        gathered = res
    if task_id != 1 and previous_path is not None:
        dev = next(net.parameters()).device
        old = load_handoff(previous_path, dev)
        gathered = [torch.cat([o, g], dim=0) for o, g in zip(old, gathered)]
    if save_path is not None and D.get_rank() == 0:
        torch.save(gathered, save_path)
    return gathered


def guard_conv_weights(model: nn.Module, max_moves: int = 64) -> int:
    """Guard against a stock MIOpen defect (profiles/README.md, incident analysis; tools/miopen_overread_repro.py): MIOpen's
    backward-data kernel of a 1x1 convolution reads past the end of its weight tensor -- a GPU memory access fault when the caching
    allocator has placed that weight as the LAST block of a segment and nothing is mapped behind it.  Every convolution weight of
    ``model`` (teacher copy included) that ends exactly where its allocator segment ends is moved to a fresh allocation -- whether
    or not another segment happens to start there today: ``empty_cache()`` (the reference's ``train`` calls it) can unmap that
    neighbour later.  The old block stays referenced by the model (``_segment_end_pads``), so no other weight can land in the slot.
    Both runner faces call this after the model is on the device, after a checkpoint load and after ``attach_teacher`` (the deep
    copy allocates new weights).  Returns how many weights were moved; a no-op on the CPU; costs one allocator snapshot per pass."""
    model = unwrap(model)
    params = [m.weight for m in model.modules() if isinstance(m, nn.Conv2d) and m.weight.is_cuda]
    if not params:
        return 0
    pads = getattr(model, "_segment_end_pads", [])
    moved = 0
    for _ in range(max_moves):
        ends = {s["address"] + s["total_size"] for s in torch.cuda.memory_snapshot()}
        hit = [p for p in params if (p.data_ptr() + p.numel() * p.element_size()) in ends]
        if not hit:
            break
        for p in hit:
            pads.append(p.data)                 # keeps the segment-final block occupied
            p.data = p.data.clone()
            moved += 1
    model._segment_end_pads = pads
    if moved:
        logger.info("guard_conv_weights: moved %d convolution weight(s) off the end of their allocator segment", moved)
    return moved


def find_checkpoint(directory: str, ckpt_keywords: str) -> str:
    """First directory entry whose name contains ``ckpt_keywords`` (runner:295-299, 710-713)."""
    for f in os.listdir(directory):
        if ckpt_keywords in f:
            return osp.join(directory, f)
    raise FileNotFoundError(f"no file containing {ckpt_keywords!r} in {directory}")
