"""Regional Prototype Replay RoI heads -- interface mirror of
mmdet/models/roi_heads/standard_roi_replay_head.py (``StandardRoIReplayHead`` :30,
``StandardMultiPrototypeReplayHead`` :375-501).

``PrototypeReplay`` carries the fork's own logic (bank construction at start of task t, the
per-step replay loss) against any object that has a ``bbox_head``; when mmdet is importable the
registered head classes inherit its ``StandardRoIHead`` so the reference configs build them
unchanged; without mmdet they inherit ``detection.StandaloneRoIHead`` (the same stock recipe in plain
PyTorch) and take the same constructor keywords.
"""
import os.path as osp
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import MODELS, register
from .. import ops
from .prototype_bank import build_prototype_bank
from .roi_dump import RoIDump

try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.roi_heads.standard_roi_head import StandardRoIHead as _Base
    _HAVE_MMDET = True
except Exception:
    from ..detection.roi_parts import StandaloneRoIHead as _Base     # the same recipe in plain torch
    _HAVE_MMDET = False


def get_work_dir(previous_path: str) -> str:
    """standard_roi_replay_head.py:363-370: the current work dir is the previous one with its
    trailing ``_N`` bumped; paths containing "coco" map to "./"."""
    if "coco" in previous_path:
        return "./"
    parts = previous_path.split("_")
    parts[-1] = str(int(parts[-1]) + 1)
    return "_".join(parts)


class PrototypeReplay:
    """Bank construction + replay loss, independent of the detector framework."""

    def init_prototype_replay(self, previous_path: Optional[str], task_id: int, task_split: Sequence[int],
                              max_prototype: int = 10, device=None):
        self.replay = False
        self.task_split = list(task_split)
        self.task_id = task_id
        self.max_proto = max_prototype
        if previous_path is None or not osp.exists(previous_path):
            return
        assert task_id != 1                                                       # head:400
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.replay = True
        with torch.no_grad():
            rois = torch.load(osp.join(previous_path, "rois_etc.pth"), map_location=device, weights_only=True)
            (feats, self.cls_targets, self.cls_weights, self.bbox_targets, self.bbox_weights, self.roiss) = rois
            mask_file = osp.join(previous_path, "mask.pth")
            saved = torch.load(mask_file, map_location="cpu", weights_only=True) if osp.exists(mask_file) else []
            feats = feats.reshape(feats.shape[0], -1).float().contiguous()
            bank, labels, masks, _ = build_prototype_bank(feats, self.cls_targets, self.task_split, task_id,
                                                          max_prototype, saved=saved)
            self.bbox_featss = bank                                               # [K x 12544]
            self.tmp_label = labels                                               # [K] int64
            # the (possibly extended) mask list goes to the CURRENT work dir (head:451-452)
            merged = list(saved)
            for c, ml in zip(range(self.task_split[0], self.task_split[task_id - 1]), masks):
                if c < len(merged):
                    merged[c] = ml
                else:
                    merged.append(ml)
            torch.save(merged, osp.join(get_work_dir(previous_path), "mask.pth"))

    def replay_loss(self, bbox_feats, sampling_results=None, rois=None) -> dict:
        """head:468-501: the bank through the bbox head; keep the columns of the classes seen so
        far + background; ``CE(softmax(.), labels)`` (the double softmax is the reference's)."""
        if getattr(self, "with_shared_head", False):
            bbox_feats = self.shared_head(bbox_feats)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        results = dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)
        pre_idx = self.task_split[self.task_id]
        kept = torch.cat([cls_score[:, :pre_idx], cls_score[:, -1:]], dim=-1)
        labels = self.tmp_label.to(kept.device)
        loss = ops.double_softmax_cross_entropy(kept, labels)   # fused wave-reduction kernels (fp32; GPU only, no torch fallback)
        results.update(replay_loss=dict(replay_loss_cls=loss))
        return results

    #: run the per-step replay pass on the fused HIP path (csrc/replay_head.hip) when the bbox head has the Shared2FC shape;
    #: False = always the module-by-module path of ``replay_loss`` (torch GEMMs + the fused CE)
    fused_replay = True

    def _fused_replay_operands(self):
        """(w1, b1, w2, b2, [head weights], [head biases]) when ``bbox_head`` is a two-shared-FC task head whose kept class columns are whole Linear heads
        (the ``Shared2FCBBoxHeadTask`` of every reference config), else None."""
        h = self.bbox_head
        if getattr(self, "with_shared_head", False) or not hasattr(h, "shared_fcs") or not hasattr(h, "fc_cls"):
            return None
        if (getattr(h, "num_shared_convs", 0) or getattr(h, "num_shared_fcs", 0) != 2 or getattr(h, "num_cls_convs", 0)
                or getattr(h, "num_cls_fcs", 0) or getattr(h, "with_avg_pool", False) or getattr(h, "background_nums", 1) != 1):
            return None
        split, tid = list(h.task_split), h.task_id
        if split != list(self.task_split) or tid != self.task_id or len(h.fc_cls) != len(split):
            return None
        mods = list(h.fc_cls._modules.values())            # (slicing a ModuleList builds a new container: 20 us of a 0.2 ms host-bound pass)
        live = mods[:tid] + [mods[-1]]                      # tasks 1..task_id + background: exactly the kept columns (head:495-497)
        if sum(m.out_features for m in live[:-1]) != split[tid] - split[0] or split[0] != 0:
            return None
        fc1, fc2 = h.shared_fcs
        if any(m.bias is None for m in (fc1, fc2, *live)):
            return None
        if len(live) > 16:
            return None
        return (fc1.weight, fc1.bias, fc2.weight, fc2.bias, [m.weight for m in live], [m.bias for m in live])

    def replay_loss_fused(self, bbox_feats):
        """``replay_loss`` for the loss alone, fused (6 + 5 launches instead of ~40): returns ``dict(replay_loss_cls=...)`` or None
        when the head does not have the shape the fused path covers."""
        ops_ = self._fused_replay_operands() if self.fused_replay else None
        if ops_ is None or not bbox_feats.is_cuda:
            return None
        feats = bbox_feats.reshape(bbox_feats.shape[0], -1)
        if feats.shape[0] > 512 or sum(w.shape[0] for w in ops_[4]) > 256 or feats.shape[1] != ops_[0].shape[1]:
            return None
        loss, _scores = ops.replay_head_loss(feats, self.tmp_label.to(feats.device), *ops_)
        return dict(replay_loss_cls=loss)

    def add_replay_loss(self, losses: dict) -> dict:
        """The tail of ``loss`` (head:454-466): stock RoI losses + ``replay_loss_cls``."""
        if self.replay:
            # the bank pass stays fp32 under an autocast training step (>= the reference's precision; the
            # bf16 path of this K x 12544 x 1024 GEMM measured 7x slower on hipBLASLt)
            with torch.autocast(device_type=self.bbox_featss.device.type, enabled=False):
                fused = self.replay_loss_fused(self.bbox_featss)
                losses.update(fused if fused is not None else self.replay_loss(self.bbox_featss)["replay_loss"])
        return losses


def _init_base(self, bbox_roi_extractor, bbox_head, mask_roi_extractor, mask_head, shared_head, train_cfg, test_cfg,
               init_cfg):
    if _HAVE_MMDET:  # pragma: no cover
        _Base.__init__(self, bbox_roi_extractor, bbox_head, mask_roi_extractor, mask_head, shared_head, train_cfg,
                       test_cfg, init_cfg)
    else:
        nn.Module.__init__(self)
        self.init_standalone(bbox_roi_extractor, bbox_head, train_cfg, test_cfg)


@register(MODELS)
class StandardRoIReplayHead(RoIDump, _Base):
    """Raw-RoI replay against the teacher (head:30-104): 64 random stored RoIs per step, MSE between the
    student's and the teacher's class scores.  ``teacher_model`` is attached by the runner (runner:533)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None, init_cfg=None, previous_path=None):
        _init_base(self, bbox_roi_extractor, bbox_head, mask_roi_extractor, mask_head, shared_head, train_cfg, test_cfg,
                   init_cfg)
        self.replay = False
        if previous_path is not None and osp.exists(previous_path):
            self.replay = True
            (self.bbox_featss, self.cls_targets, self.cls_weights, self.bbox_targets, self.bbox_weights,
             self.roiss) = torch.load(osp.join(previous_path, "rois_etc.pth"), weights_only=True)

    def replay_loss(self, bbox_feats, sampling_results=None, rois=None) -> dict:
        if getattr(self, "with_shared_head", False):
            bbox_feats = self.shared_head(bbox_feats)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        teacher_cls_score, _ = self.teacher_model.bbox_head(bbox_feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats,
                    replay_loss=dict(replay_loss_cls=F.mse_loss(cls_score, teacher_cls_score)))

    def add_replay_loss(self, losses: dict) -> dict:
        if self.replay:
            dev = next(self.parameters()).device
            pick = torch.randperm(self.bbox_featss.shape[0])[:64].to(self.bbox_featss.device)       # head:56
            losses.update(self.replay_loss(self.bbox_featss[pick].to(dev))["replay_loss"])
        return losses


@register(MODELS)
class StandardPrototypeReplayHead(PrototypeReplay, RoIDump, _Base):
    """One (coarse) prototype per old class, label = row index (head:206-300)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None, init_cfg=None, previous_path=None, task_id=1,
                 task_split=(0, 10, 20)):
        _init_base(self, bbox_roi_extractor, bbox_head, mask_roi_extractor, mask_head, shared_head, train_cfg, test_cfg,
                   init_cfg)
        self.replay, self.task_split, self.task_id = False, list(task_split), task_id
        if previous_path is not None and osp.exists(previous_path):
            assert task_id != 1
            self.replay = True
            dev = torch.device("cuda", torch.cuda.current_device())
            rois = torch.load(osp.join(previous_path, "rois_etc.pth"), map_location=dev, weights_only=True)
            feats, self.cls_targets = rois[0].reshape(rois[0].shape[0], -1).float().contiguous(), rois[1]
            rows = [ops.masked_mean(feats[self.cls_targets == c].contiguous())
                    for c in range(self.task_split[0], self.task_split[task_id - 1])]
            self.bbox_featss = torch.cat(rows, dim=0)
            self.tmp_label = torch.arange(self.bbox_featss.shape[0], device=dev)                     # head:293


@register(MODELS)
class StandardMultiPrototypeReplayHead(PrototypeReplay, RoIDump, _Base):
    """Same keywords as the reference (head:377-390)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None, init_cfg=None, previous_path=None, task_id=1,
                 task_split=(0, 10, 20), max_prototype=10, work_dir=None):
        _init_base(self, bbox_roi_extractor, bbox_head, mask_roi_extractor, mask_head, shared_head, train_cfg, test_cfg,
                   init_cfg)
        self.init_prototype_replay(previous_path, task_id, task_split, max_prototype)

    def loss(self, x, rpn_results_list, batch_data_samples) -> dict:
        """head:454-466: the stock RoI losses, then ``replay_loss_cls`` from the bank."""
        return self.add_replay_loss(_Base.loss(self, x, rpn_results_list, batch_data_samples))
