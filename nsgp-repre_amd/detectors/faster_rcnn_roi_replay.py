"""``FasterRCNNRoIReplay`` -- the detector-side boundary of the hot path
(mmdet/models/detectors/faster_rcnn_roi_replay.py:14, forward :189-236).

The fork's own parts: the mode dispatch (``mode='nullspace'``: a loss pass without the teacher, used
under the covariance hooks by ``cal_fea_in``; ``mode='roi_replay'``: the RoI feature dump used by
``cal_rois``) and the teacher pseudo-labelling at the top of ``loss`` (det:65-109), whose
per-box Python loop is one kernel here.  The detector forward/backward itself is stock PyTorch-ROCm
(north_star): mmdet's ``TwoStageDetector`` when it is installed, ``nsgp_repre_amd.detection`` otherwise.
"""
import torch
import torch.nn as nn

from ..registry import MODELS, register

try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.detectors.two_stage import TwoStageDetector as _Base
    _HAVE_MMDET = True
except Exception:
    _Base = nn.Module
    _HAVE_MMDET = False


def filter_pseudo_labels(pred_bboxes, pred_scores, gt_bboxes, rpn_thresh: float, roi_thresh: float):
    """The teacher pseudo-label filter of one image (det:78-108) as ONE kernel instead of a Python loop
    with a host sync per box: returns (add_to_rpn, add_to_roi) bool masks over the teacher's predictions,
    in order, so ``gt.cat([gt, pseudo[mask]])`` reproduces the reference's appended sets."""
    from .. import ops
    return ops.pseudo_label_filter(pred_bboxes.float().contiguous(), pred_scores.float().contiguous(),
                                   gt_bboxes.float().contiguous(), rpn_thresh, roi_thresh)


class RoIReplayModes:
    """``forward(inputs, data_samples, mode)`` with the two extra modes of det:229-232."""

    def forward(self, inputs, data_samples=None, mode: str = "tensor"):
        if mode == "loss":
            return self.loss(inputs, data_samples)
        elif mode == "predict":
            return self.predict(inputs, data_samples)
        elif mode == "tensor":
            return self._forward(inputs, data_samples)
        elif mode == "nullspace":
            return self.loss(inputs, data_samples, use_teacher_student=False)
        elif mode == "roi_replay":
            return self.get_bbox_stuff(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')

    def get_bbox_stuff(self, batch_inputs, batch_data_samples):
        """det:146-186: features -> RPN proposals -> ``roi_head.get_bbox_stuff`` 6-tuple."""
        x = self.extract_feat(batch_inputs)
        rpn_results_list = self.rpn_head.predict(x, batch_data_samples, rescale=False)
        return self.roi_head.get_bbox_stuff(x, rpn_results_list, batch_data_samples)


@register(MODELS)
class FasterRCNNRoIReplay(RoIReplayModes, _Base):
    def __init__(self, *args, **kwargs):
        if _HAVE_MMDET:  # pragma: no cover
            super().__init__(*args, **kwargs)
        else:
            nn.Module.__init__(self)
            for k, v in kwargs.items():      # stand-alone: sub-modules are passed in ready-built
                setattr(self, k, v)
        self.rpn_thresh, self.roi_thresh = 0.5, 0.7   # det:39-40; the runner overwrites them from rr_thresh (runner:439-440)

    # -- the stock TwoStageDetector pieces, for the stand-alone build (mmdet provides them otherwise) ----------
    if not _HAVE_MMDET:
        with_rpn = True

        def extract_feat(self, batch_inputs):
            return self.neck(self.backbone(batch_inputs))

        def predict(self, batch_inputs, batch_data_samples, rescale=True):
            x = self.extract_feat(batch_inputs)
            proposals = self.rpn_head.predict(x, batch_data_samples, rescale=False)
            preds = self.roi_head.predict(x, proposals, batch_data_samples, rescale=rescale)
            for sample, p in zip(batch_data_samples, preds):
                sample.pred_instances = p
            return batch_data_samples

        def _forward(self, batch_inputs, batch_data_samples=None):
            x = self.extract_feat(batch_inputs)
            return self.rpn_head(x)

    def pseudo_labelled_samples(self, batch_inputs, batch_data_samples):
        """det:65-109: teacher predictions -> (rpn_data_samples, batch_data_samples) with the accepted pseudo
        boxes appended.  The per-box Python loop (one ``.item()`` per box) is the fused filter above."""
        import copy
        with torch.no_grad():
            self.teacher_model.eval()
            preds = self.teacher_model.predict(batch_inputs, copy.deepcopy(batch_data_samples), rescale=False)
            rpn_samples = copy.deepcopy(batch_data_samples)
            for pred, gt_sample, rpn_sample in zip(preds, batch_data_samples, rpn_samples):
                inst = pred.pred_instances
                if len(inst) == 0:
                    continue
                to_rpn, to_roi = filter_pseudo_labels(inst.bboxes, inst.scores, gt_sample.gt_instances.bboxes,
                                                      self.rpn_thresh, self.roi_thresh)
                pseudo = inst[:]
                pseudo.__delattr__("scores")
                rpn_sample.gt_instances = rpn_sample.gt_instances.cat([rpn_sample.gt_instances, pseudo[to_rpn]])
                gt_sample.gt_instances = gt_sample.gt_instances.cat([gt_sample.gt_instances, pseudo[to_roi]])
        return rpn_samples, batch_data_samples

    def loss(self, batch_inputs, batch_data_samples, use_teacher_student=True):
        """det:44-142: features; teacher pseudo-labels (task >= 2, unless ``mode='nullspace'``); RPN loss on the
        RPN set with labels zeroed; RoI-head loss (stock + replay) on the RoI set."""
        import copy
        x = self.extract_feat(batch_inputs)
        rpn_data_samples = None
        if hasattr(self, "teacher_model") and use_teacher_student:
            rpn_data_samples, batch_data_samples = self.pseudo_labelled_samples(batch_inputs, batch_data_samples)
        losses = dict()
        if self.with_rpn:
            train_cfg = getattr(self, "train_cfg", None)
            proposal_cfg = train_cfg.get("rpn_proposal", self.test_cfg.rpn) if train_cfg is not None else None
            rpn_data_samples = rpn_data_samples if rpn_data_samples else copy.deepcopy(batch_data_samples)
            for data_sample in rpn_data_samples:                                   # class-agnostic RPN targets
                data_sample.gt_instances.labels = torch.zeros_like(data_sample.gt_instances.labels)
            rpn_losses, rpn_results_list = self.rpn_head.loss_and_predict(x, rpn_data_samples, proposal_cfg=proposal_cfg)
            for key in list(rpn_losses.keys()):
                if "loss" in key and "rpn" not in key:
                    rpn_losses[f"rpn_{key}"] = rpn_losses.pop(key)
            losses.update(rpn_losses)
        else:
            assert batch_data_samples[0].get("proposals", None) is not None
            rpn_results_list = [data_sample.proposals for data_sample in batch_data_samples]
        losses.update(self.roi_head.loss(x, rpn_results_list, batch_data_samples))
        return losses
