"""RoI-feature dump for RePRE (``StandardRoIReplayHead.get_bbox_stuff``,
mmdet/models/roi_heads/standard_roi_replay_head.py:106-202).

The assign / sample / RoIAlign part is stock mmdet; the fork's own logic is the selection of
EXACTLY five rows per batch (``:163-199``): keep the foreground rows, pad with random background
rows if there are fewer than five, drop random foreground rows if there are more.
"""
import torch

TARGET_ROIS_PER_BATCH = 5   # head:168


def select_five_rois(cls_target: torch.Tensor, bg_class_id: int, target_count: int = TARGET_ROIS_PER_BATCH) -> torch.Tensor:
    """Bool mask over the sampled RoIs with exactly ``target_count`` True entries (or all rows if
    there are fewer rows than that).  Random picks use ``torch.randperm`` on the default generator
    like the reference (``:186,194``)."""
    mask = cls_target != bg_class_id
    delta = target_count - int(torch.sum(mask).item())
    if delta > 0:
        false_indices = torch.where(mask == False)[0]  # noqa: E712
        if len(false_indices) < delta:
            mask[:] = True
        else:
            pick = torch.randperm(len(false_indices))[:delta]
            mask[false_indices[pick.to(false_indices.device)]] = True
    elif delta < 0:
        true_indices = torch.where(mask == True)[0]  # noqa: E712
        drop = torch.randperm(len(true_indices))[:-delta]
        mask[true_indices[drop.to(true_indices.device)]] = False
    return mask


class RoIDump:
    """``get_bbox_stuff`` over the stock assigner / sampler / RoI extractor / ``get_roi_targets`` of whichever
    ``StandardRoIHead`` the head inherits (mmdet's, or ``detection.StandaloneRoIHead``)."""

    def get_bbox_stuff(self, x, rpn_results_list, batch_data_samples, extract_gt=False):
        if hasattr(self, "sampled_roi_stuff"):       # stand-alone head (detection.StandaloneRoIHead): same stock half
            bbox_feats, cls_t, cls_w, box_t, box_w, rois = self.sampled_roi_stuff(x, rpn_results_list, batch_data_samples)
            mask = select_five_rois(cls_t, self.bbox_head.num_classes)
            return bbox_feats[mask], cls_t[mask], cls_w[mask], box_t[mask], box_w[mask], rois[mask]
        return self._get_bbox_stuff_mmdet(x, rpn_results_list, batch_data_samples)

    def _get_bbox_stuff_mmdet(self, x, rpn_results_list, batch_data_samples):  # pragma: no cover - needs mmdet
        from mmdet.models.utils import unpack_gt_instances
        from mmdet.structures.bbox import bbox2roi
        assert len(rpn_results_list) == len(batch_data_samples)
        batch_gt_instances, batch_gt_instances_ignore, _ = unpack_gt_instances(batch_data_samples)
        sampling_results = []
        for i in range(len(batch_data_samples)):
            rpn_results = rpn_results_list[i]
            rpn_results.priors = rpn_results.pop("bboxes")
            assign_result = self.bbox_assigner.assign(rpn_results, batch_gt_instances[i], batch_gt_instances_ignore[i])
            sampling_results.append(self.bbox_sampler.sample(assign_result, rpn_results, batch_gt_instances[i],
                                                             feats=[lvl_feat[i][None] for lvl_feat in x]))
        rois = bbox2roi([res.priors for res in sampling_results])
        bbox_feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        if self.with_shared_head:
            bbox_feats = self.shared_head(bbox_feats)
        bbox_feats = self.bbox_head.get_mid_features(bbox_feats)
        cls_t, cls_w, box_t, box_w = self.bbox_head.get_roi_targets(sampling_results=sampling_results,
                                                                    rcnn_train_cfg=self.train_cfg)
        mask = select_five_rois(cls_t, self.bbox_head.num_classes)
        return bbox_feats[mask], cls_t[mask], cls_w[mask], box_t[mask], box_w[mask], rois[mask]
