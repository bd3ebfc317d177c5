"""Minimal stand-ins for mmengine's ``InstanceData`` / mmdet's ``DetDataSample``: a bag of equally long
tensors with the handful of operations the fork's code uses (``len``, indexing, ``cat``, attribute
get/set/del, ``pop``) -- faster_rcnn_roi_replay.py:78-108, standard_roi_replay_head.py:133-150."""
import torch


class Instances:
    def __init__(self, **fields):
        object.__setattr__(self, "_fields", dict(fields))

    def __getattr__(self, name):
        fields = object.__getattribute__(self, "_fields")
        if name in fields:
            return fields[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._fields[name] = value

    def __delattr__(self, name):
        del self._fields[name]

    def __contains__(self, name):
        return name in self._fields

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._fields[item]
        if isinstance(item, int):
            item = slice(item, item + 1) if item != -1 else slice(-1, None)
        return Instances(**{k: v[item] for k, v in self._fields.items()})

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __len__(self):
        for v in self._fields.values():
            return int(v.shape[0])
        return 0

    def __deepcopy__(self, memo):
        return Instances(**{k: v.clone() for k, v in self._fields.items()})

    def keys(self):
        return self._fields.keys()

    def pop(self, name):
        return self._fields.pop(name)

    def get(self, name, default=None):
        return self._fields.get(name, default)

    @staticmethod
    def cat(instances_list):
        first = instances_list[0]
        return Instances(**{k: torch.cat([inst[k] for inst in instances_list], dim=0) for k in first.keys()})


class DetSample:
    """One image's annotations: ``gt_instances`` (bboxes xyxy [G x 4], labels [G]), after ``predict`` also
    ``pred_instances`` (bboxes, scores, labels); ``img_shape`` = (H, W) of the padded input."""

    def __init__(self, gt_instances=None, img_shape=None, pred_instances=None):
        self.gt_instances = gt_instances if gt_instances is not None else Instances(
            bboxes=torch.zeros(0, 4), labels=torch.zeros(0, dtype=torch.int64))
        self.img_shape = img_shape
        self.pred_instances = pred_instances
        self.ignored_instances = None

    def get(self, name, default=None):
        return getattr(self, name, default)

    def __deepcopy__(self, memo):
        import copy
        return DetSample(copy.deepcopy(self.gt_instances, memo), self.img_shape,
                         copy.deepcopy(self.pred_instances, memo) if self.pred_instances is not None else None)
