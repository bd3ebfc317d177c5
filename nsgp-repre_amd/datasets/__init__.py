"""Task-split annotation readers: ``XMLTask`` / ``VOCTask`` (mmdet/datasets/xml_style_task.py:13-194,
voc_task.py:7-31) and ``CocoTaskDataset`` (mmdet/datasets/coco_task.py:14-230) without MMDetection.

The fork's own part of those classes is one rule: of all annotated objects only those whose label lies in
``[task_split[task_id-1], task_split[task_id])`` exist for task ``task_id`` -- old-class objects in a new-task
image are simply not annotated (that is what the teacher's pseudo labels later restore).  These readers produce
the same ``data_list`` dictionaries (``img_path, img_id, height, width, instances[{bbox, bbox_label,
ignore_flag}]``) with the same per-format rules, from the standard library only (``xml.etree`` / ``json``); the
image pipeline stays with whoever consumes the list.  Where mmdet is installed the fork's own dataset classes
(host-only Python) keep working unchanged and these are not needed.
"""
from .task_split import COCO_CLASSES, VOC_CLASSES, CocoTaskDataset, VOCTask, XMLTask, task_label_range

__all__ = ["COCO_CLASSES", "VOC_CLASSES", "CocoTaskDataset", "VOCTask", "XMLTask", "task_label_range"]
