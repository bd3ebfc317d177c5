import json
import os.path as osp
import xml.etree.ElementTree as ET
from typing import List, Optional, Sequence

from ..registry import DATASETS, register

VOC_CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")

COCO_CLASSES = (
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant",
    "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe",
    "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat",
    "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl",
    "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
    "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave",
    "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush")


def task_label_range(task_split: Sequence[int], task_id: int) -> range:
    """Labels that exist for task ``task_id`` (1-based): xml_style_task.py:33,161; coco_task.py:70-72."""
    assert 0 < task_id < len(task_split), \
        f"Task split start from 1, end with {len(task_split) - 1}, current task_id == {task_id}"
    return range(task_split[task_id - 1], task_split[task_id])


class _TaskDataset:
    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, i):
        return self.data_list[i]

    def _finish(self, filter_cfg, test_mode):
        self.filter_cfg, self.test_mode = filter_cfg, test_mode
        self.data_list = self.filter_data(self.load_data_list())


@register(DATASETS)
class XMLTask(_TaskDataset):
    """VOC-style XML annotations, one file per image listed in ``ann_file`` (xml_style_task.py)."""

    CLASSES = None

    def __init__(self, ann_file: str, data_root: str = "", data_prefix: Optional[dict] = None, img_subdir: str = "JPEGImages",
                 ann_subdir: str = "Annotations", task_split: Sequence[int] = (0, 10, 20), task_id: int = 1,
                 classes: Optional[Sequence[str]] = None, filter_cfg: Optional[dict] = None, test_mode: bool = False, **_):
        self.img_subdir, self.ann_subdir = img_subdir, ann_subdir
        self.task_split, self.task_id = list(task_split), task_id
        self.labels = task_label_range(self.task_split, task_id)
        self.classes = tuple(classes if classes is not None else self.CLASSES)
        assert self.classes, "`classes` in `XMLDataset` can not be None."
        self.cat2label = {c: i for i, c in enumerate(self.classes)}
        self.data_root = data_root
        self.sub_data_root = osp.join(data_root, (data_prefix or {}).get("sub_data_root", ""))
        self.ann_file = osp.join(data_root, ann_file)
        self.bbox_min_size = (filter_cfg or {}).get("bbox_min_size", None)
        self._finish(filter_cfg, test_mode)

    def load_data_list(self) -> List[dict]:
        out = []
        with open(self.ann_file) as f:
            img_ids = [line.strip() for line in f if line.strip()]
        for img_id in img_ids:
            info = self.parse_data_info(img_id)
            if len(info["instances"]) != 0:           # images without an object of THIS task vanish (:67-68)
                out.append(info)
        return out

    def parse_data_info(self, img_id: str) -> dict:
        xml_path = osp.join(self.sub_data_root, self.ann_subdir, f"{img_id}.xml")
        root = ET.parse(xml_path).getroot()
        size = root.find("size")
        if size is None:
            raise ValueError(f"{xml_path}: no <size>; decoding the image for its shape is the image pipeline's job")
        return dict(img_path=osp.join(self.sub_data_root, self.img_subdir, f"{img_id}.jpg"), img_id=img_id, xml_path=xml_path,
                    height=int(size.find("height").text), width=int(size.find("width").text),
                    instances=self._parse_instance_info(root, minus_one=True))

    def _parse_instance_info(self, root, minus_one: bool = True) -> List[dict]:
        instances = []
        for obj in root.findall("object"):
            name = obj.find("name").text
            if name not in self.cat2label:
                continue
            difficult = obj.find("difficult")
            difficult = 0 if difficult is None else int(difficult.text)
            bb = obj.find("bndbox")
            bbox = [int(float(bb.find(k).text)) for k in ("xmin", "ymin", "xmax", "ymax")]
            if minus_one:                              # VOC coordinates are 1-based
                bbox = [x - 1 for x in bbox]
            ignore = False
            if self.bbox_min_size is not None:
                assert not self.test_mode
                ignore = (bbox[2] - bbox[0]) < self.bbox_min_size or (bbox[3] - bbox[1]) < self.bbox_min_size
            label = self.cat2label[name]
            if label in self.labels:                   # the task split (:161-162)
                instances.append(dict(ignore_flag=1 if (difficult or ignore) else 0, bbox=bbox, bbox_label=label))
        return instances

    def filter_data(self, data_list: List[dict]) -> List[dict]:
        if self.test_mode:
            return data_list
        cfg = self.filter_cfg or {}
        empty, min_size = cfg.get("filter_empty_gt", False), cfg.get("min_size", 0)
        return [d for d in data_list if not (empty and len(d["instances"]) == 0) and min(d["width"], d["height"]) >= min_size]


@register(DATASETS)
class VOCTask(XMLTask):
    CLASSES = VOC_CLASSES

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.dataset_type = "VOC2007" if "VOC2007" in self.sub_data_root else "VOC2012" if "VOC2012" in self.sub_data_root else None


@register(DATASETS)
class CocoTaskDataset(_TaskDataset):
    """COCO-style JSON (coco_task.py).  Labels number the categories named in ``classes`` in annotation-file order
    (what ``COCO.get_cat_ids(cat_names=...)`` returns); the task keeps the category ids of its label range."""

    def __init__(self, ann_file: str, data_root: str = "", data_prefix: Optional[dict] = None, task_split: Sequence[int] = (0, 40, 80),
                 task_id: int = 1, classes: Optional[Sequence[str]] = None, filter_cfg: Optional[dict] = None,
                 test_mode: bool = False, **_):
        self.task_split, self.task_id = list(task_split), task_id
        lab = task_label_range(self.task_split, task_id)
        self.current_task_split = [lab.start, lab.stop]
        self.classes = tuple(classes if classes is not None else COCO_CLASSES)
        self.img_prefix = osp.join(data_root, (data_prefix or {}).get("img", ""))
        self.ann_file = osp.join(data_root, ann_file)
        self._finish(filter_cfg, test_mode)

    def load_data_list(self) -> List[dict]:
        with open(self.ann_file) as f:
            coco = json.load(f)
        self.cat_ids = [c["id"] for c in coco["categories"] if c["name"] in self.classes]
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        self.keep_cat = [self.cat_ids[i] for i in range(*self.current_task_split)]
        anns, ids = {}, []
        self.cat_img_map = {cid: [] for cid in self.cat_ids}
        for a in coco["annotations"]:
            anns.setdefault(a["image_id"], []).append(a)
            ids.append(a["id"])
            if a["category_id"] in self.cat_img_map:
                self.cat_img_map[a["category_id"]].append(a["image_id"])
        assert len(set(ids)) == len(ids), f"Annotation ids in '{self.ann_file}' are not unique!"
        return [self.parse_data_info(img, anns.get(img["id"], [])) for img in coco["images"]]

    def parse_data_info(self, img: dict, ann_info: List[dict]) -> dict:
        instances = []
        for ann in ann_info:
            if ann.get("ignore", False):
                continue
            x1, y1, w, h = ann["bbox"]
            inter_w = max(0, min(x1 + w, img["width"]) - max(x1, 0))
            inter_h = max(0, min(y1 + h, img["height"]) - max(y1, 0))
            if inter_w * inter_h == 0 or ann["area"] <= 0 or w < 1 or h < 1:
                continue
            if ann["category_id"] not in self.keep_cat:          # the task split (:175-177)
                continue
            inst = dict(ignore_flag=1 if ann.get("iscrowd", False) else 0, bbox=[x1, y1, x1 + w, y1 + h],
                        bbox_label=self.cat2label[ann["category_id"]])
            if ann.get("segmentation", None):
                inst["mask"] = ann["segmentation"]
            instances.append(inst)
        return dict(img_path=osp.join(self.img_prefix, img["file_name"]), img_id=img["id"], seg_map_path=None, height=img["height"],
                    width=img["width"], instances=instances)

    def filter_data(self, data_list: List[dict]) -> List[dict]:
        if self.test_mode or self.filter_cfg is None:
            return data_list
        empty, min_size = self.filter_cfg.get("filter_empty_gt", False), self.filter_cfg.get("min_size", 0)
        ids_in_cat = set()
        for cid in self.keep_cat:
            ids_in_cat |= set(self.cat_img_map[cid])
        ids_in_cat &= set(d["img_id"] for d in data_list)
        return [d for d in data_list if not (empty and d["img_id"] not in ids_in_cat) and min(d["width"], d["height"]) >= min_size]
