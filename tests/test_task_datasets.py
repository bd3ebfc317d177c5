"""the task-split annotation readers against golden G7 (the reference's own XMLTask / CocoTaskDataset
methods run on the same synthetic documents, tests/golden/make_golden.py::run_task_split)."""
import json
import os

import pytest

import inputs as I
import nsgp_repre_amd as N


@pytest.fixture(scope="module")
def g7(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g7_task_split.json")))


def _write_voc(tmp_path):
    docs = I.g7_xml_docs()
    root = tmp_path / "VOCdevkit"
    (root / "VOC2007" / "Annotations").mkdir(parents=True)
    (root / "VOC2007" / "ImageSets" / "Main").mkdir(parents=True)
    ids = [f"{i:06d}" for i in range(len(docs))]
    for i, d in zip(ids, docs):
        (root / "VOC2007" / "Annotations" / f"{i}.xml").write_text(d)
    (root / "VOC2007" / "ImageSets" / "Main" / "trainval.txt").write_text("\n".join(ids) + "\n")
    return str(root), ids


def test_voc_task_reader_matches_the_reference(g7, tmp_path):
    root, ids = _write_voc(tmp_path)
    for case in g7["xml"]:
        cfg = dict(filter_empty_gt=True, min_size=250, bbox_min_size=case["bbox_min_size"])
        ds = N.datasets.VOCTask(ann_file="VOC2007/ImageSets/Main/trainval.txt", data_root=root,
                                data_prefix=dict(sub_data_root="VOC2007/"), task_split=case["task_split"], task_id=case["task_id"],
                                filter_cfg=cfg)
        assert ds.dataset_type == "VOC2007"
        # per document: the reference's instance list (documents without instances of this task never enter the list)
        unfiltered = {d["img_id"]: d for d in ds.load_data_list()}
        for i, want in zip(ids, case["instances"]):
            got = unfiltered[i]["instances"] if i in unfiltered else []
            assert got == want, (case["task_split"], case["task_id"], i)
        assert [int(d["img_id"]) for d in ds] == case["kept_after_filter"]
        lab = N.datasets.task_label_range(case["task_split"], case["task_id"])
        assert all(inst["bbox_label"] in lab for d in ds for inst in d["instances"])
    assert len(N.datasets.VOC_CLASSES) == 20


def test_coco_task_reader_matches_the_reference(g7, tmp_path):
    (tmp_path / "annotations").mkdir()
    (tmp_path / "annotations" / "instances.json").write_text(json.dumps(I.g7_coco()))
    for case in g7["coco"]:
        ds = N.datasets.CocoTaskDataset(ann_file="annotations/instances.json", data_root=str(tmp_path), data_prefix=dict(img="imgs/"),
                                        task_split=case["task_split"], task_id=case["task_id"], classes=I.G7_COCO_CLASSES,
                                        filter_cfg=dict(filter_empty_gt=True, min_size=32))
        full = ds.load_data_list()
        assert len(full) == len(case["data_list"])
        for got, want in zip(full, case["data_list"]):
            assert got["instances"] == want["instances"] and got["img_id"] == want["img_id"]
            assert (got["height"], got["width"]) == (want["height"], want["width"])
            assert got["img_path"].endswith(want["img_path"])
        assert [d["img_id"] for d in ds] == case["kept_after_filter"]
    assert len(N.datasets.COCO_CLASSES) == 80


def test_task_id_bounds_and_test_mode(tmp_path):
    root, _ = _write_voc(tmp_path)
    kw = dict(ann_file="VOC2007/ImageSets/Main/trainval.txt", data_root=root, data_prefix=dict(sub_data_root="VOC2007/"))
    for bad in (0, 3):
        with pytest.raises(AssertionError):
            N.datasets.VOCTask(task_split=[0, 15, 20], task_id=bad, **kw)
    # evaluation split of the reference configs: val_task_split = [0, train_task_split[task_id]] with the default task_id 1
    ds = N.datasets.VOCTask(task_split=[0, 20], test_mode=True, **kw)
    assert all(0 <= inst["bbox_label"] < 20 for d in ds for inst in d["instances"])
    assert N.registry.DATASETS.build(dict(type="VOCTask", task_split=[0, 20], test_mode=True, **kw)).data_list == ds.data_list
