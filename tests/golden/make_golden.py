"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``python tests/golden/make_golden.py``):
it executes the reference's own files from /root/reference through the
stand-in loader in ``_ref_import.py`` and stores inputs' seeds + the
reference's outputs as small ``.npz`` fixtures.  The tests regenerate the
inputs from the same seeds with ``inputs.py`` (numpy Generator streams are
stable) and never touch /root/reference.
"""
import os
import sys
import tempfile
import warnings
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
import inputs as I  # noqa: E402

warnings.filterwarnings("ignore")
torch.manual_seed(0)
torch.set_num_threads(8)

ref_sgd = R.load("mmdet/engine/optimizers/SGD_NSCL.py")
ref_adamw = R.load("mmdet/engine/optimizers/AdamW_NSCL.py")
ref_adam = R.load("mmdet/engine/optimizers/Adam_NSCL.py")
ref_sgdna = R.load("mmdet/engine/optimizers/SGD_NSCL_NoAdaptive.py")
ref_runner = R.load("mmdet/engine/runner/nsrunner_roi_replay.py")
ref_head = R.load("mmdet/models/roi_heads/standard_roi_replay_head.py")
ref_bbox = R.load("mmdet/models/roi_heads/bbox_heads/convfc_bbox_head_task.py")


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ---------------------------------------------------------------- G1 optimizers
def run_optimizer(kind):
    names, shapes = I.g1_layers()
    params = [nn.Parameter(torch.from_numpy(a)) for a in I.g1_params()]
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1_covariances().items()}
    hp = I.G1_HYPER[kind]
    cls = dict(sgd=ref_sgd.SGDNSCL, sgd_nesterov=ref_sgd.SGDNSCL, adamw=ref_adamw.AdamWNSCL,
               adamw_amsgrad=ref_adamw.AdamWNSCL, adam=ref_adam.AdamNSCL, sgdna=ref_sgdna.SGDNSCLNA)[kind]
    opt = cls(params, svd=True, **hp)
    opt.param_groups[0]["names"] = list(names)
    opt.get_eigens(fea_in)
    opt.get_transforms(offset=I.G1_OFFSET)
    out = {}
    for n in names:
        if n in opt.transforms:
            key = n.replace(".", "_")
            out[f"sigma__{key}"] = opt.eigens[n]["eigen_value"].numpy()
            if kind in ("sgd", "adam", "sgdna"):  # the others share sgd's projector rule
                out[f"P__{key}"] = opt.transforms[n].numpy()
            else:
                out[f"Pnorm__{key}"] = opt.transforms[n].norm().numpy()
    for step in range(I.G1_STEPS):
        for p, g in zip(params, I.g1_grads(step)):
            p.grad = torch.from_numpy(g)
        opt.step()
        for n, p in zip(names, params):
            key = n.replace(".", "_")
            out[f"p_step{step}__{key}"] = p.detach().numpy().copy()
            out[f"g_step{step}__{key}"] = p.grad.numpy().copy()  # the reference mutates .grad
    for n, p in zip(names, params):
        key = n.replace(".", "_")
        st = opt.state[p]
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            if sk in st:
                out[f"{sk}__{key}"] = st[sk].numpy().copy()
    save(f"g1_{kind}.npz", **out)


def run_optimizer_aligned(kind):
    """G1b: the same reference classes on 128-aligned layers (see inputs.py); P only with 'sgd' (AdamW shares its rule)."""
    names, shapes = I.g1b_layers()
    params = [nn.Parameter(torch.from_numpy(a)) for a in I.g1b_params()]
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1b_covariances().items()}
    cls = dict(sgd=ref_sgd.SGDNSCL, adamw=ref_adamw.AdamWNSCL)[kind]
    opt = cls(params, svd=True, **I.G1_HYPER[kind])
    opt.param_groups[0]["names"] = list(names)
    opt.get_eigens(fea_in)
    opt.get_transforms(offset=I.G1_OFFSET)
    out = {}
    for n in names:
        if n in opt.transforms:
            key = n.replace(".", "_")
            out[f"sigma__{key}"] = opt.eigens[n]["eigen_value"].numpy()
            if kind == "sgd":
                out[f"P__{key}"] = opt.transforms[n].numpy()
            else:
                out[f"Pnorm__{key}"] = opt.transforms[n].norm().numpy()
    for step in range(I.G1B_STEPS):
        for p, g in zip(params, I.g1b_grads(step)):
            p.grad = torch.from_numpy(g)
        opt.step()
        for n, p in zip(names, params):
            key = n.replace(".", "_")
            out[f"p_step{step}__{key}"] = p.detach().numpy().copy()
            if step == 0:
                out[f"g_step{step}__{key}"] = p.grad.numpy().copy()
    for n, p in zip(names, params):
        key = n.replace(".", "_")
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq"):
            if sk in opt.state[p]:
                out[f"{sk}__{key}"] = opt.state[p][sk].numpy().copy()
    save(f"g1b_{kind}.npz", **out)


def run_optimizer_lowrank_basis(kind):
    """G1c: the reference's own removed directions U = eigen_vector[:, :r] (SGD_NSCL.py:377 torch.svd; the mask of :270 is a suffix) and
    its step() outputs on layers that reach every rank class and the K-range path of the default low-rank step (see inputs.py)."""
    names, shapes = I.g1c_layers()
    params = [nn.Parameter(torch.from_numpy(a)) for a in I.g1c_params()]
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1c_covariances().items()}
    cls = dict(sgd=ref_sgd.SGDNSCL, sgd_nesterov=ref_sgd.SGDNSCL, adamw=ref_adamw.AdamWNSCL)[kind]
    opt = cls(params, svd=True, **I.G1_HYPER[kind])
    opt.param_groups[0]["names"] = list(names)
    opt.get_eigens(fea_in)
    opt.get_transforms(offset=I.G1_OFFSET)
    out = {}
    for n in names:
        if n in opt.transforms:
            key = n.replace(".", "_")
            mask = opt.adaptive_threshold(opt.eigens[n]["eigen_value"], offset=I.G1_OFFSET)
            r = int(mask.to(torch.int8).argmax())
            assert mask[r:].all() and not mask[:r].any()
            out[f"rank__{key}"] = np.int64(r)
            out[f"Pnorm__{key}"] = opt.transforms[n].norm().numpy()
            if "backbone" in n:    # the value the reference divided by: its own fp32 `torch.norm(transform)` (SGD_NSCL.py:282-283), same ops
                basis = opt.eigens[n]["eigen_vector"][:, mask]
                out[f"refnorm__{key}"] = torch.norm(torch.mm(basis, basis.transpose(1, 0))).numpy()
                out[f"refnorm64__{key}"] = torch.mm(basis, basis.transpose(1, 0)).double().norm().numpy()
            if kind == "sgd":      # the basis is the optimizer-independent part: stored once
                out[f"sigma__{key}"] = opt.eigens[n]["eigen_value"].numpy()
                out[f"U__{key}"] = opt.eigens[n]["eigen_vector"][:, :r].contiguous().numpy()
    # only the projected parameters after each step are stored (the elementwise state is pinned by G1 / G1b); the second step -- the
    # momentum recurrence on a non-first step, p != 0 -- for plain SGD only, to keep the fixture small
    for step in range(I.G1C_STEPS[kind]):
        for p, g in zip(params, I.g1c_grads(step)):
            p.grad = torch.from_numpy(g)
        opt.step()
        for n, p in zip(names, params):
            if n in opt.transforms:
                out[f"p_step{step}__{n.replace('.', '_')}"] = p.detach().numpy().copy()
    save(f"g1c_{kind}.npz", **out)


# ---------------------------------------------------------------- G2 thresholds
def run_thresholds():
    spectra = I.g2_spectra()
    dummy = [nn.Parameter(torch.zeros(1))]
    sgd = ref_sgd.SGDNSCL(dummy)
    adamw = ref_adamw.AdamWNSCL(dummy)
    out = {}
    for si, s in enumerate(spectra):
        t = torch.from_numpy(s)
        for oi, off in enumerate(I.G2_OFFSETS):
            m1 = sgd.adaptive_threshold(t, off).numpy()
            m2 = adamw.adaptive_threshold(t, off).numpy()
            m3 = ref_head.adaptive_threshold(t, off).numpy()
            assert m1.any() and m2.any()
            out[f"sgd_{si}_{oi}"] = np.int64(m1.argmax())
            out[f"adam_{si}_{oi}"] = np.int64(m2.argmax())
            out[f"head_{si}_{oi}"] = np.int64(m3.argmax())
            assert m1[m1.argmax():].all() and not m1[:m1.argmax()].any()
    save("g2_thresholds.npz", **out)


# ---------------------------------------------------------------- G3 covariance
def run_covariance():
    Runner = ref_runner.BRNullSpaceRunner
    out = {}
    for ci, cfg in enumerate(I.g3_cases()):
        if cfg["kind"] == "conv":
            mod = nn.Conv2d(cfg["cin"], 4, cfg["k"], stride=cfg["s"], padding=cfg["p"], bias=False)
        else:
            mod = nn.Linear(cfg["cin"], 4, bias=False)
        model = nn.Sequential()
        model.add_module("layer", mod)
        fake = object.__new__(Runner)
        fake.model = model
        fake.fea_in = defaultdict(dict)
        h = mod.register_forward_hook(hook=fake.compute_cov)
        with torch.no_grad():
            for x in I.g3_inputs(ci):
                model(torch.from_numpy(x))
        h.remove()
        (k, v), = fake.fea_in.items()
        assert k == "layer.weight"
        out[f"C_{ci}"] = v.numpy()
    save("g3_covariance.npz", **out)


# ---------------------------------------------------------------- G4 prototypes
def run_prototypes():
    feats, cls_t = I.g4_rois()
    feats_t, cls_tt = torch.from_numpy(feats), torch.from_numpy(cls_t)
    n = feats.shape[0]
    rois = [feats_t, cls_tt, torch.ones(n), torch.zeros(n, 4), torch.zeros(n, 4), torch.zeros(n, 5)]
    out = {}
    with tempfile.TemporaryDirectory() as td:
        prev = os.path.join(td, "x_15_5_1")
        cur = os.path.join(td, "x_15_5_2")
        os.makedirs(prev)
        os.makedirs(cur)
        torch.save(rois, os.path.join(prev, "rois_etc.pth"))
        head = ref_head.StandardMultiPrototypeReplayHead(
            previous_path=prev, task_id=2, task_split=I.G4_TASK_SPLIT, max_prototype=I.G4_MAX_PROTO)
        masks = torch.load(os.path.join(cur, "mask.pth"))
    out["bank"] = head.bbox_featss.numpy()
    out["labels"] = head.tmp_label.numpy()
    for c, ml in enumerate(masks):
        out[f"nmask_{c}"] = np.int64(len(ml))
        for j, m in enumerate(ml):
            out[f"mask_{c}_{j}"] = m.numpy()
        # the visiting order the reference used (same ops, same inputs, same build)
        Fc = feats_t[cls_tt == c]
        nrm = Fc.reshape(-1, 7 * 7 * 256) / Fc.reshape(-1, 7 * 7 * 256).norm(dim=-1, keepdim=True)
        sim = nrm @ nrm.t()
        cnt, idx = (sim >= 0.6).long().sum(dim=-1).sort(dim=-1, descending=True)
        out[f"order_{c}"] = idx.numpy()
        out[f"counts_{c}"] = (sim >= 0.6).long().sum(dim=-1).numpy()
        sim64 = (Fc.double() / Fc.double().norm(dim=-1, keepdim=True))
        sim64 = sim64 @ sim64.t()
        out[f"margin_{c}"] = np.float64((sim64 - 0.6).abs().min().item())
    save("g4_prototypes.npz", **out)
    return head


# ---------------------------------------------------------------- G5 replay loss
class _TinyTaskHead(nn.Module):
    """Carrier for the attributes ConvFCBBoxHeadTask.forward reads
    (convfc_bbox_head_task.py:209-288); the forward itself is the reference's."""

    def __init__(self, w):
        super().__init__()
        self.num_shared_convs, self.num_shared_fcs = 0, 2
        self.with_avg_pool, self.with_cls, self.with_reg = False, True, True
        self.reg_class_agnostic = False
        self.task_split, self.task_id = I.G5_TASK_SPLIT, I.G5_TASK_ID
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs, self.cls_fcs = nn.ModuleList(), nn.ModuleList()
        self.reg_convs, self.reg_fcs = nn.ModuleList(), nn.ModuleList()

        def lin(W, b):
            m = nn.Linear(W.shape[1], W.shape[0])
            m.weight.data = torch.from_numpy(W)
            m.bias.data = torch.from_numpy(b)
            return m
        self.shared_fcs = nn.ModuleList([lin(*w["shared"][0]), lin(*w["shared"][1])])
        self.fc_cls = nn.ModuleList([lin(*t) for t in w["cls"]])
        self.fc_reg = nn.ModuleList([lin(*t) for t in w["reg"]])

    forward = ref_bbox.ConvFCBBoxHeadTask.forward


def run_replay_loss():
    w = I.g5_weights()
    bank, labels = I.g5_bank()
    head = object.__new__(ref_head.StandardMultiPrototypeReplayHead)
    nn.Module.__init__(head)
    head.bbox_head = _TinyTaskHead(w)
    head.task_split, head.task_id = I.G5_TASK_SPLIT, I.G5_TASK_ID
    head.tmp_label = torch.from_numpy(labels)
    type(head).with_shared_head = property(lambda self: False)
    res = head.replay_loss(torch.from_numpy(bank), None, None)
    loss = res["replay_loss"]["replay_loss_cls"]
    loss.backward()
    out = dict(loss=loss.detach().numpy(), cls_score=res["cls_score"].detach().numpy(),
               bbox_pred=res["bbox_pred"].detach().numpy())
    for i, m in enumerate(head.bbox_head.shared_fcs):
        out[f"gW_shared{i}"] = m.weight.grad.numpy()
        out[f"gb_shared{i}"] = m.bias.grad.numpy()
    for i, m in enumerate(head.bbox_head.fc_cls):
        out[f"gW_cls{i}"] = np.zeros_like(m.weight.detach().numpy()) if m.weight.grad is None else m.weight.grad.numpy()
        out[f"has_grad_cls{i}"] = np.bool_(m.weight.grad is not None)
    save("g5_replay_loss.npz", **out)


# ---------------------------------------------------------------- G6 EWC regulariser
def run_ewc():
    Runner = ref_runner.BRNullSpaceRunner
    tensors = I.g6_tensors()
    model = nn.Module()
    plist = {}
    for n, (theta, imp, old) in tensors.items():
        p = nn.Parameter(torch.from_numpy(theta.copy()))
        model.register_parameter(n.replace(".", "__"), p)
        plist[n] = p
    # named_parameters() must yield the dotted names: a tiny shim
    model.named_parameters = lambda *a, **k: iter(plist.items())
    fake = object.__new__(Runner)
    fake.model = model
    fake.register_params()                                   # runner:1010-1031
    reg_names = sorted(fake.reg_params.keys())
    terms = {"importance": {n: [torch.from_numpy(a) for a in tensors[n][1]] for n in reg_names},
             "task_param": {n: [torch.from_numpy(a) for a in tensors[n][2]] for n in reg_names}}
    holder = type("M", (), {})()
    holder.loss = lambda *a, **k: {"loss_cls": torch.tensor(1.0)}
    hook = ref_runner.EWCHook(module=holder, reg_params=fake.reg_params, ewc_reg_terms=terms)
    res = hook()
    res["ewc_loss"].backward()
    out = {"ewc_loss": res["ewc_loss"].detach().numpy(), "n_reg": np.int64(len(reg_names))}
    for k, n in enumerate(reg_names):
        out[f"name_{k}"] = np.array(n)
        out[f"grad_{k}"] = fake.reg_params[n].grad.numpy()
    save("g6_ewc.npz", **out)


def run_task_split():
    """G7: the fork's task-split annotation rules, run through its own dataset classes (unbound methods on a
    namespace carrying exactly the attributes they read).  Stored as JSON."""
    import json
    import types
    import xml.etree.ElementTree as ET
    ref_xml = R.load("mmdet/datasets/xml_style_task.py")
    ref_coco = R.load("mmdet/datasets/coco_task.py")
    out = {"xml": [], "coco": []}
    docs = I.g7_xml_docs()
    for split, task_id in I.G7_SPLITS:
        for min_size in (None, 5):
            ns = types.SimpleNamespace(_metainfo={"classes": I.G7_VOC}, cat2label={c: i for i, c in enumerate(I.G7_VOC)},
                                       bbox_min_size=min_size, test_mode=False, task_split=split, task_id=task_id)
            per_doc = [ref_xml.XMLTask._parse_instance_info(ns, ET.ElementTree(ET.fromstring(d)), minus_one=True) for d in docs]
            ns.data_list = [dict(width=int(ET.fromstring(d).find("size/width").text), height=int(ET.fromstring(d).find("size/height").text),
                                 instances=inst, img_id=i) for i, (d, inst) in enumerate(zip(docs, per_doc)) if len(inst) != 0]
            ns.filter_cfg = dict(filter_empty_gt=True, min_size=250, bbox_min_size=min_size)
            kept = [d["img_id"] for d in ref_xml.XMLTask.filter_data(ns)]
            out["xml"].append(dict(task_split=split, task_id=task_id, bbox_min_size=min_size, instances=per_doc, kept_after_filter=kept))
    coco = I.g7_coco()
    cat_ids = [c["id"] for c in coco["categories"] if c["name"] in I.G7_COCO_CLASSES]      # COCO.get_cat_ids(cat_names=...)
    anns = {}
    cat_img_map = {cid: [] for cid in [c["id"] for c in coco["categories"]]}
    for a in coco["annotations"]:
        anns.setdefault(a["image_id"], []).append(a)
        cat_img_map[a["category_id"]].append(a["image_id"])
    for split, task_id in I.G7_COCO_SPLITS:
        ns = types.SimpleNamespace(data_prefix={"img": "imgs/"}, seg_map_suffix=".png", return_classes=False, test_mode=False,
                                   cat2label={cid: i for i, cid in enumerate(cat_ids)}, cat_img_map=cat_img_map,
                                   keep_cat=[cat_ids[i] for i in range(split[task_id - 1], split[task_id])])
        ns.data_list = []
        for img in coco["images"]:
            raw = dict(img)
            raw["img_id"] = img["id"]
            ns.data_list.append(ref_coco.CocoTaskDataset.parse_data_info(ns, dict(raw_ann_info=anns.get(img["id"], []), raw_img_info=raw)))
        ns.filter_cfg = dict(filter_empty_gt=True, min_size=32)
        kept = [d["img_id"] for d in ref_coco.CocoTaskDataset.filter_data(ns)]
        out["coco"].append(dict(task_split=split, task_id=task_id, data_list=ns.data_list, kept_after_filter=kept))
    path = os.path.join(HERE, "g7_task_split.json")
    json.dump(out, open(path, "w"))
    print(f"g7_task_split.json: {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------- G8 RoI dump (get_bbox_stuff)
def _load_stock_roi_parts():
    """The stock half of ``get_bbox_stuff`` is mmdet code the fork vendors: MaxIoUAssigner, RandomSampler, DeltaXYWHBBoxCoder,
    ``BBoxHead.get_targets``, ``bbox2roi``, ``unpack_gt_instances``.  Load those files for real (in dependency order) and hang them
    on the stand-in packages, so the reference's own ``get_bbox_stuff`` runs its own assign / sample / target code."""
    import importlib
    import re

    def expose(pkg, **names):
        m = importlib.import_module(pkg)
        for k, v in names.items():
            setattr(m, k, v)

    expose("mmengine.utils", digit_version=lambda v: tuple(int(x) for x in re.findall(r"\d+", v)[:3]))
    expose("mmdet.utils", util_mixins=R.load("mmdet/utils/util_mixins.py"), util_random=R.load("mmdet/utils/util_random.py"))
    overlaps = R.load("mmdet/structures/bbox/bbox_overlaps.py")
    expose("mmdet.structures.bbox", bbox_overlaps=overlaps.bbox_overlaps, get_box_tensor=lambda b: b,
           cat_boxes=lambda boxes, dim=0: torch.cat(boxes, dim=dim))
    ar = R.load("mmdet/models/task_modules/assigners/assign_result.py")
    R.load("mmdet/models/task_modules/assigners/base_assigner.py")
    iou = R.load("mmdet/models/task_modules/assigners/iou2d_calculator.py")
    assigner = R.load("mmdet/models/task_modules/assigners/max_iou_assigner.py")
    expose("mmdet.models.task_modules.assigners", AssignResult=ar.AssignResult)
    sr = R.load("mmdet/models/task_modules/samplers/sampling_result.py")
    R.load("mmdet/models/task_modules/samplers/base_sampler.py")
    sampler = R.load("mmdet/models/task_modules/samplers/random_sampler.py")
    expose("mmdet.models.task_modules.samplers", SamplingResult=sr.SamplingResult)
    R.load("mmdet/models/task_modules/coders/base_bbox_coder.py")
    coder = R.load("mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py")
    transforms = R.load("mmdet/structures/bbox/transforms.py")
    misc = R.load("mmdet/models/utils/misc.py")
    expose("mmdet.models.utils", multi_apply=misc.multi_apply, unpack_gt_instances=misc.unpack_gt_instances)
    bbox_head = R.load("mmdet/models/roi_heads/bbox_heads/bbox_head.py")

    class _Build:                       # TASK_UTILS.build(dict(type='BboxOverlaps2D')) inside MaxIoUAssigner.__init__
        def register_module(self, *a, **k):
            return lambda c: c

        def build(self, cfg, *a, **k):
            assert cfg["type"] == "BboxOverlaps2D"
            return iou.BboxOverlaps2D()
    assigner.TASK_UTILS = _Build()
    # the fork's head file was loaded against the stand-ins: give it the real helpers it calls in get_bbox_stuff
    ref_head.unpack_gt_instances = misc.unpack_gt_instances
    ref_head.bbox2roi = transforms.bbox2roi
    return assigner.MaxIoUAssigner, sampler.RandomSampler, coder.DeltaXYWHBBoxCoder, bbox_head.BBoxHead


class _Bag:
    """Attribute bag standing in for mmengine's InstanceData / DetDataSample (absent here)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def pop(self, k):
        return self.__dict__.pop(k)

    def __contains__(self, k):
        return k in self.__dict__

    def __len__(self):
        return int(next(iter(self.__dict__.values())).shape[0])


def run_roi_dump():
    from types import SimpleNamespace
    MaxIoUAssigner, RandomSampler, Coder, BBoxHead = _load_stock_roi_parts()
    out = {}
    for ci, (case, seed, _) in enumerate(I.G8_CASES):
        imgs = I.g8_case(ci)
        # rcnn train_cfg of _base_/models/faster-rcnn_r50_fpn.py:86-101
        assigner = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
        sampler = RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
        seen = []
        real_assign, real_sample = assigner.assign, sampler.sample

        def assign(pred, gt, ign=None):
            res = real_assign(pred, gt, ign)
            seen.append(dict(gt_inds=res.gt_inds.clone()))
            return res

        def sample(res, pred, gt, **kw):
            s = real_sample(res, pred, gt, **kw)
            seen[-1].update(pos_inds=s.pos_inds.clone(), neg_inds=s.neg_inds.clone())
            return s
        assigner.assign, sampler.sample = assign, sample
        head = SimpleNamespace(num_classes=I.G8_NUM_CLASSES, reg_decoded_bbox=False, bbox_coder=Coder(target_stds=[0.1, 0.1, 0.2, 0.2]))
        head._get_targets_single = lambda *a, **k: BBoxHead._get_targets_single(head, *a, **k)
        head.get_targets = lambda *a, **k: BBoxHead.get_targets(head, *a, **k)
        head.get_roi_targets = lambda sampling_results, rcnn_train_cfg: list(head.get_targets(sampling_results, rcnn_train_cfg))  # bbox_task:449-454
        head.get_mid_features = lambda f: f.flatten(1)                                          # bbox_task:290-323 without shared convs
        extractor = lambda x, rois: I.g8_extract(rois)
        extractor.num_inputs = 4
        self_ = SimpleNamespace(bbox_assigner=assigner, bbox_sampler=sampler, bbox_roi_extractor=extractor, bbox_head=head,
                                with_shared_head=False, train_cfg=SimpleNamespace(pos_weight=-1), counter=defaultdict(int))
        rpn = [_Bag(bboxes=torch.from_numpy(im["proposals"]), scores=torch.from_numpy(im["scores"])) for im in imgs]
        samples = [_Bag(metainfo={}, gt_instances=_Bag(bboxes=torch.from_numpy(im["gt_bboxes"]), labels=torch.from_numpy(im["gt_labels"])))
                   for im in imgs]
        x = [torch.zeros(len(imgs), 1, 1, 1)] * 4
        torch.manual_seed(seed)
        feats, cls_t, cls_w, box_t, box_w, rois = ref_head.StandardRoIReplayHead.get_bbox_stuff(self_, x, rpn, samples)
        assert feats.shape[0] == 5, feats.shape
        for i, s in enumerate(seen):
            for k, v in s.items():
                out[f"{case}__img{i}__{k}"] = v.numpy()
        for k, v in dict(feats=feats, cls_t=cls_t, cls_w=cls_w, box_t=box_t, box_w=box_w, rois=rois).items():
            out[f"{case}__{k}"] = v.numpy()
        print(f"  {case}: fg rows kept {(cls_t != I.G8_NUM_CLASSES).sum().item()}, positives per image "
              f"{[int(s['pos_inds'].numel()) for s in seen]}, negatives {[int(s['neg_inds'].numel()) for s in seen]}")
    # the RPN-side use of the same assigner (faster-rcnn_r50_fpn.py:56-63: 0.7 / 0.3 / 0.3, low-quality matches on)
    rpn_assigner = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1)
    for ci, (case, _, _) in enumerate(I.G8_CASES):
        for i, im in enumerate(I.g8_case(ci)):
            res = rpn_assigner.assign(_Bag(priors=torch.from_numpy(im["proposals"])),
                                      _Bag(bboxes=torch.from_numpy(im["gt_bboxes"]), labels=torch.from_numpy(im["gt_labels"])), None)
            out[f"{case}__img{i}__rpn_gt_inds"] = res.gt_inds.numpy()
    save("g8_roi_dump.npz", **out)


# ---------------------------------------------------------------- G9 teacher pseudo-labelling (det:65-109)
def run_pseudo_labels():
    """The reference's own ``FasterRCNNRoIReplay.loss`` with recording stand-ins for the teacher, the RPN head and the RoI head: what
    is pinned is the fork's loop (which predictions join the RPN set / the RoI set, the growing RoI set, label zeroing for the RPN).
    ``torchvision.ops.box_iou`` is absent from the image; the loop is given the reference tree's own ``bbox_overlaps``
    (mmdet/structures/bbox/bbox_overlaps.py: the same intersection-over-union) in its place, so the IoU ARITHMETIC of torchvision
    stays unpinned -- the inputs keep every IoU far from the 0.7 cut."""
    import copy
    import importlib
    from types import SimpleNamespace

    class Instances:
        """What the loop needs of mmengine's InstanceData (absent here), written out independently of the product's own stand-in:
        equally long tensors; ``len``; iteration yields one-row instances; ``inst['field']``; ``__delattr__``; ``cat`` of a list."""

        def __init__(self, **fields):
            object.__setattr__(self, "_f", dict(fields))

        def __getattr__(self, k):
            f = object.__getattribute__(self, "_f")
            if k in f:
                return f[k]
            raise AttributeError(k)

        def __setattr__(self, k, v):
            self._f[k] = v

        def __delattr__(self, k):
            del self._f[k]

        def __getitem__(self, k):
            return self._f[k] if isinstance(k, str) else Instances(**{n: v[k:k + 1] for n, v in self._f.items()})

        def __len__(self):
            return int(next(iter(self._f.values())).shape[0]) if self._f else 0

        def __iter__(self):
            return (self[i] for i in range(len(self)))

        def __deepcopy__(self, memo):
            return Instances(**{n: v.clone() for n, v in self._f.items()})

        def cat(self, items):
            return Instances(**{n: torch.cat([it._f[n] for it in items]) for n in items[0]._f})

    class DetSample:
        def __init__(self, gt_instances, img_shape=None):
            self.gt_instances, self.img_shape, self.pred_instances = gt_instances, img_shape, None

        def __deepcopy__(self, memo):
            return DetSample(copy.deepcopy(self.gt_instances, memo), self.img_shape)
    overlaps = sys.modules.get("mmdet.structures.bbox.bbox_overlaps") or R.load("mmdet/structures/bbox/bbox_overlaps.py")
    ref_det = R.load("mmdet/models/detectors/faster_rcnn_roi_replay.py")
    ref_det.box_iou = lambda a, b: overlaps.bbox_overlaps(a, b)
    out = {}
    for ti, (rpn_t, roi_t) in enumerate(I.G9_THRESHOLDS):
        imgs = I.g9_batch()
        seen = {}

        class Teacher:
            def eval(self):
                return self

            def predict(self, inputs, samples, rescale=False):
                for s, im in zip(samples, imgs):
                    s.pred_instances = Instances(bboxes=torch.from_numpy(im["pred_bboxes"]), scores=torch.from_numpy(im["pred_scores"]),
                                                 labels=torch.from_numpy(im["pred_labels"]))
                return samples

        def loss_and_predict(x, rpn_samples, proposal_cfg=None):
            seen["rpn"] = [(s.gt_instances.bboxes.clone(), s.gt_instances.labels.clone()) for s in rpn_samples]
            return {}, [None] * len(rpn_samples)

        def roi_loss(x, rpn_results, samples):
            seen["roi"] = [(s.gt_instances.bboxes.clone(), s.gt_instances.labels.clone()) for s in samples]
            return {}
        self_ = SimpleNamespace(extract_feat=lambda b: None, teacher_model=Teacher(), rpn_thresh=rpn_t, roi_thresh=roi_t, with_rpn=True,
                                train_cfg=dict(rpn_proposal=None), test_cfg=SimpleNamespace(rpn=None),
                                rpn_head=SimpleNamespace(loss_and_predict=loss_and_predict), roi_head=SimpleNamespace(loss=roi_loss))
        samples = [DetSample(Instances(bboxes=torch.from_numpy(im["gt_bboxes"]), labels=torch.from_numpy(im["gt_labels"])), img_shape=I.G8_CANVAS)
                   for im in imgs]
        ref_det.FasterRCNNRoIReplay.loss(self_, torch.zeros(len(imgs), 3, 8, 8), samples)
        for which in ("rpn", "roi"):
            for i, (b, l) in enumerate(seen[which]):
                out[f"t{ti}__{which}__img{i}__bboxes"] = b.numpy()
                out[f"t{ti}__{which}__img{i}__labels"] = l.numpy()
        print(f"  thresholds {rpn_t}/{roi_t}: RPN set sizes {[len(b) for b, _ in seen['rpn']]}, RoI set sizes {[len(b) for b, _ in seen['roi']]} "
              f"from gts {[len(im['gt_bboxes']) for im in imgs]} + predictions {[len(im['pred_bboxes']) for im in imgs]}")
    save("g9_pseudo_labels.npz", **out)

# ---------------------------------------------------------------- hand-off files
def run_handoff():
    """A task-1 work dir WRITTEN BY THE REFERENCE'S OWN CODE -- ``cal_fea_in`` (runner:705-763: hooks, compute_cov / update_cov, the
    torch.save of :757), ``cal_rois`` (runner:777-868), ``calculate_save_importance`` (runner:946-990) run as unbound methods on an
    object carrying exactly the attributes they read, plus the ``mask.pth`` the reference's task-2 head writes (head:451-452) -- under
    tests/golden/handoff/, and what the reference's task-2 start makes of those files (projectors + one SGDNSCL step, prototype bank,
    EWC loss) in expected.npz.  mmengine's dist helpers are absent: at world size 1 they are the identity."""
    import shutil
    import types
    Runner = ref_runner.BRNullSpaceRunner
    ref_runner.get_world_size, ref_runner.get_rank = (lambda: 1), (lambda: 0)
    ref_runner.all_reduce = lambda t, *a, **k: t
    root = os.path.join(HERE, "handoff")
    shutil.rmtree(root, ignore_errors=True)
    w1, w2 = os.path.join(root, "work_1"), os.path.join(root, "work_2")
    os.makedirs(w1)
    os.makedirs(w2)
    net = I.handoff_net()
    torch.save({"meta": {"note": "task-1 checkpoint of the hand-off net"}, "state_dict": net.state_dict()}, os.path.join(w1, "best_task1.pth"))
    net.data_preprocessor = lambda batch, training=False: batch
    roi_batches = I.handoff_roi_batches()

    def forward(self, inputs, data_samples=None, mode="tensor"):
        if mode == "roi_replay":
            return tuple(torch.from_numpy(a) for a in roi_batches[int(data_samples)])
        out = self.features(inputs)
        return {"loss": (out ** 2).mean()} if mode == "loss" else out
    net.forward = types.MethodType(forward, net)
    net._run_forward = lambda data, mode: net(data["inputs"], data["data_samples"], mode=mode)
    net.parse_losses = lambda losses: (sum(losses.values()), {})
    loader = [dict(inputs=torch.from_numpy(x), data_samples=b) for b, x in enumerate(I.handoff_images())]
    fake = object.__new__(Runner)
    log = types.SimpleNamespace(info=lambda *a, **k: None)
    fake.__dict__.update(logger=log, work_dir=w1, ckpt_keywords="best", load_or_resume=lambda: None, model=net, task_id=1,
                         ignore_keys=I.HANDOFF_IGNORE_KEYS + ["roi_head.bbox_head.fc_cls", "roi_head.bbox_head.fc_reg", "teacher"],   # runner:355
                         fea_in_save_path=os.path.join(w1, "covariance.pth"), fea_in_load_path=None, previous_dir=None,
                         reserve_per_class=0, ewc_reg_terms={}, train_dataloader=loader)
    Runner.cal_fea_in(fake, loader)
    fake.ckpt_keywords = "no-such-keyword"       # runner:782-786: `load_from` is only bound when the first entries do NOT match
    Runner.cal_rois(fake, loader)
    fake.ckpt_keywords = "best"
    fake.optim_wrapper = types.SimpleNamespace(scale_loss=lambda l: l, backward=lambda l: l.backward(), zero_grad=lambda: net.zero_grad(set_to_none=True))
    Runner.calculate_save_importance(fake, loader)
    net.train()
    # ---- what the reference's task-2 start makes of those files
    out = {}
    cov = torch.load(os.path.join(w1, "covariance.pth"), weights_only=False)        # a file this script wrote a second ago
    out["cov_type"] = np.array(type(cov).__name__)
    out["cov_keys"] = np.array(sorted(cov.keys()))
    names = [n for n, _ in net.named_parameters()]
    params = [nn.Parameter(p.detach().clone()) for _, p in net.named_parameters()]
    opt = ref_sgd.SGDNSCL(params, svd=True, **I.G1_HYPER["sgd"])
    opt.param_groups[0]["names"] = names
    import re
    fea_in = {k: v for k, v in cov.items() if not any(re.match(ik, k) for ik in fake.ignore_keys)}        # runner:643-650
    opt.get_eigens(fea_in)
    opt.get_transforms(offset=0.0)
    for n in names:
        if n in opt.transforms:
            key = n.replace(".", "_")
            mask = opt.adaptive_threshold(opt.eigens[n]["eigen_value"], offset=0.0)
            out[f"rank__{key}"] = np.int64(int(mask.to(torch.int8).argmax()))
            out[f"sigma__{key}"] = opt.eigens[n]["eigen_value"].numpy()
            out[f"P__{key}"] = opt.transforms[n].numpy()
    grads = I.handoff_step_inputs()
    for n, p in zip(names, params):
        p.grad = torch.from_numpy(grads[n].copy())
    opt.step()
    for n, p in zip(names, params):
        out[f"p_step0__{n.replace('.', '_')}"] = p.detach().numpy().copy()
    head = ref_head.StandardMultiPrototypeReplayHead(previous_path=w1, task_id=2, task_split=I.HANDOFF_TASK_SPLIT, max_prototype=10)
    out["bank"], out["labels"] = head.bbox_featss.numpy(), head.tmp_label.numpy()
    rois = torch.load(os.path.join(w1, "rois_etc.pth"), weights_only=True)
    for c in range(I.HANDOFF_TASK_SPLIT[0], I.HANDOFF_TASK_SPLIT[1]):
        Fc = rois[0][rois[1] == c].double()
        nrm = Fc / Fc.norm(dim=-1, keepdim=True)
        out[f"margin_{c}"] = np.float64(((nrm @ nrm.t()) - 0.6).abs().min().item())
        assert out[f"margin_{c}"] > 1e-5, "a pair sits within rounding of the 0.6 similarity cut: pick another seed"
    terms = torch.load(os.path.join(w1, "ewc_reg_terms_ewc.pth"), weights_only=False)    # ditto
    out["ewc_types"] = np.array([type(terms).__name__, type(terms["importance"]).__name__])
    theta = I.handoff_theta()
    reg = {n: nn.Parameter(torch.from_numpy(v.copy())) for n, v in theta.items()}
    holder = type("M", (), {})()
    holder.loss = lambda *a, **k: {"loss_cls": torch.tensor(1.0)}
    res = ref_runner.EWCHook(module=holder, reg_params=reg, ewc_reg_terms=terms)()
    res["ewc_loss"].backward()
    out["ewc_loss"] = res["ewc_loss"].detach().numpy()
    for n, p in reg.items():
        out[f"ewc_grad__{n.replace('.', '_')}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(root, "expected.npz"), **out)
    for d, _, files in os.walk(root):
        for f in files:
            print(f"  {os.path.relpath(os.path.join(d, f), HERE)}: {os.path.getsize(os.path.join(d, f)) / 1024:.1f} KiB")
    print("  covariance.pth:", out["cov_type"], list(out["cov_keys"]), "| ewc file:", list(out["ewc_types"]),
          "| ranks", {k[6:]: int(v) for k, v in out.items() if k.startswith("rank__")}, "| bank rows", out["bank"].shape[0])

def check_reverse_handoff():
    """Reverse direction of 8f-3 (build container only, nothing stored): files written by THIS PACKAGE's writers are read by the
    reference's own loaders -- ``update_optim_transforms`` (runner:635-662), the task-2 head constructor (head:397-452),
    ``load_importance`` (runner:996-999) + ``EWCHook`` -- and give what the reference made of its own files (handoff/expected.npz).
    The package's writers that are plain torch run here on the CPU: ``runner.nullspace.cal_rois`` and ``runner.ewc.save_importance``;
    its covariance pass needs the GPU, so the covariance file is written with the package's layout (a plain dict, what
    ``cal_fea_in`` saves) from the reference-computed matrices."""
    import tempfile
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from nsgp_repre_amd.runner import ewc as EWC
    from nsgp_repre_amd.runner import nullspace as NS
    from nsgp_repre_amd.runner.safe_load import load_handoff
    Runner = ref_runner.BRNullSpaceRunner
    ref_runner.get_world_size, ref_runner.get_rank = (lambda: 1), (lambda: 0)
    root = os.path.join(HERE, "handoff")
    E = np.load(os.path.join(root, "expected.npz"))
    log = types.SimpleNamespace(info=lambda *a, **k: None)
    with tempfile.TemporaryDirectory() as td:
        w1, w2 = os.path.join(td, "mine_1"), os.path.join(td, "mine_2")
        os.makedirs(w1)
        os.makedirs(w2)
        # --- the package's writers
        net = I.handoff_net()
        roi_batches = I.handoff_roi_batches()
        NS.cal_rois(net, range(len(roi_batches)), os.path.join(w1, "rois_etc.pth"), None, 1, 0, 20,
                    forward=lambda m, b: tuple(torch.from_numpy(a) for a in roi_batches[b]))
        terms = load_handoff(os.path.join(root, "work_1", "ewc_reg_terms_ewc.pth"))
        reg = {n: nn.Parameter(v[0].squeeze(0).clone()) for n, v in terms["task_param"].items()}
        EWC.save_importance(w1, {}, {n: v[0].squeeze(0).clone() for n, v in terms["importance"].items()}, reg)
        torch.save(dict(load_handoff(os.path.join(root, "work_1", "covariance.pth"))), os.path.join(w1, "covariance.pth"))
        # --- the reference's loaders
        params = [nn.Parameter(p.detach().clone()) for _, p in net.named_parameters()]
        opt = ref_sgd.SGDNSCL(params, svd=True, **I.G1_HYPER["sgd"])
        opt.param_groups[0]["names"] = [n for n, _ in net.named_parameters()]
        fake = object.__new__(Runner)
        fake.__dict__.update(logger=log, model=net, fea_in_load_path=os.path.join(w1, "covariance.pth"), offset=0.0, previous_dir=w1,
                             ignore_keys=I.HANDOFF_IGNORE_KEYS + ["roi_head.bbox_head.fc_cls", "roi_head.bbox_head.fc_reg", "teacher"],
                             optim_wrapper=types.SimpleNamespace(optimizer=opt))
        Runner.update_optim_transforms(fake, None)
        for n in opt.transforms:
            assert np.array_equal(opt.transforms[n].numpy(), E[f"P__{n.replace('.', '_')}"]), n
        head = ref_head.StandardMultiPrototypeReplayHead(previous_path=w1, task_id=2, task_split=I.HANDOFF_TASK_SPLIT, max_prototype=10)
        assert np.array_equal(head.bbox_featss.numpy(), E["bank"]) and np.array_equal(head.tmp_label.numpy(), E["labels"])
        Runner.load_importance(fake)
        theta = {n: nn.Parameter(torch.from_numpy(v.copy())) for n, v in I.handoff_theta().items()}
        holder = type("M", (), {})()
        holder.loss = lambda *a, **k: {"loss_cls": torch.tensor(1.0)}
        res = ref_runner.EWCHook(module=holder, reg_params=theta, ewc_reg_terms=fake.ewc_reg_terms)()
        assert np.array_equal(res["ewc_loss"].detach().numpy(), E["ewc_loss"])
    print("reverse hand-off ok: the reference's update_optim_transforms / head constructor / load_importance read the package's "
          "covariance.pth (layout) / rois_etc.pth / ewc_reg_terms_ewc.pth and reproduce expected.npz bit for bit")


if __name__ == "__main__":
    if sys.argv[1:] == ["--check"]:
        check_reverse_handoff()
        sys.exit(0)
    if sys.argv[1:] == ["handoff"]:
        run_handoff()
        sys.exit(0)
    if sys.argv[1:] == ["g1b"]:        # only the 128-aligned optimizer fixtures
        for kind in I.G1B_KINDS:
            run_optimizer_aligned(kind)
        sys.exit(0)
    if sys.argv[1:] == ["g1c"]:        # only the low-rank-basis optimizer fixtures
        for kind in I.G1C_KINDS:
            run_optimizer_lowrank_basis(kind)
        sys.exit(0)
    if sys.argv[1:] == ["g8"]:         # only the RoI-dump and pseudo-label fixtures
        run_roi_dump()
        run_pseudo_labels()
        sys.exit(0)
    for kind in ("sgd", "sgd_nesterov", "adamw", "adamw_amsgrad", "adam", "sgdna"):
        run_optimizer(kind)
    for kind in I.G1B_KINDS:
        run_optimizer_aligned(kind)
    for kind in I.G1C_KINDS:
        run_optimizer_lowrank_basis(kind)
    run_thresholds()
    run_covariance()
    run_prototypes()
    run_replay_loss()
    run_ewc()
    run_task_split()
    run_roi_dump()
    run_pseudo_labels()
    run_handoff()
