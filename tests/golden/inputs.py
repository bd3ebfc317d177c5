"""Deterministic synthetic inputs shared by make_golden.py (which feeds them to
the reference) and the tests (which feed them to the oracle and the HIP path).
numpy ``default_rng`` streams are stable across platforms, so only seeds travel.
"""
import numpy as np

# ------------------------------------------------------------------ G1
G1_OFFSET = 0.0
G1_STEPS = 3
G1_HYPER = dict(
    sgd=dict(lr=0.02, momentum=0.9, weight_decay=1e-4),                     # schedule_1x_sgdnscl.py:21
    sgd_nesterov=dict(lr=0.02, momentum=0.9, dampening=0.1, nesterov=True, weight_decay=1e-4),
    adamw=dict(lr=1e-4, weight_decay=0.1),                                  # schedule_1x_sgdnscl.py:43 (commented AdamWNSCL)
    adamw_amsgrad=dict(lr=1e-3, weight_decay=0.05, amsgrad=True),
    adam=dict(lr=1e-3, weight_decay=1e-4),
    sgdna=dict(lr=0.02, momentum=0.9, weight_decay=1e-4, thres=50.0),
)

_G1 = [
    ("backbone.layer2.0.conv2.weight", (32, 32, 3, 3), True),
    ("backbone.layer2.0.conv1.weight", (24, 96, 1, 1), True),
    ("backbone.conv1.weight", (16, 3, 7, 7), True),
    ("neck.fpn_convs.0.conv.weight", (16, 36, 3, 3), True),
    ("neck.lateral_convs.0.conv.weight", (20, 160, 1, 1), True),
    ("roi_head.fc.weight", (12, 40), True),
    ("backbone.layer2.0.bn1.weight", (32,), False),
    ("neck.fpn_convs.0.conv.bias", (16,), False),
    ("rpn_head.rpn_conv.weight", (8, 8, 3, 3), False),
]


def g1_layers():
    return [n for n, _, _ in _G1], [s for _, s, _ in _G1]


def g1_projected():
    return [n for n, _, p in _G1 if p]


def g1_params():
    out = []
    for i, (_, shp, _) in enumerate(_G1):
        rng = np.random.default_rng(100 + i)
        out.append((rng.standard_normal(shp) * 0.02).astype(np.float32))
    return out


def g1_grads(step):
    out = []
    for i, (_, shp, _) in enumerate(_G1):
        rng = np.random.default_rng(1000 + 37 * step + i)
        out.append(rng.standard_normal(shp).astype(np.float32))
    return out


def covariance_like(D, seed, rows_mult=4):
    """C = X^T X, X = [rows_mult*D, D] ~ N(0,1) * diag(logspace(0,-3,D))."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((rows_mult * D, D)).astype(np.float32)
    X *= np.logspace(0, -3, D).astype(np.float32)[None, :]
    return (X.T @ X).astype(np.float32)


def g1_covariances():
    out = {}
    for i, (n, shp, proj) in enumerate(_G1):
        if proj:
            D = int(np.prod(shp[1:]))
            out[n] = covariance_like(D, 2000 + i)
    return out


# ------------------------------------------------------------------ G1b: 128-aligned layers
# Shapes the MFMA fast tiles and the split kernels take (rows % 128 == 0, D % 128 == 0) -- G1's layers all have Cout <= 32 and
# run the guarded generic tile.  The projected parameters start at ZERO, so after the first step they ARE the projected
# update (0 + u is exact, and weight decay adds nothing): the fixture pins the reference's `torch.mm(update, P)` itself, row by
# row, with no ulp(p) in the way.  The gradient rows span six decades: a per-row check then means something.
G1B_STEPS = 2
G1B_KINDS = ("sgd", "adamw")
_G1B = [
    ("neck.lateral_convs.1.conv.weight", (128, 128, 1, 1), True),
    ("backbone.layer3.0.conv1.weight", (256, 256, 1, 1), True),
    ("neck.fc.weight", (128, 256), True),
    ("backbone.layer3.0.bn1.weight", (256,), False),
]


def g1b_layers():
    return [n for n, _, _ in _G1B], [s for _, s, _ in _G1B]


def g1b_projected():
    return [n for n, _, p in _G1B if p]


def g1b_params():
    out = []
    for i, (_, shp, proj) in enumerate(_G1B):
        rng = np.random.default_rng(150 + i)
        out.append(np.zeros(shp, np.float32) if proj else (rng.standard_normal(shp) * 0.02).astype(np.float32))
    return out


def g1b_grads(step):
    out = []
    for i, (_, shp, proj) in enumerate(_G1B):
        rng = np.random.default_rng(1500 + 37 * step + i)
        g = rng.standard_normal(shp).astype(np.float32)
        if proj:    # row r scaled by 10^(-6 (r mod 32) / 31)
            rows = np.power(10.0, -6.0 * (np.arange(shp[0]) % 32) / 31.0).astype(np.float32)
            g *= rows.reshape((-1,) + (1,) * (len(shp) - 1))
        out.append(g)
    return out


def g1b_covariances():
    out = {}
    for i, (n, shp, proj) in enumerate(_G1B):
        if proj:
            out[n] = covariance_like(int(np.prod(shp[1:])), 2500 + i, rows_mult=2)
    return out


# ------------------------------------------------------------------ G2
G2_OFFSETS = [0.0, 0.3, -0.3, 2.0, 1.0, -1.0, -5.0]


def g2_spectra():
    out = []
    rng = np.random.default_rng(4242)
    for n in (16, 64, 100, 127, 128, 147, 256, 576, 1152, 2304):
        out.append(np.logspace(2, -4, n).astype(np.float32))
    for n in (96, 200, 600):  # plateau then cliff
        k = n // 3
        s = np.concatenate([np.full(k, 50.0), np.linspace(50, 1e-3, n - k)])
        out.append(s.astype(np.float32))
    for n in (120, 333, 1024):  # noisy exponential, sorted descending
        s = np.exp(-np.arange(n) / (n / 12.0)) * (1 + 0.2 * rng.random(n))
        out.append(np.sort(s)[::-1].astype(np.float32).copy())
    for n in (80, 288):  # ties / zeros in the tail
        s = np.logspace(1, -2, n)
        s[n // 2:] = 0.0
        s[5:9] = s[5]
        out.append(s.astype(np.float32))
    for D, seed in ((147, 7), (288, 8)):  # a real Gram spectrum
        C = covariance_like(D, seed).astype(np.float64)
        out.append(np.sort(np.linalg.eigvalsh(C))[::-1].astype(np.float32).copy())
    return out


# ------------------------------------------------------------------ G3
def g3_cases():
    return [
        dict(kind="conv", cin=8, k=(1, 1), s=(1, 1), p=(0, 0), hw=(12, 10), batch=1),
        dict(kind="conv", cin=6, k=(3, 3), s=(1, 1), p=(1, 1), hw=(9, 14), batch=2),
        dict(kind="conv", cin=5, k=(3, 3), s=(2, 2), p=(1, 1), hw=(13, 11), batch=2),
        dict(kind="conv", cin=3, k=(7, 7), s=(2, 2), p=(3, 3), hw=(20, 18), batch=1),
        dict(kind="conv", cin=4, k=(3, 1), s=(1, 2), p=(0, 1), hw=(8, 9), batch=3),
        dict(kind="linear", cin=24, batch=3),
    ]


def g3_inputs(ci, n_batches=2):
    cfg = g3_cases()[ci]
    out = []
    for b in range(n_batches):
        rng = np.random.default_rng(3000 + 10 * ci + b)
        if cfg["kind"] == "conv":
            shp = (cfg["batch"], cfg["cin"]) + cfg["hw"]
        else:
            shp = (cfg["batch"], cfg["cin"])
        out.append(np.abs(rng.standard_normal(shp)).astype(np.float32))
    return out


# ------------------------------------------------------------------ G4
G4_TASK_SPLIT = [0, 3, 5]
G4_MAX_PROTO = 10
G4_D = 7 * 7 * 256
G4_SEED = 6
# (rows, clusters) per class; classes 0-2 are the old task, 3-4 the new one
G4_CLASSES = ((45, 4), (64, 6), (150, 16), (20, 4), (11, 4))


def class_rois(n, d, seed, n_clusters=4, lo=0.25, hi=1.3):
    """One class's RoI features: ``n_clusters`` ReLU'd Gaussian centres + per-row
    noise of varying strength, ReLU'd (RoI features are post-ReLU, non-negative)."""
    rng = np.random.default_rng(seed)
    centres = np.maximum(rng.standard_normal((n_clusters, d)), 0).astype(np.float32)
    which = rng.integers(0, n_clusters, size=n)
    amp = rng.uniform(lo, hi, size=(n, 1)).astype(np.float32)
    x = centres[which] + amp * rng.standard_normal((n, d)).astype(np.float32)
    return np.maximum(x, 0).astype(np.float32)


def g4_rois(seed=G4_SEED, classes=G4_CLASSES, d=G4_D):
    feats, cls = [], []
    for c, (n, k) in enumerate(classes):
        feats.append(class_rois(n, d, seed * 100 + c, n_clusters=k))
        cls.append(np.full(n, c, dtype=np.int64))
    feats = np.concatenate(feats)
    cls = np.concatenate(cls)
    perm = np.random.default_rng(seed).permutation(len(cls))
    return feats[perm], cls[perm]


# ------------------------------------------------------------------ G5
G5_TASK_SPLIT = [0, 3, 5, 7]
G5_TASK_ID = 2
G5_IN = 7 * 7 * 4
G5_FC = 32


def g5_weights():
    rng = np.random.default_rng(77)

    def lin(o, i, std):
        return ((rng.standard_normal((o, i)) * std).astype(np.float32),
                (rng.standard_normal(o) * 0.1).astype(np.float32))
    w = dict(shared=[lin(G5_FC, G5_IN, 0.1), lin(G5_FC, G5_FC, 0.2)], cls=[], reg=[])
    for i in range(1, len(G5_TASK_SPLIT)):
        c = G5_TASK_SPLIT[i] - G5_TASK_SPLIT[i - 1]
        w["cls"].append(lin(c, G5_FC, 0.3))
        w["reg"].append(lin(4 * c, G5_FC, 0.05))
    w["cls"].append(lin(1, G5_FC, 0.3))  # background
    return w


def g5_bank(k=14):
    rng = np.random.default_rng(78)
    bank = np.maximum(rng.standard_normal((k, G5_IN)), 0).astype(np.float32)
    labels = rng.integers(0, G5_TASK_SPLIT[G5_TASK_ID - 1], size=k).astype(np.int64)
    return bank, labels


# ------------------------------------------------------------------ G6 (EWC regulariser)
G6_SHAPES = [("backbone.layer1.0.bn1.weight", (64,)), ("backbone.layer1.0.bn1.bias", (64,)),
             ("backbone.layer3.2.bn3.weight", (1031,)), ("backbone.layer4.0.bn2.bias", (512,)),
             ("backbone.bn1.weight", (7,)), ("teacher_model.backbone.bn1.weight", (7,)), ("neck.conv.weight", (4, 3, 1, 1))]
G6_TASKS = 2


def g6_tensors():
    """name -> (theta, [importance_t], [task_param_t]) ; only names with 'bn' and without
    'teacher_model' are regularised (runner:1010-1031)."""
    out = {}
    for i, (n, shp) in enumerate(G6_SHAPES):
        rng = np.random.default_rng(600 + i)
        theta = rng.standard_normal(shp).astype(np.float32)
        imp = [np.abs(rng.standard_normal((1,) + shp)).astype(np.float32) * 1e-2 for _ in range(G6_TASKS)]
        old = [(theta[None] + 0.1 * rng.standard_normal((1,) + shp)).astype(np.float32) for _ in range(G6_TASKS)]
        out[n] = (theta, imp, old)
    return out


# ----------------------------------------------------------------------------
# G7: task-split annotation parsing (xml_style_task.py / coco_task.py)
# ----------------------------------------------------------------------------
G7_VOC = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
          "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")
G7_SPLITS = [([0, 15, 20], 1), ([0, 15, 20], 2), ([0, 10, 20], 2), ([0, 5, 10, 15, 20], 3), ([0, 19, 20], 2)]


def g7_xml_docs(n_docs=12):
    """VOC-style XML strings: 0-6 objects, some unknown class names, difficult flags, boxes below bbox_min_size,
    float-formatted coordinates, one document without <difficult> tags."""
    rng = np.random.default_rng(700)
    docs = []
    for d in range(n_docs):
        w, h = int(rng.integers(200, 600)), int(rng.integers(200, 600))
        objs = []
        for k in range(int(rng.integers(0, 7))):
            name = "zebra" if rng.random() < 0.1 else G7_VOC[int(rng.integers(0, 20))]
            x1, y1 = int(rng.integers(1, w - 60)), int(rng.integers(1, h - 60))
            bw, bh = (int(rng.integers(1, 5)), int(rng.integers(1, 60))) if rng.random() < 0.2 else (int(rng.integers(5, 60)), int(rng.integers(5, 60)))
            fmt = (lambda v: f"{v}.0") if rng.random() < 0.3 else str
            diff = "" if d == 3 else f"<difficult>{int(rng.random() < 0.25)}</difficult>"
            objs.append(f"<object><name>{name}</name>{diff}<bndbox><xmin>{fmt(x1)}</xmin><ymin>{fmt(y1)}</ymin>"
                        f"<xmax>{fmt(x1 + bw)}</xmax><ymax>{fmt(y1 + bh)}</ymax></bndbox></object>")
        docs.append(f"<annotation><filename>{d:06d}.jpg</filename><size><width>{w}</width><height>{h}</height><depth>3</depth></size>"
                    + "".join(objs) + "</annotation>")
    return docs


def g7_coco():
    """A COCO-style dict: 12 categories with non-contiguous ids (two of them outside `classes`), 10 images,
    annotations incl. crowd, zero-area, out-of-image, ignore and sub-pixel boxes."""
    rng = np.random.default_rng(701)
    names = ["person", "bicycle", "car", "unicorn", "motorcycle", "airplane", "bus", "train", "dragon", "truck", "boat", "traffic light"]
    cats = [dict(id=3 * i + 1, name=n) for i, n in enumerate(names)]
    images = [dict(id=100 + i, file_name=f"{i:012d}.jpg", width=int(rng.integers(100, 640)), height=int(rng.integers(20, 480))) for i in range(10)]
    anns, aid = [], 1
    for img in images[:-1]:                      # the last image has no annotation at all
        for _ in range(int(rng.integers(1, 6))):
            c = cats[int(rng.integers(0, len(cats)))]
            x, y = float(rng.uniform(-20, img["width"] - 5)), float(rng.uniform(-20, img["height"] - 5))
            w, h = float(rng.uniform(0.5, 120)), float(rng.uniform(0.5, 120))
            a = dict(id=aid, image_id=img["id"], category_id=c["id"], bbox=[round(x, 2), round(y, 2), round(w, 2), round(h, 2)],
                     area=0.0 if rng.random() < 0.1 else round(w * h, 2), iscrowd=int(rng.random() < 0.15))
            if rng.random() < 0.1:
                a["ignore"] = True
            anns.append(a)
            aid += 1
    return dict(images=images, annotations=anns, categories=cats)


G7_COCO_CLASSES = ("person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light")
G7_COCO_SPLITS = [([0, 5, 10], 1), ([0, 5, 10], 2), ([0, 4, 8, 10], 3)]


# ---------------------------------------------------------------- G8: RoI dump (get_bbox_stuff, head:106-202)
G8_NUM_CLASSES = 20
G8_CANVAS = (800, 1344)                 # H, W of the padded input
G8_FEAT_C = 4                           # channels of the stand-in RoI extractor (mmcv's RoIAlign is absent)
# (name, torch seed, per image: (number of gts, number of jittered copies per gt, number of random proposals))
G8_CASES = (
    ("many_fg", 11, ((3, 12, 500), (4, 9, 480))),        # > 5 foreground rows: random foreground rows are dropped
    ("few_fg", 12, ((1, 0, 300), (0, 0, 250))),          # 1 foreground row (the gt itself) + an image without gts: padded with background
    ("pos_overflow", 13, ((5, 70, 700), (2, 10, 400))),  # > 128 positives in image 0: the sampler's random_choice on positives runs too
)


def g8_case(ci):
    """-> per image dict(gt_bboxes [G x 4], gt_labels [G], proposals [N x 4], scores [N]); all fp32 / int64."""
    name, seed, imgs = G8_CASES[ci]
    rng = np.random.default_rng(8000 + seed)
    H, W = G8_CANVAS
    out = []
    for (g, jit, nrand) in imgs:
        cx, cy = rng.uniform(100, W - 100, g), rng.uniform(100, H - 100, g)
        bw, bh = rng.uniform(40, 400, g), rng.uniform(40, 300, g)
        gt = np.stack([np.clip(cx - bw / 2, 0, W), np.clip(cy - bh / 2, 0, H), np.clip(cx + bw / 2, 0, W), np.clip(cy + bh / 2, 0, H)], 1)
        props = []
        for b in gt:
            w_, h_ = b[2] - b[0], b[3] - b[1]
            s = rng.uniform(0.02, 0.45, (jit, 1))                # small to large jitter: IoUs on both sides of 0.5
            d = rng.normal(0, 1, (jit, 4)) * s * np.array([w_, h_, w_, h_])
            props.append(b[None] + d)
        x1, y1 = rng.uniform(0, W - 20, nrand), rng.uniform(0, H - 20, nrand)
        props.append(np.stack([x1, y1, np.minimum(x1 + rng.uniform(8, 500, nrand), W), np.minimum(y1 + rng.uniform(8, 400, nrand), H)], 1))
        p = np.concatenate(props, 0) if props else np.zeros((0, 4))
        p = np.stack([np.clip(np.minimum(p[:, 0], p[:, 2]), 0, W), np.clip(np.minimum(p[:, 1], p[:, 3]), 0, H),
                      np.clip(np.maximum(p[:, 0], p[:, 2]), 0, W), np.clip(np.maximum(p[:, 1], p[:, 3]), 0, H)], 1)
        p = p[rng.permutation(p.shape[0])]
        out.append(dict(gt_bboxes=gt.astype(np.float32).reshape(-1, 4), gt_labels=rng.integers(0, G8_NUM_CLASSES, g).astype(np.int64),
                        proposals=p.astype(np.float32), scores=rng.uniform(0, 1, p.shape[0]).astype(np.float32)))
    return out


def g8_extract(rois, channels=G8_FEAT_C, out=7):
    """Stand-in for the RoI extractor (``bbox_roi_extractor(x, rois)``): a closed-form fp32 feature of the RoI's own
    coordinates, so that a row of the dumped features identifies the RoI it came from.  torch in, torch out."""
    import torch
    g = torch.arange(out, dtype=torch.float32, device=rois.device)
    c = torch.arange(1, channels + 1, dtype=torch.float32, device=rois.device)
    base = rois[:, 1:5] * rois.new_tensor([1 / 1344.0, 1 / 800.0, 1 / 1344.0, 1 / 800.0])       # [R x 4] in [0, 1]
    f = (base[:, 0, None, None, None] * c[None, :, None, None] + base[:, 1, None, None, None] * g[None, None, :, None] * 0.125
         + base[:, 2, None, None, None] * g[None, None, None, :] * 0.0625 + base[:, 3, None, None, None] + rois[:, 0, None, None, None])
    return f.contiguous()


# ---------------------------------------------------------------- G9: teacher pseudo-labelling (det:65-109)
G9_THRESHOLDS = ((0.5, 0.7), (0.5, 0.5), (0.3, 0.9))     # (rpn_thresh, roi_thresh): detector defaults (det:39-40), runner default (runner:356), a wide pair


def g9_batch(seed=9):
    """Two images: ground truth + what the teacher 'predicted' (boxes in score order as ``predict`` returns them, scores, old-class labels).
    Built so that every branch of the loop is taken: predictions that sit on a gt (IoU > 0.7: skipped), near-duplicates of an earlier
    prediction (skipped only if the earlier one was appended to the RoI set -- the set grows inside the loop), scores on both sides of
    both thresholds; the second image has no ground truth at all."""
    rng = np.random.default_rng(9000 + seed)
    H, W = G8_CANVAS
    out = []
    for g, n_free in ((3, 14), (0, 9)):
        cx, cy = rng.uniform(150, W - 150, g), rng.uniform(150, H - 150, g)
        bw, bh = rng.uniform(60, 300, g), rng.uniform(60, 300, g)
        gt = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).reshape(-1, 4)
        preds = [b + rng.normal(0, 2.0, 4) for b in gt]                                   # on a gt
        preds += [b + np.array([0.3, 0.2, -0.1, 0.1]) * (b[2] - b[0]) for b in gt]        # beside a gt (IoU around 0.5-0.6)
        x1, y1 = rng.uniform(0, W - 200, n_free), rng.uniform(0, H - 200, n_free)
        free = np.stack([x1, y1, x1 + rng.uniform(40, 190, n_free), y1 + rng.uniform(40, 190, n_free)], 1)
        preds += list(free)
        preds += [b + rng.normal(0, 1.5, 4) for b in free[: n_free // 2]]                 # near-duplicates of earlier predictions
        p = np.stack(preds).astype(np.float32)
        s = rng.uniform(0.05, 1.0, p.shape[0]).astype(np.float32)
        nd = n_free // 2
        s[-nd:] = s[-nd - n_free:-n_free][:nd] * np.float32(0.97)        # a duplicate scores just below its original: same side of most thresholds
        order = np.argsort(-s, kind="stable")
        out.append(dict(gt_bboxes=gt.astype(np.float32), gt_labels=rng.integers(15, 20, g).astype(np.int64),
                        pred_bboxes=p[order], pred_scores=s[order], pred_labels=rng.integers(0, 15, p.shape[0]).astype(np.int64)))
    return out


# ------------------------------------------------------------------ G1c: the low-rank step on the REFERENCE's own basis
# What G1b cannot reach: the default step applies `p += c (u - (u U) U^T)` from the r removed directions U (rank classes rpad = 32 / 64 /
# 128; layers wider than 256 columns are cut into K ranges whose slabs a reduce launch sums).  The fixture stores the reference's own
# U = eigen_vector[:, :r] (torch.svd, SGD_NSCL.py:377) for G1b's three layers plus two wide ones whose seeded spectrum puts the elbow in
# the two upper rank classes, and the reference's step() outputs; a test installs exactly that U (`set_basis`) and compares with nothing
# added.  Parameters start at zero and gradient rows span six decades, as in G1b.  P itself is not stored (only U [D x r]).
G1C_STEPS = dict(sgd=2, sgd_nesterov=1, adamw=1)
G1C_KINDS = ("sgd", "sgd_nesterov", "adamw")
_G1C = [
    ("neck.lateral_convs.1.conv.weight", (128, 128, 1, 1), True, None),
    ("backbone.layer3.0.conv1.weight", (256, 256, 1, 1), True, None),
    ("neck.fc.weight", (128, 256), True, None),
    ("neck.wide64.weight", (64, 1152), True, (40, 1.0)),          # elbow 48: rpad 64, 5 K ranges
    ("backbone.layer4.wide128.weight", (64, 256, 3, 3), True, (90, 1.0)),   # D = 2304, elbow 99: rpad 128, 9 K ranges; backbone -> / ||P||_F
    ("backbone.layer3.0.bn1.weight", (256,), False, None),
]


def g1c_layers():
    return [n for n, *_ in _G1C], [s for _, s, *_ in _G1C]


def g1c_projected():
    return [n for n, _, p, _ in _G1C if p]


def g1c_params():
    out = []
    for i, (_, shp, proj, _) in enumerate(_G1C):
        rng = np.random.default_rng(170 + i)
        out.append(np.zeros(shp, np.float32) if proj else (rng.standard_normal(shp) * 0.02).astype(np.float32))
    return out


def g1c_grads(step):
    out = []
    for i, (_, shp, proj, _) in enumerate(_G1C):
        rng = np.random.default_rng(1700 + 37 * step + i)
        g = rng.standard_normal(shp).astype(np.float32)
        if proj:    # row r scaled by 10^(-6 (r mod 32) / 31)
            rows = np.power(10.0, -6.0 * (np.arange(shp[0]) % 32) / 31.0).astype(np.float32)
            g *= rows.reshape((-1,) + (1,) * (len(shp) - 1))
        out.append(g)
    return out


def covariance_with_head(D, seed, head, gap, rows_mult=2):
    """C = X^T X with a flat head of ``head`` strong directions (column scales 1 ... 10^-0.1), a cliff of ``gap`` decades and a
    tail decaying to 10^-3, rotated by a seeded orthogonal matrix so that the eigenvectors are dense."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((rows_mult * D, D)).astype(np.float32)
    scale = np.concatenate([np.logspace(0, -0.1, head), np.logspace(-0.1 - gap, -3, D - head)]).astype(np.float32)
    X *= scale[None, :]
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)).astype(np.float32))
    X = X @ Q.astype(np.float32)
    return (X.T @ X).astype(np.float32)


def g1c_covariances():
    out = {}
    for i, (n, shp, proj, head) in enumerate(_G1C):
        if proj:
            D = int(np.prod(shp[1:]))
            out[n] = covariance_like(D, 2700 + i, rows_mult=2) if head is None else covariance_with_head(D, 2700 + i, *head)
    return out


# ------------------------------------------------------------------ hand-off files written by the reference's own code
# A task-1 work dir as the reference leaves it: covariance.pth (cal_fea_in, runner:705-763), rois_etc.pth (cal_rois, runner:777-868),
# ewc_reg_terms_ewc.pth (calculate_save_importance, runner:946-990), and the mask.pth its task-2 head writes (head:451-452).  The net
# is tiny (the files are the fixture); module names follow the detector's so that ignore_keys, 'backbone' and 'bn' rules all bite.
HANDOFF_TASK_SPLIT = [0, 3, 5]
HANDOFF_IGNORE_KEYS = ["rpn", "roi_head"]
HANDOFF_CLASSES = ((12, 3), (9, 2), (14, 4), (3, 1))     # (rows, clusters): classes 0-2 are the old task; class 3 stands for the rest
HANDOFF_BATCHES = 4


def handoff_net():
    """torch.nn.Module with backbone.conv1 / backbone.bn1 / backbone.layer1 / neck.conv / rpn_head.conv, seeded weights."""
    import torch
    import torch.nn as nn

    class Backbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 8, 3, stride=2, padding=1, bias=False)
            self.bn1 = nn.BatchNorm2d(8)
            self.layer1 = nn.Conv2d(8, 16, 1, bias=False)

        def forward(self, x):
            return self.layer1(torch.relu(self.bn1(self.conv1(x))))

    class Neck(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(16, 16, 3, padding=1)

        def forward(self, x):
            return self.conv(x)

    class RPN(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(16, 4, 1)

        def forward(self, x):
            return self.conv(x)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone, self.neck, self.rpn_head = Backbone(), Neck(), RPN()

        def features(self, x):
            return self.rpn_head(torch.relu(self.neck(torch.relu(self.backbone(x)))))

    net = Net()
    rng = np.random.default_rng(4100)
    with torch.no_grad():
        for _, p in sorted(net.named_parameters()):
            p.copy_(torch.from_numpy((rng.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
        net.backbone.bn1.running_mean.copy_(torch.from_numpy((rng.standard_normal(8) * 0.1).astype(np.float32)))
        net.backbone.bn1.running_var.copy_(torch.from_numpy((1.0 + 0.2 * rng.random(8)).astype(np.float32)))
    return net


def handoff_images():
    """HANDOFF_BATCHES input batches [2 x 3 x 20 x 28] (non-negative)."""
    return [np.abs(np.random.default_rng(4200 + b).standard_normal((2, 3, 20, 28))).astype(np.float32) for b in range(HANDOFF_BATCHES)]


def handoff_roi_batches():
    """What ``mode='roi_replay'`` returns per batch (the 6-tuple of get_bbox_stuff, head:106-202): rows of clustered RoI features
    [n x 12544] with their class targets, dealt over the batches; weights / boxes are seeded fillers of the right shapes."""
    feats, cls = [], []
    for c, (n, k) in enumerate(HANDOFF_CLASSES):
        feats.append(class_rois(n, G4_D, 4300 + c, n_clusters=k))
        cls.append(np.full(n, c if c < 3 else HANDOFF_TASK_SPLIT[-1], dtype=np.int64))      # class 3 = background label
    feats, cls = np.concatenate(feats), np.concatenate(cls)
    perm = np.random.default_rng(4400).permutation(len(cls))
    feats, cls = feats[perm], cls[perm]
    rng = np.random.default_rng(4500)
    out = []
    for idx in np.array_split(np.arange(len(cls)), HANDOFF_BATCHES):
        n = len(idx)
        out.append((feats[idx], cls[idx], np.ones(n, np.float32), rng.standard_normal((n, 4)).astype(np.float32),
                    (rng.random((n, 4)) < 0.5).astype(np.float32), np.concatenate([np.zeros((n, 1)), rng.uniform(0, 200, (n, 4))], 1).astype(np.float32)))
    return out


def handoff_step_inputs():
    """Seeded gradients for one optimizer step on the net's trainable tensors, in named_parameters() order."""
    import torch  # noqa: F401
    net = handoff_net()
    rng = np.random.default_rng(4600)
    return {n: rng.standard_normal(tuple(p.shape)).astype(np.float32) for n, p in net.named_parameters()}


def handoff_theta():
    """Task-2 values of the EWC-registered (BatchNorm) parameters the regulariser is evaluated at."""
    rng = np.random.default_rng(4700)
    return {"backbone.bn1.weight": (1.0 + 0.3 * rng.standard_normal(8)).astype(np.float32), "backbone.bn1.bias": (0.3 * rng.standard_normal(8)).astype(np.float32)}
