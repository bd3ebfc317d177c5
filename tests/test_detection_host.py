"""CPU tests of the stand-alone detector parts (nsgp_repre_amd.detection): the pieces that decide WHICH
parameters the NSGP side sees (names, shapes) and the box / assignment arithmetic of the stock recipe.
NMS and the whole-model passes need the GPU (tests/test_gpu_detector.py)."""
import copy
import math

import pytest
import torch

import nsgp_oracle as O
import nsgp_repre_amd as N
from nsgp_repre_amd import detection as D


@pytest.mark.parametrize("depth", [50, 101])
def test_parameter_names_and_shapes_are_the_projected_layer_table(depth):
    """The conv weights of backbone.layer2-4 + neck of the built model ARE the table bench.py and the oracle use
    (50 layers for R-50, 101 for R-101), name for name, shape for shape."""
    with torch.device("meta"):
        model = D.build_faster_rcnn(depth=depth)
    named = dict(model.named_parameters())
    table = O.resnet_fpn_projected_layers(depth)
    assert len(table) == {50: 50, 101: 101}[depth]
    for name, cout, d in table:
        w = named[name]
        assert w.requires_grad and w.shape[0] == cout and w[0].numel() == d, name
    # everything else that is trainable and 4-D lives under the ignore keys (rpn / roi_head) or is frozen
    ignore = N.runner.nullspace.full_ignore_keys(["rpn", "roi_head"])
    rest = [n for n, p in named.items() if p.dim() == 4 and p.requires_grad and n not in {t[0] for t in table}]
    assert all(N.runner.nullspace.should_ignore(n, ignore) for n in rest), rest
    frozen = [n for n, p in named.items() if not p.requires_grad]
    assert all(n.startswith(("backbone.conv1", "backbone.bn1", "backbone.layer1", "roi_head.bbox_head.fc_")) for n in frozen)


def test_hooked_modules_match_the_61_covariance_keys():
    with torch.device("meta"):
        model = D.build_faster_rcnn(depth=50)
    col = N.runner.nullspace.CovarianceCollector(model, N.runner.nullspace.full_ignore_keys(["rpn", "roi_head"]))
    # every module with a `.weight` is hooked (runner:731-732); only the convolutions contribute (runner:900-913)
    names = [n for n, m in col.hooked_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    assert len(names) == 61                                  # 61 keys for R-50-FPN
    assert "backbone.conv1" in names and "neck.fpn_convs.3.conv" in names and not any("rpn" in n or "roi_head" in n for n in names)


def test_anchor_grid_order_and_geometry():
    gen = D.AnchorGenerator()
    a = gen.grid([(2, 3), (1, 1)], "cpu")
    assert a[0].shape == (2 * 3 * 3, 4) and a[1].shape == (3, 4)
    # level 0: stride 4, scale 8 -> area 32^2 for every ratio; centred on (x*4, y*4), location-major
    wh = a[0][:, 2:] - a[0][:, :2]
    torch.testing.assert_close(wh[:, 0] * wh[:, 1], torch.full((18,), 1024.0))
    torch.testing.assert_close((wh[:, 1] / wh[:, 0])[:3], torch.tensor([0.5, 1.0, 2.0]))
    ctr = (a[0][:, :2] + a[0][:, 2:]) / 2
    torch.testing.assert_close(ctr[3 * 4], torch.tensor([4.0, 4.0]))          # location (y=1, x=1) -> index (1*3+1)*3


def test_delta_coder_round_trip_and_clip():
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(50, 2, generator=g) * 300
    rois = torch.cat([xy, xy + torch.rand(50, 2, generator=g) * 200 + 8], -1)
    xy2 = torch.rand(50, 2, generator=g) * 300
    gt = torch.cat([xy2, xy2 + torch.rand(50, 2, generator=g) * 200 + 8], -1)
    stds = (0.1, 0.1, 0.2, 0.2)
    back = D.delta2bbox(rois, D.bbox2delta(rois, gt, stds=stds), stds=stds)
    torch.testing.assert_close(back, gt, rtol=1e-4, atol=1e-2)
    clipped = D.delta2bbox(rois, torch.full((50, 4), 50.0), max_shape=(100, 120))
    assert clipped[:, 0::2].max() <= 120 and clipped[:, 1::2].max() <= 100 and clipped.min() >= 0


def test_max_iou_assignment_against_a_loop():
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(400, 2, generator=g) * 100
    priors = torch.cat([xy, xy + torch.rand(400, 2, generator=g) * 60 + 4], -1)
    gt = priors[torch.tensor([3, 77, 200])] + torch.tensor([2.0, -1.0, 3.0, 1.0])
    for pos, neg, mn, lowq in [(0.7, 0.3, 0.3, True), (0.5, 0.5, 0.5, False)]:
        got = D.assign_max_iou(priors, gt, pos, neg, mn, lowq)
        iou = D.box_iou(gt, priors)
        want = torch.full((400,), -1, dtype=torch.int64)
        for j in range(400):
            m, a = iou[:, j].max(0)
            if m < neg:
                want[j] = 0
            if m >= pos:
                want[j] = a + 1
        if lowq:
            for i in range(3):
                if iou[i].max() >= mn:
                    want[iou[i] == iou[i].max()] = i + 1
        assert torch.equal(got, want)
    assert torch.equal(D.assign_max_iou(priors, gt[:0], 0.7, 0.3, 0.3, True), torch.zeros(400, dtype=torch.int64))


def test_random_sampler_counts():
    assigned = torch.cat([torch.ones(300, dtype=torch.int64), torch.zeros(1000, dtype=torch.int64), -torch.ones(50, dtype=torch.int64)])
    pos, neg = D.random_sample(assigned, 512, 0.25)
    assert pos.numel() == 128 and neg.numel() == 384 and (assigned[pos] > 0).all() and (assigned[neg] == 0).all()
    pos, neg = D.random_sample(assigned[:20], 512, 0.25)
    assert pos.numel() == 20 and neg.numel() == 0
    assert len(set(pos.tolist())) == 20


def test_roi_align_on_a_linear_ramp():
    """Bilinear sampling of f(x, y) = 2x + 3y + 1 is exact, and the bin average of a linear function is its value
    at the bin centre: out[i, j] = f(centre of bin (i, j)) in feature-pixel coordinates (aligned=True)."""
    ext = D.RoIAlignExtractor()
    feats = []
    for s in (4, 8, 16, 32):
        h, w = 256 // s, 320 // s
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        feats.append((2 * xx + 3 * yy + 1)[None, None].repeat(2, 1, 1, 1))
    rois = torch.tensor([[0, 20.0, 30.0, 60.0, 70.0],          # scale 40  -> level 0
                         [1, 40.0, 40.0, 200.0, 168.0],        # scale 143 -> level 1
                         [1, 16.0, 16.0, 300.0, 240.0]])       # scale 252 -> level 2
    assert ext.map_levels(rois).tolist() == [0, 1, 2]
    out = ext(feats, rois)
    assert out.shape == (3, 1, 7, 7)
    for k, s in enumerate((4, 8, 16)):
        x1, y1, x2, y2 = (rois[k, 1:] / s).tolist()
        cx = x1 - 0.5 + (torch.arange(7) + 0.5) * (x2 - x1) / 7
        cy = y1 - 0.5 + (torch.arange(7) + 0.5) * (y2 - y1) / 7
        want = 2 * cx[None, :] + 3 * cy[:, None] + 1
        torch.testing.assert_close(out[k, 0], want, rtol=1e-4, atol=1e-3)


def test_instances_behave_like_instance_data():
    a = D.Instances(bboxes=torch.arange(12.).view(3, 4), labels=torch.tensor([1, 2, 3]), scores=torch.tensor([.9, .8, .7]))
    assert len(a) == 3 and len(a[torch.tensor([True, False, True])]) == 2 and len(a[:]) == 3
    one = [b for b in a][1]
    assert one.labels.tolist() == [2] and float(one["scores"]) == pytest.approx(0.8)
    b = a[:]
    b.__delattr__("scores")
    assert "scores" not in b and "scores" in a
    c = b.cat([b, b[:1]])
    assert len(c) == 4 and c.labels.tolist() == [1, 2, 3, 1]
    d = copy.deepcopy(a)
    d.labels[0] = 9
    assert a.labels[0] == 1
    s = D.DetSample(a, (10, 20))
    s2 = copy.deepcopy(s)
    s2.gt_instances.labels = torch.zeros_like(s2.gt_instances.labels)
    assert s.gt_instances.labels.tolist() == [1, 2, 3] and s2.img_shape == (10, 20)


def test_backbone_freezing_and_norm_eval():
    net = D.ResNet(50, frozen_stages=1, norm_eval=True, width=8).train()
    assert not net.conv1.weight.requires_grad and not net.layer1[0].conv1.weight.requires_grad
    assert net.layer2[0].conv1.weight.requires_grad and net.layer2[0].bn1.weight.requires_grad
    assert not any(m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    outs = D.FPN(net.out_channels, 16, 5)(net(torch.zeros(1, 3, 64, 96)))
    assert [tuple(o.shape[2:]) for o in outs] == [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]


def test_oracle_nms_on_a_hand_case():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6])
    assert O.nms_greedy(boxes, scores, 0.5).tolist() == [0, 2]
    assert O.nms_greedy(boxes, scores, 0.5, idxs=torch.tensor([0, 1, 0, 1])).tolist() == [0, 1, 2]     # [3] overlaps [1] in group 1
    assert O.nms_greedy(boxes, scores, 0.5, max_keep=1).tolist() == [0]
    assert O.nms_greedy(boxes, scores, 0.9).tolist() == [0, 1, 2]                                     # exact duplicate still goes
