"""CPU tests of the runner / head host logic and of the N>1 exchange steps (gloo, world_size 2)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

import inputs as I
import nsgp_oracle as O


@pytest.fixture(scope="module")
def N():
    import nsgp_repre_amd
    return nsgp_repre_amd


class TinyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential()
        self.backbone.add_module("conv1", nn.Conv2d(3, 8, 3, padding=1))
        self.backbone.add_module("bn1", nn.BatchNorm2d(8))
        self.backbone.add_module("conv2", nn.Conv2d(8, 8, 3, stride=2, padding=1))
        self.neck = nn.Conv2d(8, 4, 1)
        self.rpn_head = nn.Conv2d(4, 4, 3, padding=1)
        self.roi_head = nn.Module()
        self.roi_head.bbox_head = nn.Module()
        self.roi_head.bbox_head.fc_cls = nn.Linear(4, 3)

    def forward(self, x):
        return self.rpn_head(self.neck(self.backbone(x)))


def test_registry_has_every_reference_name(N):
    r = N.registry
    assert r.RUNNERS.get("BRNullSpaceRunner") is N.runner.BRNullSpaceRunner
    for name in ("FasterRCNNRoIReplay", "StandardMultiPrototypeReplayHead", "StandardPrototypeReplayHead",
                 "StandardRoIReplayHead", "Shared2FCBBoxHeadTask", "ConvFCBBoxHeadTask"):
        assert r.MODELS.get(name) is not None, name


def test_wire_param_names_follows_named_parameters(N):
    net = TinyNet()
    net.backbone.conv1.weight.requires_grad_(False)          # frozen stage drops out of the optimizer
    bn = [p for n, p in net.named_parameters() if "bn" in n]
    rest = [p for n, p in net.named_parameters() if "bn" not in n]
    opt = N.SGDNSCL([dict(params=rest), dict(params=bn, weight_decay=0.0)], lr=0.1)
    N.runner.wire_param_names(opt, net)
    names0, names1 = opt.param_groups[0]["names"], opt.param_groups[1]["names"]
    assert "backbone.conv1.weight" not in names0 and "backbone.conv1.bias" in names0
    assert names1 == ["backbone.bn1.weight", "backbone.bn1.bias"]
    expected = [n for n, p in net.named_parameters() if p.requires_grad and "bn" not in n]
    assert names0 == expected
    for g in opt.param_groups:
        assert len(g["names"]) == len(g["params"])
        for n, p in zip(g["names"], g["params"]):
            assert dict(net.named_parameters())[n] is p


def test_ignore_keys_are_anchored_regexes(N):
    keys = N.runner.full_ignore_keys(["rpn", "roi_head"])
    assert keys[-3:] == ["roi_head.bbox_head.fc_cls", "roi_head.bbox_head.fc_reg", "teacher"]
    assert N.runner.should_ignore("rpn_head.rpn_conv.weight", keys)
    assert N.runner.should_ignore("roi_head.bbox_head.shared_fcs.0.weight", keys)
    assert N.runner.should_ignore("teacher_model.backbone.conv1.weight", keys)
    assert not N.runner.should_ignore("backbone.layer1.0.conv1.weight", keys)
    assert not N.runner.should_ignore("neck.rpn_like.weight", keys)      # re.match anchors at the start
    net = TinyNet()
    col = N.runner.CovarianceCollector(net, keys)
    hooked = [n for n, _ in col.hooked_modules()]
    assert hooked == ["backbone.conv1", "backbone.bn1", "backbone.conv2", "neck"]
    assert O.filter_ignore({"rpn_head.x.weight": 1, "neck.weight": 2}, keys) == {"neck.weight": 2}


def test_get_work_dir_and_find_checkpoint(N):
    from nsgp_repre_amd.roi_heads import get_work_dir
    assert get_work_dir("./work_dirs/ns3/cl_faster_rcnn_15_5_1") == "./work_dirs/ns3/cl_faster_rcnn_15_5_2"
    assert get_work_dir("./work_dirs/cl_coco/x_40_40_1") == "./"
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "best_pascal_voc_mAP_epoch_3.pth"), "w").close()
        assert N.runner.find_checkpoint(td, "best").endswith("best_pascal_voc_mAP_epoch_3.pth")
        with pytest.raises(FileNotFoundError):
            N.runner.find_checkpoint(td, "nothing")


def _g5_head(N, device="cpu"):
    w = I.g5_weights()
    head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=4, fc_out_channels=I.G5_FC, roi_feat_size=7, num_classes=7,
                                             task_split=I.G5_TASK_SPLIT, task_id=I.G5_TASK_ID)
    with torch.no_grad():
        for m, (W, b) in zip(head.shared_fcs, w["shared"]):
            m.weight.copy_(torch.from_numpy(W)); m.bias.copy_(torch.from_numpy(b))
        for m, (W, b) in zip(head.fc_cls, w["cls"]):
            m.weight.copy_(torch.from_numpy(W)); m.bias.copy_(torch.from_numpy(b))
        for m, (W, b) in zip(head.fc_reg, w["reg"]):
            m.weight.copy_(torch.from_numpy(W)); m.bias.copy_(torch.from_numpy(b))
    return head.to(device)


def test_task_head_forward_vs_reference_golden(N, golden_dir):
    """Shared2FCBBoxHeadTask is plain torch modules: its forward against the reference's own outputs (G5) on the CPU.
    The replay LOSS is a HIP kernel and has no CPU route (checked on the GPU in test_gpu_runner.py)."""
    g = np.load(os.path.join(golden_dir, "g5_replay_loss.npz"))
    head = _g5_head(N)
    # the future task's heads are frozen, background stays trainable
    assert [m.weight.requires_grad for m in head.fc_cls] == [True, True, False, True]
    assert [m.weight.requires_grad for m in head.fc_reg] == [True, True, False]
    bank, labels = I.g5_bank()
    feats = torch.from_numpy(bank).reshape(-1, 4, 7, 7)
    cls_score, bbox_pred = head(feats)
    score = cls_score.detach().numpy()
    np.testing.assert_array_equal(np.isinf(score), np.isinf(g["cls_score"]))
    fin = np.isfinite(g["cls_score"])
    np.testing.assert_allclose(score[fin], g["cls_score"][fin], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bbox_pred.detach().numpy(), g["bbox_pred"], rtol=1e-5, atol=1e-6)

    class Holder(N.roi_heads.PrototypeReplay):
        pass
    h = Holder()
    h.bbox_head, h.task_split, h.task_id = head, I.G5_TASK_SPLIT, I.G5_TASK_ID
    h.tmp_label, h.replay, h.bbox_featss = torch.from_numpy(labels), True, feats
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        h.replay_loss(h.bbox_featss)


def test_select_five_rois_keeps_exactly_five(N):
    """head:163-199: foreground rows kept, padded with random background / trimmed at random to 5."""
    from nsgp_repre_amd.roi_heads import select_five_rois
    bg = 20
    torch.manual_seed(0)
    for n_fg, n_bg in ((0, 30), (2, 30), (5, 30), (9, 30), (3, 1), (0, 3)):
        cls = torch.cat([torch.randint(0, bg, (n_fg,)), torch.full((n_bg,), bg)])[torch.randperm(n_fg + n_bg)]
        m = select_five_rois(cls.clone(), bg)
        assert int(m.sum()) == min(5, n_fg + n_bg)
        fg = cls != bg
        if n_fg <= 5:
            assert bool((m | ~fg).all())            # every foreground row survives
        else:
            assert bool((~m | fg).all())            # only foreground rows survive


def test_detector_mode_dispatch(N):
    calls = []

    class Det(N.detectors.RoIReplayModes):
        def loss(self, a, b, use_teacher_student=True):
            calls.append(("loss", use_teacher_student))
        def predict(self, a, b):
            calls.append(("predict",))
        def _forward(self, a, b):
            calls.append(("tensor",))
        def get_bbox_stuff(self, a, b):
            calls.append(("roi",))
    d = Det()
    for m in ("loss", "predict", "tensor", "nullspace", "roi_replay"):
        d.forward(None, None, mode=m)
    assert calls == [("loss", True), ("predict",), ("tensor",), ("loss", False), ("roi",)]
    with pytest.raises(RuntimeError):
        d.forward(None, None, mode="bogus")


def test_shard_by_cost_balances(N):
    owner = N.runner.dist.shard_by_cost([d ** 3 for _, _, d in O.resnet_fpn_projected_layers(50)], 8)
    assert len(owner) == 50 and set(owner) == set(range(8))
    load = [0.0] * 8
    for o, (_, _, d) in zip(owner, O.resnet_fpn_projected_layers(50)):
        load[o] += d ** 3
    # three indivisible 4608^3 decompositions dominate: the optimum is one of them alone on a rank
    assert max(load) == 4608.0 ** 3


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nsgp_repre_amd as N
    D = N.runner.dist
    g = torch.Generator().manual_seed(100 + rank)
    local = {"b.weight": torch.randn(6, 6, generator=g), "a.weight": torch.randn(3, 3, generator=g)}
    mine = {k: v.clone() for k, v in local.items()}
    D.all_reduce_dict(mine)                                   # C1
    t = torch.arange((rank + 2) * 3, dtype=torch.float32).reshape(rank + 2, 3) + 100 * rank
    parts = D.all_gather_different_shape(t)                   # C2, ragged first dim
    empty = D.all_gather_different_shape(torch.zeros(0, 5) if rank == 0 else torch.ones(2, 5))
    # C4: every rank ends with every layer's spectrum, bit-identical to a single-process decomposition
    shapes = {"backbone.l1.weight": 24, "backbone.l2.weight": 40, "neck.l3.weight": 16, "neck.l4.weight": 40, "neck.l5.weight": 8}
    params = [torch.nn.Parameter(torch.zeros(4, d)) for d in shapes.values()]
    fea = {}
    for i, (n, d) in enumerate(shapes.items()):
        x = torch.randn(3 * d, d, generator=torch.Generator().manual_seed(7 + i))
        fea[n] = x.t() @ x
    opt = N.SGDNSCL(params, lr=0.1, svd=True)
    opt.param_groups[0]["names"] = list(shapes)
    owner = D.sharded_eigens(opt, fea)
    solo = N.SGDNSCL(params, lr=0.1, svd=True)
    solo.param_groups[0]["names"] = list(shapes)
    solo.get_eigens(fea)
    same = all(torch.equal(opt.eigens[n]["eigen_value"], solo.eigens[n]["eigen_value"])
               and torch.equal(opt.eigens[n]["eigen_vector"], solo.eigens[n]["eigen_vector"]) for n in shapes)
    q.put((rank, {k: v.numpy() for k, v in local.items()}, {k: v.numpy() for k, v in mine.items()},
           [p.numpy() for p in parts], [e.shape for e in empty], owner, same))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_steps_gloo_world2(N):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    locals_ = [{k: torch.from_numpy(v) for k, v in g[1].items()} for g in got]
    ref = O.all_reduce_dict_sum(locals_)
    for rank, _, reduced, parts, eshapes, owner, same in got:
        assert same and sorted(set(owner)) == [0, 1] and owner == got[0][5]     # both ranks work, and agree on who owns what
        for k in ref:
            np.testing.assert_allclose(reduced[k], ref[k].numpy(), rtol=1e-6)
        exp = O.all_gather_different_shape([torch.arange((r + 2) * 3, dtype=torch.float32).reshape(r + 2, 3) + 100 * r for r in range(2)])
        assert len(parts) == 2
        for a, b in zip(parts, exp):
            np.testing.assert_array_equal(a, b.numpy())
        assert [tuple(s) for s in eshapes] == [(0, 5), (2, 5)]


def test_exchange_steps_are_identity_without_process_group(N):
    D = N.runner.dist
    d = {"x": torch.ones(2, 2)}
    D.all_reduce_dict(d)
    assert torch.equal(d["x"], torch.ones(2, 2))
    t = torch.arange(6.0).reshape(3, 2)
    assert len(D.all_gather_different_shape(t)) == 1 and D.get_world_size() == 1 and D.get_rank() == 0


def test_task_flow_pieces_of_the_mixin(N):
    """Host side of the reference's train() additions that need no GPU: rr_thresh wiring (runner:439-441), checkpoints by
    keyword (runner:295-299, 710-716), the EWC importance pass and its file (runner:946-990)."""
    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(2, 3, 1)
            self.bn1 = nn.BatchNorm2d(3)

        def forward(self, x):
            return self.bn1(self.conv(x)).square().mean()

    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as td:
        w1, w2 = os.path.join(td, "t_1"), os.path.join(td, "t_2")
        os.makedirs(w1), os.makedirs(w2)
        mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
        mix.init_task_state(w1, task_id=1, rr_thresh=[0.5, 0.7], ckpt_keywords="best")
        net = Tiny()
        mix.set_pseudo_label_thresholds(net)
        assert (net.rpn_thresh, net.roi_thresh) == (0.5, 0.7)
        assert mix.reload_task_checkpoint(net) is None                     # nothing saved yet
        mix.save_checkpoint(net, "best_x.pth")
        saved = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            net.conv.weight.add_(1.0)
        assert mix.reload_task_checkpoint(net).endswith("best_x.pth")
        assert all(torch.equal(v, saved[k]) for k, v in net.state_dict().items())
        # importance: F = sum_b grad_b^2 * len(batch)/len(loader), only the "bn" parameters, in eval mode
        data = [torch.randn(4, 2, 5, 5) for _ in range(3)]
        terms = mix.calculate_save_importance(net, data, lambda m, b: m(b))
        assert sorted(terms["importance"]) == ["bn1.bias", "bn1.weight"] and not net.training
        want = torch.zeros(3)
        for b in data:
            net.zero_grad()
            net(b).backward()
            want += net.bn1.weight.grad ** 2 * (len(b) / len(data))
        assert torch.allclose(terms["importance"]["bn1.weight"][0][0], want, rtol=1e-6, atol=1e-12)
        disk = torch.load(os.path.join(w1, "ewc_reg_terms_ewc.pth"), weights_only=True)
        assert torch.equal(disk["task_param"]["bn1.bias"][0][0], net.bn1.bias.detach())
        # the next task finds the previous directory's checkpoint and importance file
        nxt = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
        nxt.init_task_state(w2, task_id=2, previous_dir=w1, ckpt_keywords="best")
        net2 = Tiny()
        assert nxt.load_previous_checkpoint(net2).endswith("best_x.pth")
        assert torch.equal(net2.conv.weight, net.conv.weight)
        assert nxt.rr_thresh == [0.5, 0.5]                                   # runner:356 default
        assert (nxt.cov_grouped, nxt.cov_streams) == (True, 4)               # how cal_fea_in issues the covariance launches: the defaults ...
        alt = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
        alt.init_task_state(w2, task_id=2, previous_dir=w1, cov_grouped=False, cov_streams=1)
        assert (alt.cov_grouped, alt.cov_streams) == (False, 1)              # ... and the reference's own order of operations (per hook, one stream)
        loaded = nxt.load_importance(net2)
        assert sorted(nxt.reg_params) == ["bn1.bias", "bn1.weight"] and len(loaded["importance"]["bn1.weight"]) == 1


def test_general_convfc_task_head(N):
    """``ConvFCBBoxHeadTask`` beyond the Shared2FC shape (convfc_bbox_head_task.py:14-288): shared convs, separate cls / reg
    conv + fc branches, the reference's layer names and dimension rules, future-task logits at -inf; unknown keywords raise."""
    H = N.roi_heads.ConvFCBBoxHeadTask
    head = H(num_shared_convs=1, num_cls_convs=1, num_cls_fcs=1, num_reg_fcs=2, conv_out_channels=6, fc_out_channels=10,
             in_channels=4, roi_feat_size=3, num_classes=6, task_split=[0, 2, 4, 6], task_id=2)
    names = [n for n, _ in head.named_parameters()]
    assert "shared_convs.0.conv.weight" in names and "cls_convs.0.conv.weight" in names and "cls_fcs.0.weight" in names
    assert "reg_fcs.1.bias" in names and len(head.fc_cls) == 4 and len(head.fc_reg) == 3
    assert head.cls_fcs[0].in_features == 6 * 9 and head.reg_fcs[0].in_features == 6 * 9 and head.fc_reg[0].out_features == 8
    # heads of the task that has not arrived are frozen, the background head is not (:130-144)
    assert [m.weight.requires_grad for m in head.fc_cls] == [True, True, False, True]
    assert [m.weight.requires_grad for m in head.fc_reg] == [True, True, False]
    x = torch.randn(5, 4, 3, 3)
    cls, reg = head(x)
    assert cls.shape == (5, 7) and reg.shape == (5, 24)
    assert torch.isinf(cls[:, 4:6]).all() and (cls[:, 4:6] < 0).all() and torch.isfinite(cls[:, :4]).all() and torch.isfinite(cls[:, 6]).all()
    assert (reg[:, 16:] == 0).all()
    assert head.get_mid_features(x).shape == (5, 6 * 9)
    # no fcs in a branch: its predictor sees the flattened map (:119-123)
    h2 = H(num_shared_convs=1, conv_out_channels=6, in_channels=4, roi_feat_size=3, num_classes=4, task_split=[0, 2, 4], task_id=1)
    assert h2.fc_cls[0].in_features == 6 * 9 and h2(x)[0].shape == (5, 5)
    with pytest.raises(AssertionError):
        H(num_cls_convs=1, num_shared_fcs=1)                           # :89-90
    with pytest.raises(AssertionError):
        H()                                                            # :87-88
    with pytest.raises(TypeError):
        H(num_shared_fcs=1, loss_clz=dict(type="CrossEntropyLoss"))    # unknown keyword: not swallowed
    s2 = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=4, fc_out_channels=8, roi_feat_size=3, num_classes=4, task_split=[0, 2, 4], task_id=2)
    assert isinstance(s2, H) and [n for n, _ in s2.named_parameters()][:2] == ["shared_fcs.0.weight", "shared_fcs.0.bias"]


def _oracle_select(Fc, max_proto=10, saved_masks=None, order=None):
    co, fine, masks, cen, _ = O.prototype_select(Fc, max_proto, order=order, saved_masks=saved_masks)
    return co, fine, masks, cen


def _bank_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    feats, cls_t = _bank_inputs()
    bank, labels, masks, centres = build_prototype_bank(feats, cls_t, [0, 5, 7], 2, 6, select_fn=_oracle_select)
    q.put((rank, bank.numpy(), labels.numpy(), [[m.numpy() for m in ml] for ml in masks], centres))
    dist.barrier()
    dist.destroy_process_group()


def _bank_inputs():
    """Five old classes of very different sizes (cost N_c^2: 120^2 dominates) + two new ones."""
    sizes = (120, 30, 45, 20, 60, 10, 10)
    feats = torch.cat([torch.from_numpy(I.class_rois(n, 96, seed=40 + c, n_clusters=3)) for c, n in enumerate(sizes)])
    cls_t = torch.cat([torch.full((n,), c, dtype=torch.long) for c, n in enumerate(sizes)])
    perm = torch.randperm(feats.shape[0], generator=torch.Generator().manual_seed(1))      # classes interleaved, as in rois_etc.pth
    return feats[perm].contiguous(), cls_t[perm].contiguous()


def test_class_sharded_prototype_bank_gloo_world2(N):
    """SURVEY 8e / VERDICT r1 e5: the bank build sharded by class over two ranks (cost N_c^2) + all-gather equals the
    single-process build bit for bit on both ranks (bank rows, labels, masks, centre ids)."""
    import torch.multiprocessing as mp
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    feats, cls_t = _bank_inputs()
    bank, labels, masks, centres = build_prototype_bank(feats, cls_t, [0, 5, 7], 2, 6, select_fn=_oracle_select)
    assert sorted(set(labels.tolist())) == [0, 1, 2, 3, 4] and bank.shape[0] == labels.shape[0] > 5
    owner = N.runner.dist.shard_by_cost([float(n) ** 2 for n in (120, 30, 45, 20, 60)], 2)
    assert owner[0] != owner[4] and sorted(set(owner)) == [0, 1]       # the 120-RoI class alone on one rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, b, l, m, c in got:
        np.testing.assert_array_equal(b, bank.numpy())
        np.testing.assert_array_equal(l, labels.numpy())
        assert c == centres
        assert all(np.array_equal(x, y.numpy()) for ml, rl in zip(m, masks) for x, y in zip(ml, rl))


def test_roi_dump_vs_reference_golden(N, golden_dir):
    """a13, the whole of ``get_bbox_stuff`` (head:106-202) against the reference's own output (G8): assignment, the sampler's
    seeded draws, RoI order, targets and the exactly-five selection."""
    from roi_dump_check import check_roi_dump
    check_roi_dump(N, golden_dir, "cpu")
