"""GPU tests of the end-to-end route: NMS kernel vs the oracle, the stand-alone Faster R-CNN through all five
modes, and a complete two-task NSGP-RePRE cycle on it (task 1 -> covariance.pth + rois_etc.pth -> task 2 with
teacher pseudo-labels, prototype bank, projected optimizer step)."""
import copy
import os
import tempfile

import pytest
import torch

import nsgp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    import nsgp_repre_amd
    assert torch.cuda.is_available()
    return nsgp_repre_amd


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _clustered_boxes(n, seed, span=400.0):
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(max(n // 12, 1), 2, generator=g) * span
    c = centres[torch.randint(0, centres.shape[0], (n,), generator=g)] + torch.randn(n, 2, generator=g) * 6
    wh = torch.rand(n, 2, generator=g) * 40 + 20
    return torch.cat([c - wh / 2, c + wh / 2], -1), torch.rand(n, generator=g)


@pytest.mark.parametrize("n,thr,groups,max_keep", [(1, 0.5, 0, None), (64, 0.5, 0, None), (65, 0.7, 3, None), (1000, 0.5, 20, 100),
                                                    (3000, 0.7, 5, 1000), (3000, 0.3, 0, None)])
def test_nms_is_bit_exact_against_the_oracle(N, dev, n, thr, groups, max_keep):
    boxes, scores = _clustered_boxes(n, seed=n + groups)
    scores[: n // 3] = scores[: n // 3].round(decimals=1)          # ties: the stable order decides
    idxs = torch.randint(0, groups, (n,), generator=torch.Generator().manual_seed(7)) if groups else None
    want = O.nms_greedy(boxes, scores, thr, idxs, max_keep)
    got = N.ops.nms(boxes.to(dev), scores.to(dev), thr, None if idxs is None else idxs.to(dev), max_keep)
    assert torch.equal(got.cpu(), want)
    assert 0 < got.numel() < n or n == 1


def test_nms_empty_and_limits(N, dev):
    assert N.ops.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5).numel() == 0
    b, s = _clustered_boxes(10, 0)
    assert N.ops.nms(b.to(dev), s.to(dev), 0.5, max_keep=3).numel() == 3
    with pytest.raises(RuntimeError):
        N.ops.nms(torch.zeros(70000, 4, device=dev), torch.zeros(70000, device=dev), 0.5)


def _batches(dev, n, classes, seed, h=192, w=256):
    """One image per batch, five boxes whose labels cycle through ``classes`` so that every class shows up."""
    from nsgp_repre_amd.detection import DetSample, Instances
    g = torch.Generator().manual_seed(seed)
    lo, hi = classes
    out = []
    for k in range(n):
        x = torch.rand(1, 3, h, w, generator=g).to(dev)
        wh = torch.rand(5, 2, generator=g) * torch.tensor([w * 0.3, h * 0.3]) + 24
        xy = torch.rand(5, 2, generator=g) * (torch.tensor([w, h]) - wh)
        labels = (torch.arange(5) + 5 * k) % (hi - lo) + lo
        out.append((x, [DetSample(Instances(bboxes=torch.cat([xy, xy + wh], -1).to(dev), labels=labels.to(dev)), img_shape=(h, w))]))
    return out


def test_detector_modes(N, dev):
    from nsgp_repre_amd.detection import build_faster_rcnn, relocate_segment_final_weights
    torch.manual_seed(0)
    model = build_faster_rcnn(width=16, fc_out_channels=64, task_id=1).to(dev).train()
    relocate_segment_final_weights(model)
    x, samples = _batches(dev, 1, (0, 15), 0)[0]
    losses = model(x, copy.deepcopy(samples), mode="loss")
    assert set(losses) == {"loss_rpn_cls", "loss_rpn_bbox", "loss_cls", "loss_bbox", "acc"}
    assert all(torch.isfinite(v) for v in losses.values())
    sum(v for k, v in losses.items() if "loss" in k).backward()
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)
    ns = model(x, copy.deepcopy(samples), mode="nullspace")             # the same pass without the teacher
    assert set(ns) == set(losses)
    model.eval()
    preds = model(x, copy.deepcopy(samples), mode="predict")
    inst = preds[0].pred_instances
    assert 0 < len(inst) <= 100 and (inst.scores[:-1] >= inst.scores[1:]).all() and inst.labels.max() < 15
    stuff = model(x, copy.deepcopy(samples), mode="roi_replay")
    assert [tuple(t.shape) for t in stuff] == [(5, 12544), (5,), (5,), (5, 4), (5, 4), (5, 5)]
    with pytest.raises(RuntimeError):
        model(x, samples, mode="bogus")


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_two_task_cycle_on_the_detector(N, dev):
    """The reference's whole flow on a narrow R-50-FPN: task 1 on classes 0-14 (3 steps), end-of-task covariance
    pass under the hooks + RoI dump; task 2 on classes 15-19 with the teacher, the prototype bank from task 1's
    files and NSGP-projected steps.  Checks the hand-off files, the loss dict, and the defining property of NSGP
    on a real backbone layer: the weight change annihilates the old task's dominant input directions."""
    from nsgp_repre_amd.detection import build_faster_rcnn, relocate_segment_final_weights
    torch.manual_seed(3)
    split = [0, 15, 20]
    ignore = ["rpn", "roi_head"]

    def step_fn(model, batch):
        losses = model(batch[0], copy.deepcopy(batch[1]), mode="loss")
        step_fn.last = losses
        return sum(v for k, v in losses.items() if "loss" in k)

    def cov_fwd(model, batch):
        model(batch[0], copy.deepcopy(batch[1]), mode="nullspace")

    def roi_fwd(model, batch):
        return model(batch[0], copy.deepcopy(batch[1]), mode="roi_replay")

    with tempfile.TemporaryDirectory() as td:
        w1, w2 = os.path.join(td, "run_1"), os.path.join(td, "run_2")
        os.makedirs(w1), os.makedirs(w2)
        m1 = build_faster_rcnn(width=16, fc_out_channels=64, task_id=1, task_split=split).to(dev)
        relocate_segment_final_weights(m1)
        o1 = N.SGDNSCL(m1.parameters(), lr=0.002, momentum=0.9, weight_decay=1e-4, svd=True)
        r1 = N.runner.BRNullSpaceRunner(m1, o1, w1, task_id=1, train_task_split=split, ignore_keys=ignore)
        cov, rois = r1.train(step_fn, _batches(dev, 3, (0, 15), 1), cov_forward=cov_fwd, cov_batches=_batches(dev, 15, (0, 15), 2),
                             roi_forward=roi_fwd)
        assert not hasattr(m1, "teacher_model") and len(o1.transforms) == 0
        assert len(cov) == 61 and cov["backbone.layer2.0.conv2.weight"].shape == (32 * 9, 32 * 9)
        feats, cls_t = rois[0], rois[1]
        assert feats.shape == (75, 12544) and sorted(set(cls_t.tolist()) - {20}) == list(range(15))
        assert os.path.exists(os.path.join(w1, "covariance.pth")) and os.path.exists(os.path.join(w1, "rois_etc.pth"))
        # runner:591, 946-990: task 1 leaves the Fisher diagonal of its BN parameters next to them; runner:295-299: and a checkpoint
        ewc1 = torch.load(os.path.join(w1, "ewc_reg_terms_ewc.pth"), weights_only=True)
        bn_names = [n for n, _ in m1.named_parameters() if "bn" in n]
        assert sorted(ewc1["importance"]) == sorted(bn_names) and all(len(v) == 1 for v in ewc1["task_param"].values())
        assert sum(float(v[0].sum()) for v in ewc1["importance"].values()) > 0
        assert os.path.exists(os.path.join(w1, "best_final.pth"))

        m2 = build_faster_rcnn(width=16, fc_out_channels=64, task_id=2, task_split=split, previous_path=w1).to(dev)
        relocate_segment_final_weights(m2)
        m2.load_state_dict(m1.state_dict())
        assert m2.roi_head.replay and m2.roi_head.bbox_featss.shape[1] == 12544
        assert sorted(set(m2.roi_head.tmp_label.tolist())) == list(range(15))
        assert os.path.exists(os.path.join(w2, "mask.pth"))                     # head:451-452 writes to the next dir
        before = {n: p.detach().clone() for n, p in m2.named_parameters()}
        # a large lr on the projected part only: its normalised-projector updates must clear fp32 rounding of the weights
        body = [p for n, p in m2.named_parameters() if n.startswith(("backbone", "neck"))]
        rest = [p for n, p in m2.named_parameters() if not n.startswith(("backbone", "neck"))]
        o2 = N.SGDNSCL([dict(params=body, lr=0.5), dict(params=rest)], lr=0.002, momentum=0.9, weight_decay=0.0, svd=True)
        r2 = N.runner.BRNullSpaceRunner(m2, o2, w2, task_id=2, train_task_split=split, previous_dir=w1, ignore_keys=ignore,
                                        rr_thresh=[0.5, 0.7])
        r2.train(step_fn, _batches(dev, 2, (15, 20), 4), cov_forward=cov_fwd, cov_batches=_batches(dev, 2, (15, 20), 5))
        assert (m2.rpn_thresh, m2.roi_thresh) == (0.5, 0.7)                     # runner:439-441
        assert "replay_loss_cls" in step_fn.last and all(torch.isfinite(v) for v in step_fn.last.values())
        # runner:558-565: the task-2 loss dict carries the EWC term, by the rule of runner:1055-1071 (G6 pins the kernel):
        # 1000 * sum_n sum(F_n (theta_n - theta*_n)^2) over the BN parameters that require grad.  step_fn.last is the last
        # forward of the run (the importance pass, on the final weights).
        assert isinstance(m2.loss, N.runner.ewc.EWCHook) and "ewc_loss" in step_fn.last
        now = dict(m2.named_parameters())
        want = sum(1000.0 * (ewc1["importance"][n][0].to(dev).double() * (now[n].detach().double().unsqueeze(0) - ewc1["task_param"][n][0].to(dev).double()) ** 2).sum()
                   for n in bn_names if now[n].requires_grad)
        assert float(want) > 0 and abs(float(step_fn.last["ewc_loss"]) - float(want)) <= 1e-4 * float(want)
        ewc2 = torch.load(os.path.join(w2, "ewc_reg_terms_ewc.pth"), weights_only=True)
        assert all(len(v) == 2 for v in ewc2["importance"].values())            # task 1's entry + task 2's (runner:985-987)
        assert hasattr(m2, "teacher_model") and m2.teacher_model.roi_head.bbox_head.task_id == 1
        assert not any(p.requires_grad for p in m2.teacher_model.parameters())
        names = set(o2.transforms.keys())
        assert "backbone.layer2.0.conv2.weight" in names and "neck.fpn_convs.0.conv.weight" in names
        assert not any(n.startswith(("rpn_head", "roi_head", "teacher_model")) for n in names)
        assert not any("teacher" in n for g in o2.param_groups for n in g["names"])
        for name in ("backbone.layer2.0.conv2.weight", "neck.fpn_convs.1.conv.weight"):
            dW = (dict(m2.named_parameters())[name].detach() - before[name]).flatten(1)
            assert dW.norm() > 0
            lam, Q = torch.linalg.eigh(cov[name])
            top = Q[:, -3:]
            assert (dW @ top).norm() <= 2e-3 * dW.norm(), name
        # a layer outside the projected set moved freely
        assert (dict(m2.named_parameters())["rpn_head.rpn_conv.weight"] - before["rpn_head.rpn_conv.weight"]).norm() > 0
        cov2 = torch.load(os.path.join(w2, "covariance.pth"), weights_only=True)
        assert len(cov2) == 61


def test_voc_10_10_step_with_16_image_batches(N, dev):
    """configs[2] (VOC 10+10 task 2: split [0,10,20], 16 images per GPU -- voc_10_10_task2_2007.py:40 -- K <= 100 prototypes) on the
    narrow R-50-FPN.  (i) B = 16 through the covariance hooks: the hooks take the BATCH MEAN of the activation before the unfold
    (runner:908-913), so a 16-image batch gives one [L x D] operand -- checked on a 3x3 backbone conv, an FPN 3x3 and a strided 1x1
    against the oracle's covariance of the very activations the hooks saw; (ii) one task-2 training step of that shape: teacher,
    16-image student forward, the K = 100 bank on the fused replay path (4 row blocks), projected step -- finite losses, the fused
    replay loss equal to the module path's."""
    from nsgp_repre_amd.detection import DetSample, Instances, build_faster_rcnn
    torch.manual_seed(5)
    split, B, h, w = [0, 10, 20], 16, 128, 160
    g = torch.Generator().manual_seed(50)

    def batch(classes, seed):
        g.manual_seed(seed)
        x = torch.rand(B, 3, h, w, generator=g).to(dev)
        samples = []
        for i in range(B):
            wh = torch.rand(3, 2, generator=g) * torch.tensor([w * 0.3, h * 0.3]) + 20
            xy = torch.rand(3, 2, generator=g) * (torch.tensor([w, h]) - wh)
            labels = (torch.arange(3) + 3 * i) % (classes[1] - classes[0]) + classes[0]
            samples.append(DetSample(Instances(bboxes=torch.cat([xy, xy + wh], -1).to(dev), labels=labels.to(dev)), img_shape=(h, w)))
        return x, samples
    model = build_faster_rcnn(width=16, fc_out_channels=64, task_id=2, task_split=split).to(dev)
    N.runner.nullspace.guard_conv_weights(model)
    # (i) covariance hooks at B = 16
    watch = {"backbone.layer2.0.conv2": None, "neck.fpn_convs.1.conv": None, "backbone.layer3.0.downsample.0": None}
    mods = dict(model.named_modules())
    hs = [mods[n].register_forward_hook(lambda m, i, o, n=n: watch.__setitem__(n, i[0].detach().float().cpu())) for n in watch]
    ignore = N.runner.nullspace.full_ignore_keys(["rpn", "roi_head"])
    cov = N.runner.cal_fea_in(model, [batch((0, 10), 1)], ignore, forward=lambda m, b: m(b[0], copy.deepcopy(b[1]), mode="nullspace"))
    for h_ in hs:
        h_.remove()
    for n, x in watch.items():
        assert x is not None and x.shape[0] == B
        m = mods[n]
        want = O.cov_conv2d(x, m.kernel_size, m.stride, m.padding)
        got = cov[n + ".weight"].cpu()
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item(), n
    # (ii) one training step of the configs[2] shape
    head = model.roi_head
    K = 100
    head.replay, head.task_split, head.task_id = True, split, 2
    head.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev))
    head.tmp_label = torch.randint(0, 10, (K,), device=dev)
    mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
    mix.task_id = 2
    mix.attach_teacher(model)
    opt = N.SGDNSCL(model.parameters(), lr=0.002, momentum=0.9, weight_decay=1e-4, svd=True)
    N.runner.nullspace.wire_param_names(opt, model)
    N.runner.update_optim_transforms(opt, {k: v for k, v in cov.items()}, ignore)
    model.train()
    x, samples = batch((10, 20), 2)
    losses = model(x, copy.deepcopy(samples), mode="loss")
    assert "replay_loss_cls" in losses and all(bool(torch.isfinite(v)) for v in losses.values())
    assert head._fused_replay_operands() is not None
    head.fused_replay = False
    ref = head.add_replay_loss({})["replay_loss_cls"]
    head.fused_replay = True
    assert abs(losses["replay_loss_cls"].item() - ref.item()) <= 2e-6 * abs(ref.item())
    sum(v for k, v in losses.items() if "loss" in k).backward()
    opt.step()
    torch.cuda.synchronize()
    assert opt.lowrank_stats()[0] + opt.tile_counts()[0] + opt.tile_counts()[1] + opt.tile_counts()[2] > 0
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    opt.close()


def test_segment_final_conv_weight_is_relocated(N, dev):
    """The harness guard against the MIOpen over-read that aborted round 1 (profiles/README.md, incident analysis): rebuild the
    dangerous layout on purpose -- a 512-byte 1x1 conv weight as the last block of a full 2 MiB segment with nothing mapped behind
    it -- WITHOUT running MIOpen on it, let the guard move it, then run forward + backward (backward-data is the kernel that reads
    past the weight) on the relocated weight."""
    from nsgp_repre_amd.detection import relocate_segment_final_weights
    fill, final = [], None
    for _ in range(64 * 4096):
        t = torch.empty(128, device=dev)                      # one 512-byte block
        fill.append(t)
        end = t.data_ptr() + 512
        if end % (1 << 21):
            continue
        segs = [(s_["address"], s_["address"] + s_["total_size"]) for s_ in torch.cuda.memory_snapshot()]
        if any(e == end for _, e in segs) and not any(a == end for a, _ in segs):
            final = t
            break
    assert final is not None
    conv = torch.nn.Conv2d(16, 8, 1).to(dev)
    with torch.no_grad():
        final.view(8, 16, 1, 1).copy_(conv.weight)
    conv.weight.data = final.view(8, 16, 1, 1)                # the layout of the incident
    del fill
    end_before = conv.weight.data_ptr() + 512
    assert relocate_segment_final_weights(conv) >= 1
    assert conv.weight.data_ptr() + 512 != end_before and torch.equal(conv.weight.data.flatten(), final)
    assert relocate_segment_final_weights(conv) == 0           # nothing left to move
    x = torch.randn(2, 16, 8, 8, device=dev, requires_grad=True)
    conv(x).square().mean().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(x.grad).all() and torch.isfinite(conv.weight.grad).all()
