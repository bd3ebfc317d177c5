"""GPU tests of the runner / head layer: covariance hooks, file hand-off between tasks, the
stand-alone BRNullSpaceRunner, and the defining property of NSGP (updates do not move the old
task's features)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

import inputs as I
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


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential()
        self.backbone.add_module("conv1", nn.Conv2d(3, 16, 3, padding=1))
        self.backbone.add_module("bn1", nn.BatchNorm2d(16))
        self.backbone.add_module("relu", nn.ReLU())
        self.backbone.add_module("conv2", nn.Conv2d(16, 16, 3, stride=2, padding=1))
        self.neck = nn.Conv2d(16, 8, 1)
        self.rpn_head = nn.Conv2d(8, 4, 3, padding=1)
        self.fc = nn.Linear(8, 5)

    def forward(self, x):
        f = self.neck(torch.relu(self.backbone(x)))
        return self.rpn_head(f).mean() + self.fc(f.mean(dim=(2, 3))).mean()


def _oracle_covariances(net_cpu, batches, ignore):
    """The reference's route on the CPU: hooks + unfold + mm (restated in the oracle)."""
    fea = {}
    acts = {}
    hs = []
    for n, m in net_cpu.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)) and not any(__import__("re").match(k, n) for k in ignore):
            hs.append(m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, (mod, i[0].detach()))))
    net_cpu.eval()
    with torch.no_grad():
        for x in batches:
            acts.clear()
            net_cpu(x)
            for n, (mod, a) in acts.items():
                c = O.cov_conv2d(a, mod.kernel_size, mod.stride, mod.padding) if isinstance(mod, nn.Conv2d) else O.cov_linear(a)
                O.update_cov(fea, n + ".weight", c)
    for h in hs:
        h.remove()
    return fea


def test_cal_fea_in_matches_oracle_and_accumulates_previous_task(N, dev):
    torch.manual_seed(0)
    net_cpu = Net()
    net = Net().to(dev)
    net.load_state_dict(net_cpu.state_dict())
    batches = [torch.randn(2, 3, 20, 28) for _ in range(3)]
    ignore = N.runner.full_ignore_keys(["rpn", "roi_head"])
    ref = _oracle_covariances(net_cpu, batches, ignore)
    with tempfile.TemporaryDirectory() as td:
        p1 = os.path.join(td, "covariance_t1.pth")
        cov = N.runner.cal_fea_in(net, [b.to(dev) for b in batches], ignore, save_path=p1, task_id=1)
        assert sorted(cov) == sorted(ref) == ["backbone.conv1.weight", "backbone.conv2.weight", "fc.weight", "neck.weight"]
        for k in ref:
            assert _rel(cov[k], ref[k]) <= 1e-5, k
        loaded = torch.load(p1, weights_only=True)          # the on-disk contract: dict[str -> [D x D]]
        assert sorted(loaded) == sorted(ref)
        # task 2: the new covariance is ADDED to the previous file's (runner:751-754)
        cov2 = N.runner.cal_fea_in(net, [b.to(dev) for b in batches], ignore, previous_path=p1, task_id=2)
        for k in ref:
            assert _rel(cov2[k], 2 * ref[k]) <= 1e-5, k
    assert not any(len(m._forward_hooks) for m in net.modules())   # hooks removed


def test_covariance_side_streams_are_bitwise_the_single_stream_result(N, dev):
    """``CovarianceCollector`` runs the accumulations of different layers on side HIP streams (layer i on stream i % n,
    ordered behind the producer of its input and behind earlier accumulations into the same C); ``join()`` / ``remove()``
    order the caller behind them.  Three batches, 1 / 2 / 4 / 8 streams: every covariance bit for bit the same."""
    from nsgp_repre_amd.runner.nullspace import CovarianceCollector
    torch.manual_seed(4)
    net = nn.Sequential(*[nn.Conv2d(c_in, c_out, k, padding=k // 2) for c_in, c_out, k in
                          ((3, 16, 3), (16, 32, 1), (32, 64, 3), (64, 64, 1), (64, 128, 3), (128, 64, 1), (64, 32, 3), (32, 8, 1))]).to(dev).eval()
    batches = [torch.randn(2, 3, 40, 56, device=dev) for _ in range(3)]
    got = {}
    for n_streams in (1, 2, 4, 8):
        col = CovarianceCollector(net, [], n_streams=n_streams).register()
        with torch.no_grad():
            for b in batches:
                net(b)
        col.remove()            # joins the side streams
        torch.cuda.synchronize()
        got[n_streams] = {k: v.clone() for k, v in col.fea_in.items()}
    assert len(got[1]) == 8
    for n_streams in (2, 4, 8):
        for k in got[1]:
            assert torch.equal(got[n_streams][k], got[1][k]), (n_streams, k)


def test_covariance_hooks_under_bf16_activations(N, dev):
    """configs[4] ("fp32 covariance / bf16 activations"): when the hooked forward runs under bf16 autocast the hooks see bf16
    inputs.  The covariance stays fp32: each hook widens its input exactly (bf16 -> fp32 is lossless), so the result must equal
    -- bit for bit -- the fp32 covariance of the very bf16 values the layers consumed, and the files keep dtype float32."""
    torch.manual_seed(2)
    net = Net().to(dev).eval()
    batches = [torch.randn(2, 3, 20, 28, device=dev) for _ in range(2)]
    ignore = N.runner.full_ignore_keys(["rpn", "roi_head"])
    seen = {}
    hs = [m.register_forward_hook(lambda mod, i, o, n=n: seen.setdefault(n, []).append(i[0].detach().clone()))
          for n, m in net.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear)) and not n.startswith("rpn")]

    def fwd(model, batch):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            model(batch)
    cov = N.runner.cal_fea_in(net, batches, ignore, forward=fwd)
    for h in hs:
        h.remove()
    from nsgp_repre_amd import ops
    assert all(v.dtype == torch.float32 for v in cov.values())
    for n, m in net.named_modules():
        if n + ".weight" not in cov:
            continue
        want = None
        for x in seen[n]:
            xf = x.float().contiguous()
            want = (ops.cov_accumulate_conv2d(xf, m.kernel_size, m.stride, m.padding, want) if isinstance(m, nn.Conv2d)
                    else ops.cov_accumulate_linear(xf, want))
        assert torch.equal(cov[n + ".weight"], want), n
    assert any(x.dtype == torch.bfloat16 for xs in seen.values() for x in xs)      # the hooks did see bf16 activations


def test_prototype_replay_head_from_files(N, dev, golden_dir):
    """rois_etc.pth -> bank + mask.pth in the next work dir; then mask.pth replayed gives the same bank."""
    g = np.load(os.path.join(golden_dir, "g4_prototypes.npz"))
    feats, cls = I.g4_rois()
    n = feats.shape[0]
    rois = [torch.from_numpy(feats), torch.from_numpy(cls), torch.ones(n), torch.zeros(n, 4), torch.zeros(n, 4), torch.zeros(n, 5)]
    with tempfile.TemporaryDirectory() as td:
        prev, cur, nxt = (os.path.join(td, f"x_15_5_{i}") for i in (1, 2, 3))
        for d in (prev, cur, nxt):
            os.makedirs(d)
        torch.save(rois, os.path.join(prev, "rois_etc.pth"))
        bbox_head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=64, roi_feat_size=7, num_classes=5,
                                                      task_split=I.G4_TASK_SPLIT, task_id=2).to(dev)
        head = N.roi_heads.StandardMultiPrototypeReplayHead(bbox_head=bbox_head, previous_path=prev, task_id=2,
                                                            task_split=I.G4_TASK_SPLIT, max_prototype=I.G4_MAX_PROTO)
        assert head.replay
        np.testing.assert_array_equal(head.tmp_label.cpu().numpy(), g["labels"])
        assert _rel(head.bbox_featss, torch.from_numpy(g["bank"])) <= 1e-5
        masks = torch.load(os.path.join(cur, "mask.pth"), weights_only=True)
        for c in range(3):
            assert len(masks[c]) == int(g[f"nmask_{c}"])
            for j, m in enumerate(masks[c]):
                np.testing.assert_array_equal(m.numpy(), g[f"mask_{c}_{j}"])
        losses = head.add_replay_loss({})
        assert torch.isfinite(losses["replay_loss_cls"])
        losses["replay_loss_cls"].backward()
        assert bbox_head.shared_fcs[0].weight.grad.abs().sum() > 0
        # a later run finds mask.pth next to rois_etc.pth and replays it (head:407-408, 425-433)
        torch.save(rois, os.path.join(cur, "rois_etc.pth"))
        head2 = N.roi_heads.StandardMultiPrototypeReplayHead(bbox_head=bbox_head, previous_path=cur, task_id=2,
                                                             task_split=I.G4_TASK_SPLIT, max_prototype=I.G4_MAX_PROTO)
        assert _rel(head2.bbox_featss, torch.from_numpy(g["bank"])) <= 1e-5
    no_replay = N.roi_heads.StandardMultiPrototypeReplayHead(bbox_head=bbox_head, previous_path=None)
    assert not no_replay.replay and no_replay.add_replay_loss({}) == {}


def test_two_task_run_keeps_old_features_fixed(N, dev):
    """End-to-end on a tiny conv net: task 1 (plain SGD) -> covariance.pth -> task 2 with NSGP.
    Property: for a projected conv, the weight change dW applied to any OLD-task input patch x
    (a row of X) stays tiny relative to the un-projected change: ||X dW^T|| << ||X dW_plain^T||."""
    torch.manual_seed(1)
    data1 = [torch.randn(2, 3, 16, 16, device=dev) * torch.tensor([1.0, 0.1, 0.01], device=dev).view(1, 3, 1, 1) for _ in range(4)]
    data2 = [torch.randn(2, 3, 16, 16, device=dev) for _ in range(4)]
    with tempfile.TemporaryDirectory() as td:
        w1, w2 = os.path.join(td, "run_1"), os.path.join(td, "run_2")
        os.makedirs(w1); os.makedirs(w2)
        net = Net().to(dev)
        opt = N.SGDNSCL(net.parameters(), lr=0.05, momentum=0.9, svd=True)
        r1 = N.runner.BRNullSpaceRunner(net, opt, w1, task_id=1, ignore_keys=["rpn", "roi_head"])
        seen = {"n": 0}

        def step1(m, b):        # a "best" checkpoint is written after the second step, two more steps follow
            if seen["n"] == 2:
                r1.save_checkpoint(m, "best_loss_iter_2.pth")
                seen["best"] = {k: v.detach().clone() for k, v in m.state_dict().items()}
            seen["n"] += 1
            return m(b)
        # Round 1's "Fatal Python error: Aborted" lived here: MIOpen's backward-data kernel for the 1x1 `neck` convolution reads past the
        # end of its 512-byte weight tensor (profiles/README.md, incident analysis; tools/miopen_overread_repro.py reproduces it with
        # PyTorch + MIOpen alone) -- a GPU memory access fault whenever the caching allocator places neck.weight as the last block of a
        # segment.  Round 2 ran these passes with MIOpen off; now BOTH runner faces call `guard_conv_weights` (after the model is on the
        # device, after checkpoint loads, after the teacher copy), so the passes run on MIOpen like a real run does.
        cov, _ = r1.train(step1, data1, importance_loss=lambda m, b: m(b))
        assert not any((p.data_ptr() + p.numel() * 4) in {s_["address"] + s_["total_size"] for s_ in torch.cuda.memory_snapshot()}
                       for p in [m.weight for m in net.modules() if isinstance(m, nn.Conv2d)])
        assert os.path.exists(os.path.join(w1, "covariance.pth")) and len(opt.transforms) == 0
        # runner:710-716: the end-of-task passes run on the RELOADED ckpt_keywords checkpoint, not on the last iteration's weights:
        # the model is back at the saved state, and covariance.pth is the covariance of THAT model (conv2's input depends on conv1 + bn1)
        for k, v in net.state_dict().items():
            assert torch.equal(v, seen["best"][k]), k
        ref = Net()
        ref.load_state_dict({k: v.cpu() for k, v in seen["best"].items()})
        want = _oracle_covariances(ref, [d.cpu() for d in data1], N.runner.nullspace.full_ignore_keys(["rpn", "roi_head"]))
        assert _rel(cov["backbone.conv2.weight"], want["backbone.conv2.weight"]) <= 1e-4
        assert os.path.exists(os.path.join(w1, "ewc_reg_terms_ewc.pth"))                          # runner:591
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        opt2 = N.SGDNSCL(net.parameters(), lr=0.05, momentum=0.9, svd=True)
        r2 = N.runner.BRNullSpaceRunner(net, opt2, w2, task_id=2, previous_dir=w1, ignore_keys=["rpn", "roi_head"])
        r2.train(lambda m, b: m(b), data2)
        assert sorted(opt2.transforms.keys()) == ["backbone.conv1.weight", "backbone.conv2.weight", "fc.weight", "neck.weight"]
        assert "rpn_head.weight" not in opt2.transforms
        name = "backbone.conv1.weight"
        P = opt2.transforms[name]
        dW = (dict(net.named_parameters())[name].detach() - before[name]).view(16, -1)
        # dW lies in the row space of P (dW = U P): re-projecting changes nothing
        Pn = P * torch.trace(P)     # undo the backbone 1/||P0||_F scale: trace(P0/c) = rank/c = c for a projector
        assert _rel(dW @ Pn, dW) <= 1e-3
        # and it annihilates the dominant old-task input directions
        lam, Q = torch.linalg.eigh(cov[name])
        top = Q[:, -3:]                                      # top-3 eigen-directions of the old covariance
        assert (dW @ top).norm() <= 1e-3 * dW.norm()
        assert dW.norm() > 0
        # task 2's own covariance file = task 1's + the new pass (checked in the cal_fea_in test)
        assert os.path.exists(os.path.join(w2, "covariance.pth"))


def test_ewc_regulariser_vs_reference_golden(N, dev, golden_dir):
    """fused multi-tensor EWC loss + gradient vs the reference's EWCHook output (G6)."""
    g = np.load(os.path.join(golden_dir, "g6_ewc.npz"))
    tensors = I.g6_tensors()

    class M(nn.Module):
        pass
    model = M()
    named = {}
    for n, (theta, _, _) in tensors.items():
        named[n] = nn.Parameter(torch.from_numpy(theta.copy()).to(dev))
    model.named_parameters = lambda *a, **k: iter(named.items())
    reg_params = N.runner.ewc.register_params(model)
    assert sorted(reg_params) == sorted(str(g[f"name_{k}"]) for k in range(int(g["n_reg"])))
    terms = {"importance": {n: [torch.from_numpy(a) for a in tensors[n][1]] for n in reg_params},
             "task_param": {n: [torch.from_numpy(a) for a in tensors[n][2]] for n in reg_params}}
    holder = type("H", (), {})()
    holder.loss = lambda *a, **k: {"loss_cls": torch.ones((), device=dev)}
    hook = N.runner.ewc.EWCHook(holder, reg_params, terms)
    res = hook()
    assert set(res) == {"loss_cls", "ewc_loss"}
    np.testing.assert_allclose(res["ewc_loss"].item(), g["ewc_loss"], rtol=1e-5)
    (res["ewc_loss"] * 0.5 + res["loss_cls"]).backward()          # grad_out = 0.5 reaches the kernel
    for k in range(int(g["n_reg"])):
        n = str(g[f"name_{k}"])
        assert _rel(reg_params[n].grad, 0.5 * torch.from_numpy(g[f"grad_{k}"])) <= 1e-5, n
    # frozen parameters drop out, exactly like `if not p.requires_grad: continue`
    for p in reg_params.values():
        p.requires_grad_(False)
    assert N.runner.ewc.EWCRegulariser(reg_params, terms).n == 0


def test_ewc_importance_accumulation_and_file(N, dev):
    """calculate_save_importance (runner:946-990) restated: F += grad^2 * len(batch)/len(loader); file layout."""
    p = {"backbone.bn1.weight": nn.Parameter(torch.randn(9, device=dev)), "backbone.bn1.bias": nn.Parameter(torch.randn(9, device=dev))}
    imp = {n: torch.zeros_like(v) for n, v in p.items()}
    ref = {n: torch.zeros(9) for n in p}
    for b in range(3):
        for n, v in p.items():
            v.grad = torch.randn(9, device=dev)
            ref[n] += v.grad.cpu() ** 2 * 2 / 3
        N.runner.ewc.accumulate_importance(imp, p, batch_len=2, loader_len=3)
    for n in p:
        assert _rel(imp[n], ref[n]) <= 1e-6
    with tempfile.TemporaryDirectory() as td:
        terms = N.runner.ewc.save_importance(td, {}, imp, p)
        loaded = N.runner.ewc.load_importance(os.path.join(td, "ewc_reg_terms_ewc.pth"), dev)
        assert set(loaded) == {"importance", "task_param"} and loaded["importance"]["backbone.bn1.bias"][0].shape == (1, 9)
        terms = N.runner.ewc.save_importance(td, loaded, imp, p)
        assert len(terms["task_param"]["backbone.bn1.weight"]) == 2


def test_pseudo_label_filter_vs_oracle(N, dev):
    """SURVEY 8f-2 (parity unpinned: torchvision/mmengine absent, no golden): the fused sequential
    filter vs the oracle's restatement of det:78-108 -- bit-exact masks, including the case where a box
    is rejected only because of an EARLIER accepted pseudo box, empty ground truth and empty predictions."""
    rng = np.random.default_rng(0)
    for P, G in ((100, 6), (37, 0), (0, 4), (300, 20), (5, 1)):
        ctr = rng.uniform(50, 750, size=(max(P, 1), 2))
        wh = rng.uniform(20, 200, size=(max(P, 1), 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)[:P]
        if P > 10:                       # near-duplicates of earlier boxes: rejected through the grown RoI set
            boxes[5] = boxes[2] + 1.0
            boxes[9] = boxes[2] - 0.5
        scores = np.sort(rng.uniform(0.05, 1.0, size=P).astype(np.float32))[::-1].copy()
        gctr = rng.uniform(100, 700, size=(G, 2)); gwh = rng.uniform(30, 250, size=(G, 2))
        gt = np.concatenate([gctr - gwh / 2, gctr + gwh / 2], axis=1).astype(np.float32)
        if G > 0 and P > 3:
            boxes[1] = gt[0] + 0.25      # overlaps a real ground-truth box: dropped
        b, s, g_ = torch.from_numpy(boxes).reshape(-1, 4), torch.from_numpy(scores), torch.from_numpy(gt).reshape(-1, 4)
        ref_rpn, ref_roi = O.pseudo_label_filter(b, s, g_, 0.5, 0.7)
        rpn, roi = N.detectors.filter_pseudo_labels(b.to(dev), s.to(dev), g_.to(dev), 0.5, 0.7)
        assert torch.equal(rpn.cpu(), ref_rpn) and torch.equal(roi.cpu(), ref_roi), (P, G)
        if P > 10:
            if ref_roi[2]:
                assert not ref_roi[5] and not ref_roi[9]    # the sequential dependency is exercised
            assert bool((ref_roi <= ref_rpn).all())          # roi_thresh >= rpn_thresh


def test_replay_loss_through_the_head_vs_reference_golden(N, dev, golden_dir):
    """G5 end to end on the GPU: bank -> Shared2FCBBoxHeadTask -> kept columns -> fused CE(softmax(.)) -> backward into
    the shared FCs, against the reference's own loss and weight gradients."""
    from test_runner_host import _g5_head
    g = np.load(os.path.join(golden_dir, "g5_replay_loss.npz"))
    head = _g5_head(N, dev)
    bank, labels = I.g5_bank()

    class Holder(N.roi_heads.PrototypeReplay):
        pass
    h = Holder()
    h.bbox_head, h.task_split, h.task_id = head, I.G5_TASK_SPLIT, I.G5_TASK_ID
    h.tmp_label, h.replay = torch.from_numpy(labels).to(dev), True
    h.bbox_featss = torch.from_numpy(bank).reshape(-1, 4, 7, 7).to(dev)
    res = h.replay_loss(h.bbox_featss)
    loss = res["replay_loss"]["replay_loss_cls"]
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-6)
    loss.backward()
    for i, m in enumerate(head.shared_fcs):
        np.testing.assert_allclose(m.weight.grad.cpu().numpy(), g[f"gW_shared{i}"], rtol=1e-4, atol=1e-7)
    losses = h.add_replay_loss(dict(loss_cls=torch.tensor(1.0, device=dev)))
    assert set(losses) == {"loss_cls", "replay_loss_cls"}
    with torch.autocast("cuda", dtype=torch.bfloat16):          # the bank pass stays fp32 inside an autocast step
        inside = h.add_replay_loss({})["replay_loss_cls"]
    assert inside.dtype == torch.float32 and abs(inside.item() - g["loss"]) <= 2e-6 * abs(float(g["loss"]))


def _replay_holder(N, head, labels, bank, split, tid):
    class Holder(N.roi_heads.PrototypeReplay):
        pass
    h = Holder()
    h.bbox_head, h.task_split, h.task_id = head, list(split), tid
    h.tmp_label, h.replay, h.bbox_featss = labels, True, bank
    return h


def test_fused_replay_pass_vs_reference_golden(N, dev, golden_dir):
    """G5 on the FUSED per-step path (csrc/replay_head.hip: skinny split-K MFMA GEMMs + slab reduce + class scores + double-softmax CE,
    and the hand-written backward): the reference's own loss, its gradients of both shared FCs and of every per-task class head,
    on G5's ragged shape (K = 14, 196 -> 32 -> 32 -> 6 kept columns: the guarded tiles).  The frozen future head and the
    regression heads receive no gradient, as in the reference."""
    from test_runner_host import _g5_head
    g = np.load(os.path.join(golden_dir, "g5_replay_loss.npz"))
    head = _g5_head(N, dev)
    bank, labels = I.g5_bank()
    h = _replay_holder(N, head, torch.from_numpy(labels).to(dev), torch.from_numpy(bank).reshape(-1, 4, 7, 7).to(dev), I.G5_TASK_SPLIT, I.G5_TASK_ID)
    assert h._fused_replay_operands() is not None
    losses = h.add_replay_loss({})
    assert set(losses) == {"replay_loss_cls"}
    np.testing.assert_allclose(losses["replay_loss_cls"].item(), g["loss"], rtol=2e-6)
    losses["replay_loss_cls"].backward()
    for i, m in enumerate(head.shared_fcs):
        np.testing.assert_allclose(m.weight.grad.cpu().numpy(), g[f"gW_shared{i}"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(m.bias.grad.cpu().numpy(), g[f"gb_shared{i}"], rtol=1e-4, atol=1e-7)
    for i, m in enumerate(head.fc_cls):
        if bool(g[f"has_grad_cls{i}"]):
            np.testing.assert_allclose(m.weight.grad.cpu().numpy(), g[f"gW_cls{i}"], rtol=1e-4, atol=1e-7)
        else:
            assert m.weight.grad is None
    assert all(m.weight.grad is None for m in head.fc_reg)
    # and the module-by-module path gives the same numbers
    head.zero_grad(set_to_none=True)
    h.fused_replay = False
    ref = h.add_replay_loss({})["replay_loss_cls"]
    assert abs(ref.item() - losses["replay_loss_cls"].item()) <= 2e-6 * abs(ref.item())


@pytest.mark.parametrize("K,fin,hidden,split,tid", [(150, 12544, 1024, (0, 15, 20), 2), (100, 12544, 1024, (0, 10, 20), 2),
                                                    (400, 12544, 1024, (0, 40, 80), 2), (37, 200, 96, (0, 3, 5, 7), 2),
                                                    (161, 256, 128, (0, 5, 10), 1)])
def test_fused_replay_pass_at_config_sizes_vs_fp64(N, dev, K, fin, hidden, split, tid):
    """The fused replay pass at the sizes of configs[1] (K = 150, 21 kept columns), configs[2] (K = 100), configs[3] (K = 400, 81
    columns), a ragged shape on the guarded tiles and a 6-block bank (two row chunks) against an fp64 autograd evaluation of the
    same head on the GPU: loss 1e-6, every gradient within 1e-5 of its tensor's maximum; bitwise reproducible run to run; equal to
    the module-by-module fp32 path within 1e-5."""
    import copy
    torch.manual_seed(K + hidden)
    head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=fin, fc_out_channels=hidden, roi_feat_size=1, num_classes=split[-1],
                                             task_split=list(split), task_id=tid).to(dev)
    with torch.no_grad():
        for p in head.parameters():
            p.mul_(3.0)                     # spread the scores: the CE of a softmax is flat for small logits
    n_old = split[tid - 1] if tid > 1 else split[tid]
    bank = torch.relu(torch.randn(K, fin, device=dev))
    labels = torch.randint(0, max(1, n_old), (K,), device=dev)
    h = _replay_holder(N, head, labels, bank, split, tid)
    assert h._fused_replay_operands() is not None

    def run(fused):
        head.zero_grad(set_to_none=True)
        h.fused_replay = fused
        loss = h.add_replay_loss({})["replay_loss_cls"]
        (loss * 1.5).backward()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in head.named_parameters() if p.grad is not None}
    loss_f, g_f = run(True)
    loss_f2, g_f2 = run(True)
    assert torch.equal(loss_f, loss_f2) and all(torch.equal(g_f[n], g_f2[n]) for n in g_f), "the fused pass is not reproducible"
    loss_m, g_m = run(False)
    # fp64 reference
    h64 = copy.deepcopy(head).double()
    h64.zero_grad(set_to_none=True)
    cls64, _ = h64(bank.double())
    pre = split[tid]
    kept = torch.cat([cls64[:, :pre], cls64[:, -1:]], -1)
    loss64 = torch.nn.functional.cross_entropy(kept.softmax(-1), labels)
    (loss64 * 1.5).backward()
    g64 = {n: p.grad for n, p in h64.named_parameters() if p.grad is not None}
    assert set(g_f) == set(g64) == set(g_m), (sorted(g_f), sorted(g64))
    assert abs(loss_f.item() - loss64.item()) <= 1e-6 * abs(loss64.item()) + 1e-7
    for n in g64:
        assert _rel(g_f[n], g64[n]) <= 1e-5, (n, _rel(g_f[n], g64[n]))
        assert _rel(g_f[n], g_m[n]) <= 1e-5, (n, "fused vs module path")


def test_fused_double_softmax_ce_vs_torch_and_golden(N, dev, golden_dir):
    """head:499 ``F.cross_entropy(cls_score.softmax(-1), labels)``: fused kernels vs torch's own ops
    (fp64 reference) on ragged sizes, and vs the reference's loss value of G5."""
    from nsgp_repre_amd import ops
    g = torch.Generator().manual_seed(4)
    for K, Cn in ((150, 21), (1, 2), (400, 81), (37, 130), (5, 256)):
        s = torch.randn(K, Cn, generator=g) * 3
        y = torch.randint(0, Cn, (K,), generator=g)
        s64 = s.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(s64.softmax(-1), y)
        ref.backward()
        sd = s.to(dev).requires_grad_(True)
        out = ops.double_softmax_cross_entropy(sd, y.to(dev))
        (out * 2.0).backward()
        assert abs(out.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-7
        assert _rel(sd.grad, 2.0 * s64.grad) <= 1e-5
    gold = np.load(os.path.join(golden_dir, "g5_replay_loss.npz"))
    sc = torch.from_numpy(gold["cls_score"])
    kept = torch.cat([sc[:, :I.G5_TASK_SPLIT[I.G5_TASK_ID]], sc[:, -1:]], dim=-1).contiguous()
    _, labels = I.g5_bank()
    out = ops.double_softmax_cross_entropy(kept.to(dev), torch.from_numpy(labels).to(dev))
    np.testing.assert_allclose(out.item(), gold["loss"], rtol=1e-6)


def test_roi_dump_vs_reference_golden_on_device(N, dev, golden_dir):
    """a13 with the boxes on the GPU: same assignment, same seeded draws (the permutations come from the CPU generator as in
    mmdet's RandomSampler), same five rows as the reference's ``get_bbox_stuff`` (G8)."""
    from roi_dump_check import check_roi_dump
    check_roi_dump(N, golden_dir, dev)


def test_pseudo_labelled_sets_vs_reference_golden(N, dev, golden_dir):
    """8f-2 through the detector: ``FasterRCNNRoIReplay.loss`` (fused sequential filter kernel) hands its RPN head and its RoI
    head exactly the ground-truth sets the reference's own ``loss`` handed its heads on the same teacher output (G9), for three
    threshold pairs.  The IoU arithmetic of the absent torchvision is not pinned by this (the reference was given its own tree's
    bbox_overlaps); the loop -- growth of the RoI set, both thresholds, label zeroing for the RPN -- is."""
    from types import SimpleNamespace
    from nsgp_repre_amd.detection.structures import DetSample, Instances
    from nsgp_repre_amd.detectors.faster_rcnn_roi_replay import FasterRCNNRoIReplay
    G = np.load(f"{golden_dir}/g9_pseudo_labels.npz")
    for ti, (rpn_t, roi_t) in enumerate(I.G9_THRESHOLDS):
        imgs, seen = I.g9_batch(), {}

        class Teacher(torch.nn.Module):
            def predict(self, inputs, samples, rescale=False):
                for s, im in zip(samples, imgs):
                    s.pred_instances = Instances(bboxes=torch.from_numpy(im["pred_bboxes"]).to(dev), scores=torch.from_numpy(im["pred_scores"]).to(dev),
                                                 labels=torch.from_numpy(im["pred_labels"]).to(dev))
                return samples

        class Rpn(torch.nn.Module):
            def loss_and_predict(self, x, samples, proposal_cfg=None):
                seen["rpn"] = [(s.gt_instances.bboxes.cpu().numpy(), s.gt_instances.labels.cpu().numpy()) for s in samples]
                return {}, [None] * len(samples)

        class Roi(torch.nn.Module):
            def loss(self, x, rpn_results, samples):
                seen["roi"] = [(s.gt_instances.bboxes.cpu().numpy(), s.gt_instances.labels.cpu().numpy()) for s in samples]
                return {}
        det = FasterRCNNRoIReplay(backbone=torch.nn.Identity(), neck=torch.nn.Identity(), rpn_head=Rpn(), roi_head=Roi())
        det.teacher_model = Teacher()
        det.rpn_thresh, det.roi_thresh = rpn_t, roi_t
        samples = [DetSample(Instances(bboxes=torch.from_numpy(im["gt_bboxes"]).to(dev), labels=torch.from_numpy(im["gt_labels"]).to(dev)),
                             img_shape=I.G8_CANVAS) for im in imgs]
        det.loss(torch.zeros(len(imgs), 3, 8, 8, device=dev), samples)
        for which in ("rpn", "roi"):
            for i, (b, l) in enumerate(seen[which]):
                assert np.array_equal(b, G[f"t{ti}__{which}__img{i}__bboxes"]), (ti, which, i)
                assert np.array_equal(l, G[f"t{ti}__{which}__img{i}__labels"]), (ti, which, i)
