"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz,
made by tests/golden/make_golden.py which executed the reference's own files).
CPU only.  Tolerances: integer/bool results bit-exact; fp32 results that go
through the SAME torch-CPU calls bit-exact or 1e-6; projector via our own SVD
call 1e-5 relative to max|P|."""
import os

import numpy as np
import pytest
import torch

import inputs as I
import nsgp_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _key(n):
    return n.replace(".", "_")


STEPPERS = dict(sgd=O.sgd_nscl_step, sgd_nesterov=O.sgd_nscl_step, sgdna=O.sgd_nscl_step,
                adamw=O.adamw_nscl_step, adamw_amsgrad=O.adamw_nscl_step, adam=O.adam_nscl_step)


def _oracle_transforms(kind, names, hp):
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1_covariances().items()}
    tr = {}
    for n in names:
        if n not in fea_in:
            continue
        s, V = O.eigens(fea_in[n])
        if kind == "sgdna":
            mask = O.na_threshold(s, hp["thres"])
        else:
            mask = O.adaptive_threshold(s, I.G1_OFFSET, "sgd" if kind.startswith("sgd") else "adam")
        tr[n] = (s, mask, O.build_projector(V, mask, kind == "adam" or "backbone" in n))
    return tr


@pytest.mark.parametrize("kind", list(STEPPERS))
def test_g1_optimizer_steps(golden_dir, kind):
    g = _load(golden_dir, f"g1_{kind}.npz")
    names, _ = I.g1_layers()
    hp = dict(I.G1_HYPER[kind])
    tr = _oracle_transforms(kind, names, hp)
    assert sorted(tr) == sorted(I.g1_projected())
    for n, (s, mask, P) in tr.items():
        np.testing.assert_array_equal(s.numpy(), g[f"sigma__{_key(n)}"])
        if f"P__{_key(n)}" in g.files:
            Pg = g[f"P__{_key(n)}"]
            assert np.abs(P.numpy() - Pg).max() <= 1e-6 * np.abs(Pg).max()
        else:
            assert abs(float(P.norm()) - float(g[f"Pnorm__{_key(n)}"])) <= 1e-5 * float(g[f"Pnorm__{_key(n)}"])
    transforms = {n: t[2] for n, t in tr.items()}
    hp.pop("thres", None)
    params = [torch.from_numpy(a) for a in I.g1_params()]
    states = [dict() for _ in params]
    for step in range(I.G1_STEPS):
        grads = [torch.from_numpy(a) for a in I.g1_grads(step)]
        STEPPERS[kind](names, params, grads, states, transforms, **hp)
        for n, p, gr in zip(names, params, grads):
            ref = g[f"p_step{step}__{_key(n)}"]
            assert np.abs(p.numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30), (kind, n, step)
            refg = g[f"g_step{step}__{_key(n)}"]
            np.testing.assert_allclose(gr.numpy(), refg, rtol=1e-6, atol=1e-7)
    for n, st in zip(names, states):
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            if sk in st:
                np.testing.assert_allclose(st[sk].numpy(), g[f"{sk}__{_key(n)}"], rtol=1e-6, atol=1e-12)


def _row_rel(a, ref):
    """max over rows of max_j|a - ref| / max_j|ref|  (rows = dim 0; all-zero reference rows must match exactly)."""
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    ref = np.asarray(ref, np.float64).reshape(ref.shape[0], -1)
    err, mag = np.abs(a - ref).max(1), np.abs(ref).max(1)
    assert (err[mag == 0] == 0).all()
    return float((err[mag > 0] / mag[mag > 0]).max()) if (mag > 0).any() else 0.0


@pytest.mark.parametrize("kind", list(I.G1B_KINDS))
def test_g1b_aligned_layers_per_row(golden_dir, kind):
    """G1b: 128-aligned layers, parameters starting at zero, gradient rows over six decades.  The oracle's first step must
    reproduce the reference's projected update ROW BY ROW (same torch-CPU mm: 1e-6 of each row's own maximum)."""
    g = _load(golden_dir, f"g1b_{kind}.npz")
    gP = _load(golden_dir, "g1b_sgd.npz")
    names, _ = I.g1b_layers()
    hp = dict(I.G1_HYPER[kind])
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1b_covariances().items()}
    transforms = {}
    for n in I.g1b_projected():
        s, V = O.eigens(fea_in[n])
        np.testing.assert_array_equal(s.numpy(), g[f"sigma__{_key(n)}"])
        mask = O.adaptive_threshold(s, I.G1_OFFSET, "sgd" if kind == "sgd" else "adam")
        P = O.build_projector(V, mask, "backbone" in n)
        Pg = gP[f"P__{_key(n)}"]
        if kind == "sgd":
            assert np.abs(P.numpy() - Pg).max() <= 1e-6 * np.abs(Pg).max()
        transforms[n] = torch.from_numpy(Pg) if kind == "sgd" else P
    params = [torch.from_numpy(a) for a in I.g1b_params()]
    states = [dict() for _ in params]
    for step in range(I.G1B_STEPS):
        grads = [torch.from_numpy(a) for a in I.g1b_grads(step)]
        STEPPERS[kind](names, params, grads, states, transforms, **hp)
        for n, p, gr in zip(names, params, grads):
            ref = g[f"p_step{step}__{_key(n)}"]
            if step == 0 and n in transforms and kind == "sgd":
                assert _row_rel(p.numpy(), ref) <= 1e-6, (kind, n)
            assert np.abs(p.numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30), (kind, n, step)
            if step == 0:
                np.testing.assert_allclose(gr.numpy(), g[f"g_step0__{_key(n)}"], rtol=1e-6, atol=1e-9)
    for n, st in zip(names, states):
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq"):
            if sk in st:
                np.testing.assert_allclose(st[sk].numpy(), g[f"{sk}__{_key(n)}"], rtol=1e-6, atol=1e-14)


@pytest.mark.parametrize("kind", list(I.G1C_KINDS))
def test_g1c_low_rank_form_on_the_references_basis_per_row(golden_dir, kind):
    """G1c: the low-rank form c (u - (u U) U^T) with the REFERENCE's own removed directions U = eigen_vector[:, :r] (its torch.svd)
    reproduces the reference's step() -- which multiplies by the dense V_tail V_tail^T -- within the 1e-5 gate on every output row,
    in every rank class (r = 21-24, 48, 99), with nothing added to the tolerance.  The spectrum and the rank come out of the
    oracle's own decomposition as well (rank: exact).  (The oracle divides backbone projectors by torch's fp32 CPU ``torch.norm`` like the
    reference does -- a reduction that is 6e-4 low on the 2304-wide layer, fixture keys refnorm / refnorm64; the GPU test takes that
    scalar out, see there.)"""
    g = _load(golden_dir, f"g1c_{kind}.npz")
    gU = _load(golden_dir, "g1c_sgd.npz")
    names, _ = I.g1c_layers()
    hp = dict(I.G1_HYPER[kind])
    fea_in = {n: torch.from_numpy(c) for n, c in I.g1c_covariances().items()}
    transforms, ranks = {}, {}
    for n in I.g1c_projected():
        r = int(g[f"rank__{_key(n)}"])
        s, _V = O.eigens(fea_in[n])
        np.testing.assert_array_equal(s.numpy(), gU[f"sigma__{_key(n)}"])
        mask = O.adaptive_threshold(s, I.G1_OFFSET, "sgd" if kind.startswith("sgd") else "adam")
        assert int(mask.to(torch.int8).argmax()) == r
        U = torch.from_numpy(gU[f"U__{_key(n)}"])
        assert U.shape == (fea_in[n].shape[0], r)
        transforms[n] = O.HeadBasis(U, "backbone" in n)
        ranks[n] = r
    assert sorted(ranks.values()) == [21, 23, 24, 48, 99]        # rank classes 32 / 64 / 128 of the HIP step
    params = [torch.from_numpy(a) for a in I.g1c_params()]
    states = [dict() for _ in params]
    for step in range(I.G1C_STEPS[kind]):
        grads = [torch.from_numpy(a) for a in I.g1c_grads(step)]
        STEPPERS[kind](names, params, grads, states, transforms, **hp)
        for n, p in zip(names, params):
            if n not in transforms:
                continue
            ref = g[f"p_step{step}__{_key(n)}"]
            if step == 0:       # p0 = 0: p IS the projected update
                assert _row_rel(p.numpy(), ref) <= 1e-5, (kind, n, _row_rel(p.numpy(), ref))
            else:               # p != 0: the fp32 add of the update into p rounds at ulp(p) on both sides
                assert np.abs(p.numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), (kind, n, step)


def test_g2_elbow_indices_bit_exact(golden_dir):
    g = _load(golden_dir, "g2_thresholds.npz")
    for si, s in enumerate(I.g2_spectra()):
        for oi, off in enumerate(I.G2_OFFSETS):
            assert O.elbow_index(s, off, "sgd") == int(g[f"sgd_{si}_{oi}"]), (si, off)
            assert O.elbow_index(s, off, "adam") == int(g[f"adam_{si}_{oi}"]), (si, off)
            assert O.elbow_index(s, off, "sgd") == int(g[f"head_{si}_{oi}"]), (si, off)


def test_gaussian_filter_matches_scipy():
    import scipy.ndimage
    rng = np.random.default_rng(0)
    for n in (128, 147, 1000):
        x = np.sort(rng.random(n).astype(np.float32))[::-1].copy()
        a = O.gaussian_filter1d_reflect(x, 10)
        b = scipy.ndimage.gaussian_filter1d(x, sigma=10)
        assert a.dtype == b.dtype == np.float32
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-7 * np.abs(b).max())


def test_g3_covariance(golden_dir):
    g = _load(golden_dir, "g3_covariance.npz")
    for ci, cfg in enumerate(I.g3_cases()):
        fea = {}
        for x in I.g3_inputs(ci):
            xt = torch.from_numpy(x)
            c = O.cov_conv2d(xt, cfg["k"], cfg["s"], cfg["p"]) if cfg["kind"] == "conv" else O.cov_linear(xt)
            O.update_cov(fea, "layer.weight", c)
        ref = g[f"C_{ci}"]
        assert fea["layer.weight"].shape == ref.shape
        assert np.abs(fea["layer.weight"].numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), ci


def test_g4_prototype_bank(golden_dir):
    g = _load(golden_dir, "g4_prototypes.npz")
    feats, cls = I.g4_rois()
    feats, cls = torch.from_numpy(feats), torch.from_numpy(cls)
    for mode in ("native", "replay_order"):
        orders = None
        if mode == "replay_order":
            orders = {c: torch.from_numpy(g[f"order_{c}"]) for c in range(3)}
        bank, labels, masks, _ = O.build_bank(feats, cls, I.G4_TASK_SPLIT, 2, I.G4_MAX_PROTO, orders=orders)
        np.testing.assert_array_equal(labels.numpy(), g["labels"])
        for c in range(3):
            assert len(masks[c]) == int(g[f"nmask_{c}"])
            for j, m in enumerate(masks[c]):
                np.testing.assert_array_equal(m.numpy(), g[f"mask_{c}_{j}"])
        assert np.abs(bank.numpy() - g["bank"]).max() <= 1e-6 * np.abs(g["bank"]).max()


def test_g4_saved_masks_replay(golden_dir):
    """mask.pth replay branch (head:425-433): feeding the saved masks back gives the same bank."""
    g = _load(golden_dir, "g4_prototypes.npz")
    feats, cls = I.g4_rois()
    feats, cls = torch.from_numpy(feats), torch.from_numpy(cls)
    saved = [[torch.from_numpy(g[f"mask_{c}_{j}"]) for j in range(int(g[f"nmask_{c}"]))] for c in range(3)]
    bank, labels, masks, centres = O.build_bank(feats, cls, I.G4_TASK_SPLIT, 2, I.G4_MAX_PROTO, saved=saved)
    assert all(ci == -1 for cl in centres for ci in cl[:3])
    assert np.abs(bank.numpy() - g["bank"]).max() <= 1e-6 * np.abs(g["bank"]).max()


def test_g5_replay_loss(golden_dir):
    g = _load(golden_dir, "g5_replay_loss.npz")
    w = I.g5_weights()
    bank, labels = I.g5_bank()

    def tt(pair):
        return tuple(torch.from_numpy(a).requires_grad_(True) for a in pair)
    shared = [tt(p) for p in w["shared"]]
    cls = [tt(p) for p in w["cls"]]
    reg = [tt(p) for p in w["reg"]]
    score, pred = O.task_head_forward(torch.from_numpy(bank), shared, cls, reg, I.G5_TASK_ID, len(I.G5_TASK_SPLIT))
    np.testing.assert_array_equal(np.isinf(score.detach().numpy()), np.isinf(g["cls_score"]))
    fin = np.isfinite(g["cls_score"])
    np.testing.assert_allclose(score.detach().numpy()[fin], g["cls_score"][fin], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pred.detach().numpy(), g["bbox_pred"], rtol=1e-5, atol=1e-6)
    loss = O.replay_loss_from_scores(score, torch.from_numpy(labels), I.G5_TASK_SPLIT[I.G5_TASK_ID])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    loss.backward()
    for i, (W, b) in enumerate(shared):
        np.testing.assert_allclose(W.grad.numpy(), g[f"gW_shared{i}"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(b.grad.numpy(), g[f"gb_shared{i}"], rtol=1e-4, atol=1e-7)
    for i, (W, b) in enumerate(cls):
        assert (W.grad is not None and bool(W.grad.abs().sum() > 0)) == bool(g[f"has_grad_cls{i}"]) or not bool(g[f"has_grad_cls{i}"])
        if bool(g[f"has_grad_cls{i}"]) and W.grad is not None:
            np.testing.assert_allclose(W.grad.numpy(), g[f"gW_cls{i}"], rtol=1e-4, atol=1e-7)


def test_layer_table_sums():
    """50 layers / 118.3 GFLOP / 0.584 GB of P / 0.1063 GB of grads (R-50)."""
    L = O.resnet_fpn_projected_layers(50)
    assert len(L) == 50
    assert abs(sum(2 * c * d * d for _, c, d in L) / 1e9 - 118.3) < 0.05
    assert abs(sum(4 * d * d for _, c, d in L) / 1e9 - 0.584) < 0.001
    assert abs(sum(4 * c * d for _, c, d in L) / 1e9 - 0.1063) < 0.0001
    L = O.resnet_fpn_projected_layers(101)
    assert len(L) == 101
    assert abs(sum(2 * c * d * d for _, c, d in L) / 1e9 - 175.9) < 0.05


def test_g6_ewc_regulariser(golden_dir):
    """EWCHook (runner:1038-1073) through the reference vs the oracle restatement."""
    g = _load(golden_dir, "g6_ewc.npz")
    tensors = I.g6_tensors()
    reg = O.ewc_registered(list(tensors))
    assert sorted(reg) == sorted(str(g[f"name_{k}"]) for k in range(int(g["n_reg"])))
    params = {n: torch.from_numpy(tensors[n][0].copy()).requires_grad_(True) for n in sorted(reg)}
    imp = {n: [torch.from_numpy(a) for a in tensors[n][1]] for n in reg}
    old = {n: [torch.from_numpy(a) for a in tensors[n][2]] for n in reg}
    loss = O.ewc_loss(params, imp, old)
    np.testing.assert_allclose(loss.item(), g["ewc_loss"], rtol=1e-6)
    loss.backward()
    for k in range(int(g["n_reg"])):
        np.testing.assert_allclose(params[str(g[f"name_{k}"])].grad.numpy(), g[f"grad_{k}"], rtol=1e-5, atol=1e-7)


def _g9_sets(filter_fn, rpn_t, roi_t):
    """gt sets the RPN / the RoI head train on after det:65-109, from a (boxes, scores, gt) -> (to_rpn, to_roi) filter."""
    rpn, roi = [], []
    for im in I.g9_batch():
        b, s, g = torch.from_numpy(im["pred_bboxes"]), torch.from_numpy(im["pred_scores"]), torch.from_numpy(im["gt_bboxes"])
        to_rpn, to_roi = filter_fn(b, s, g, rpn_t, roi_t)
        lab, gl = torch.from_numpy(im["pred_labels"]), torch.from_numpy(im["gt_labels"])
        rpn.append((torch.cat([g, b[to_rpn]]).numpy(), np.zeros(len(g) + int(to_rpn.sum()), np.int64)))     # RPN labels are zeroed (det:117-119)
        roi.append((torch.cat([g, b[to_roi]]).numpy(), torch.cat([gl, lab[to_roi]]).numpy()))
    return dict(rpn=rpn, roi=roi)


def test_g9_pseudo_label_loop(golden_dir):
    """8f-2: the oracle's restatement of the teacher pseudo-label loop against what the reference's own ``loss`` handed its RPN and
    RoI heads (G9; IoU supplied by the reference tree's bbox_overlaps because torchvision is absent -- the IoU arithmetic itself
    stays unpinned).  Also shows the fixture needs the growing RoI set: a filter that tests against the ORIGINAL gts only differs."""
    G = np.load(os.path.join(golden_dir, "g9_pseudo_labels.npz"))

    def no_growth(b, s, g, rpn_t, roi_t):
        ok = (O.box_iou(b, g).max(dim=1).values <= 0.7) if len(g) else torch.ones(len(b), dtype=torch.bool)
        return ok & (s > rpn_t), ok & (s > roi_t)
    differs = False
    for ti, (rpn_t, roi_t) in enumerate(I.G9_THRESHOLDS):
        ours, naive = _g9_sets(O.pseudo_label_filter, rpn_t, roi_t), _g9_sets(no_growth, rpn_t, roi_t)
        for which in ("rpn", "roi"):
            for i, (b, l) in enumerate(ours[which]):
                assert np.array_equal(b, G[f"t{ti}__{which}__img{i}__bboxes"]), (ti, which, i)
                assert np.array_equal(l, G[f"t{ti}__{which}__img{i}__labels"]), (ti, which, i)
                differs |= naive[which][i][0].shape != b.shape
    assert differs
