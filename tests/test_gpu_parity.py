"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors produced by the reference.  Run on the GPU box with ``pytest -m gpu``.

THE GATE (north_star: "fp32 projected gradients within 1e-5 rel"), stated once and used by every projection test here:

    for every projected update and EVERY OUTPUT ROW m:   max_n |ours[m,n] - ref[m,n]|  <=  1e-5 * max_n |ref[m,n]|

(``_row_rel``; a row = one output channel of a conv / one output feature of a linear = one row of the reference's
``torch.mm(update.view(Cout, -1), P)``, SGD_NSCL.py:85-90).  Per row, not per tensor: the reference's fp32 mm keeps fp32
relative accuracy in every row however small that row is next to the others, and so must every MFMA path of the product.
An ELEMENTWISE relative bound is not meaningful (entries that cancel to ~0; the reference's own GEMM does not meet one against
fp64).  Where a test can only see ``p`` after ``p += update`` with p != 0, the fp32 rounding of that add (ulp(p)) is allowed on
top and the test says so; the tests that pin the gate itself start from p = 0, where the add is exact.
Integer / bool results (ranks, masks, prototype indices, counts) are bit-exact.
"""
import os

import numpy as np
import pytest
import torch

import inputs as I
import nsgp_oracle as O

pytestmark = pytest.mark.gpu

REL = 1e-5


def _rel(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _row_rel(a, b):
    """The gate's left-hand side: max over rows (dim 0) of max_n|a - b| / max_n|b|; rows where b is all zero must match exactly."""
    a = (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)).double()
    b = (b.detach() if isinstance(b, torch.Tensor) else torch.as_tensor(b)).double().to(a.device)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    err, mag = (a - b).abs().amax(1), b.abs().amax(1)
    assert bool((err[mag == 0] == 0).all()), "a row the reference leaves at zero is not zero"
    return (err[mag > 0] / mag[mag > 0]).max().item() if bool((mag > 0).any()) else 0.0


@pytest.fixture(scope="module")
def N():
    import nsgp_repre_amd
    assert torch.cuda.is_available()
    nsgp_repre_amd.load_library()
    return nsgp_repre_amd


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _key(n):
    return n.replace(".", "_")


# ------------------------------------------------------------------ K1 in isolation
@pytest.mark.parametrize("rows,cols", [(128, 128), (256, 1152), (512, 2304), (12, 40), (16, 147), (100, 324), (130, 260)])
def test_project_matches_oracle(N, dev, rows, cols):
    from nsgp_repre_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + cols)
    a = torch.randn(rows, cols, generator=g)
    C = torch.from_numpy(I.covariance_like(cols, 5, rows_mult=2))
    s, V = O.eigens(C)
    P = O.build_projector(V, O.adaptive_threshold(s, 0.0), True)
    ref = O.project_update(-(0.02 * a), P)                       # the reference's arithmetic (torch CPU fp32)
    ref64 = (-(0.02 * a)).double() @ P.double()
    out = ops.project(a.to(dev), P.to(dev), scale=-0.02)
    assert _rel(out, ref) <= REL
    # and close to the fp64 truth: a k-ordered fp32 fmaf chain (the MFMA's exact semantics) sits
    # within ~sqrt(K)*2^-24 of it; MKL's blocked sums are a little tighter, both far inside 1e-5
    assert _rel(out, ref64) <= 5e-6
    # accumulate form: out += scale * a @ P
    base = torch.randn(rows, cols, generator=g)
    out2 = ops.project(a.to(dev), P.to(dev), scale=-0.02, out=base.clone().to(dev), accumulate=True)
    assert _rel(out2, base + ref) <= REL


def test_project_4d_view_and_errors(N, dev):
    from nsgp_repre_amd import ops
    a = torch.randn(32, 8, 3, 3)
    P = torch.eye(72)
    out = ops.project(a.to(dev), P.to(dev))
    assert out.shape == a.shape and _rel(out, a) <= 1e-7
    with pytest.raises(RuntimeError):
        ops.project(a, P)  # CPU tensors: no fallback
    with pytest.raises(ValueError):
        ops.project(a.to(dev), torch.eye(71, device=dev))


# ------------------------------------------------------------------ a1-a4 against the reference's goldens
KINDS = ["sgd", "sgd_nesterov", "sgdna", "adamw", "adamw_amsgrad", "adam"]


def _make_opt(N, kind, params):
    hp = dict(I.G1_HYPER[kind])
    cls = dict(sgd=N.SGDNSCL, sgd_nesterov=N.SGDNSCL, sgdna=N.SGDNSCLNA, adamw=N.AdamWNSCL,
               adamw_amsgrad=N.AdamWNSCL, adam=N.AdamNSCL)[kind]
    return cls(params, svd=True, **hp)


@pytest.mark.parametrize("kind", KINDS)
def test_g1_steps_with_golden_projectors(N, dev, golden_dir, kind):
    """3 optimizer steps with the reference's own P: isolates K1+K2 from the eigensolver."""
    g = np.load(os.path.join(golden_dir, f"g1_{kind}.npz"))
    gP = np.load(os.path.join(golden_dir, "g1_adam.npz" if kind == "adam" else ("g1_sgdna.npz" if kind == "sgdna" else "g1_sgd.npz")))
    names, _ = I.g1_layers()
    params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in I.g1_params()]
    opt = _make_opt(N, kind, params)
    opt.param_groups[0]["names"] = list(names)
    for n in I.g1_projected():
        opt.transforms[n] = torch.from_numpy(gP[f"P__{_key(n)}"]).to(dev)
    prev = [torch.from_numpy(a) for a in I.g1_params()]
    for step in range(I.G1_STEPS):
        for p, a in zip(params, I.g1_grads(step)):
            p.grad = torch.from_numpy(a).to(dev)
        opt.step()
        torch.cuda.synchronize()
        for n, p, p0 in zip(names, params, prev):
            ref = torch.from_numpy(g[f"p_step{step}__{_key(n)}"])
            # compare the applied update (p_new - p_old), which is what the kernels compute
            upd_ref = ref.double() - p0.double()
            upd = p.detach().cpu().double() - p0.double()
            # 1e-5 of the update + the fp32 rounding of `p += update` itself (2 ulp of |p|)
            allowed = REL * upd_ref.abs().max().item() + 2 * 2.0 ** -23 * ref.abs().max().item()
            assert (upd - upd_ref).abs().max().item() <= allowed, (kind, n, step)
            assert _rel(p, ref) <= REL, (kind, n, step)
            assert _rel(p.grad, g[f"g_step{step}__{_key(n)}"]) <= 1e-6, (kind, n, step, "grad mutation")
        prev = [torch.from_numpy(g[f"p_step{step}__{_key(n)}"]) for n in names]
    for n, p in zip(names, params):
        st = opt.state[p]
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            if sk in st:
                assert _rel(st[sk], g[f"{sk}__{_key(n)}"]) <= 1e-6, (kind, n, sk)


@pytest.mark.parametrize("split", [False, "f16x2"])
@pytest.mark.parametrize("kind", list(I.G1B_KINDS))
def test_g1b_aligned_goldens_per_row(N, dev, golden_dir, kind, split):
    """The reference's own output on 128-aligned layers (G1b) against the kernels the bench times: the whole-tile fp32-MFMA
    kernel, the three-term bf16 split and the two-term fp16 split (256 x 128 LDS-DMA tiles).  Parameters start at zero, so the
    first step's result IS the projected update and the gate is checked row by row with nothing added; gradient rows span six
    decades, so a path that is only accurate relative to the tensor's largest entry fails here."""
    g = np.load(os.path.join(golden_dir, f"g1b_{kind}.npz"))
    gP = np.load(os.path.join(golden_dir, "g1b_sgd.npz"))
    names, _ = I.g1b_layers()
    params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in I.g1b_params()]
    opt = _make_opt(N, kind, params)
    opt.param_groups[0]["names"] = list(names)
    opt.split_mfma = split
    for n in I.g1b_projected():
        opt.transforms[n] = torch.from_numpy(gP[f"P__{_key(n)}"]).to(dev)
    prev = [torch.from_numpy(a) for a in I.g1b_params()]
    for step in range(I.G1B_STEPS):
        for p, a in zip(params, I.g1b_grads(step)):
            p.grad = torch.from_numpy(a).to(dev)
        opt.step()
        torch.cuda.synchronize()
        fast, generic, v2 = opt.tile_counts()
        assert generic == 0 and ((v2 > 0 and fast == 0) if split == "f16x2" else (fast > 0 and v2 == 0)), (fast, generic, v2)
        assert opt.uses_split_mfma() == split
        for n, p, p0 in zip(names, params, prev):
            ref = torch.from_numpy(g[f"p_step{step}__{_key(n)}"])
            if step == 0 and n in opt.transforms:
                assert _row_rel(p, ref) <= REL, (kind, split, n, _row_rel(p, ref))            # THE GATE, nothing added
            upd_ref = ref.double() - p0.double()
            upd = p.detach().cpu().double() - p0.double()
            allowed = REL * upd_ref.abs().max().item() + 2 * 2.0 ** -23 * ref.abs().max().item()
            assert (upd - upd_ref).abs().max().item() <= allowed, (kind, split, n, step)
            if step == 0:
                assert _rel(p.grad, g[f"g_step0__{_key(n)}"]) <= 1e-6, (kind, n, "grad mutation")
        prev = [torch.from_numpy(g[f"p_step{step}__{_key(n)}"]) for n in names]
    for n, p in zip(names, params):
        for sk in ("previous_grad", "exp_avg", "exp_avg_sq"):
            if sk in opt.state[p]:
                assert _rel(opt.state[p][sk], g[f"{sk}__{_key(n)}"]) <= 1e-6, (kind, n, sk)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_g1_eigens_and_transforms_pipeline(N, dev, golden_dir, kind):
    """get_eigens (eigh on the GPU) -> adaptive_threshold -> HIP projector vs the reference's
    torch.svd route.  Ranks are integers and must match; P within 1e-5 of max|P| (two fp32 eigensolvers on a spectrum
    spanning 6 decades; each is within ~4e-6 of the fp64 projector, test_eigensolver_distance_...)."""
    g = np.load(os.path.join(golden_dir, f"g1_{kind}.npz"))
    names, _ = I.g1_layers()
    params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in I.g1_params()]
    opt = _make_opt(N, kind, params)
    opt.param_groups[0]["names"] = list(names)
    fea_in = {n: torch.from_numpy(c).to(dev) for n, c in I.g1_covariances().items()}
    opt.get_eigens(fea_in)
    opt.get_transforms(offset=I.G1_OFFSET)
    assert sorted(opt.transforms.keys()) == sorted(I.g1_projected())
    for n in I.g1_projected():
        sv = opt.eigens[n]["eigen_value"].cpu().numpy()
        ref_sv = g[f"sigma__{_key(n)}"]
        assert np.abs(sv - ref_sv).max() <= 2e-6 * ref_sv.max(), n
        Pref = g[f"P__{_key(n)}"]
        # rank = trace of the un-normalised projector; recover it from the golden for the check
        P = opt.transforms[n].cpu().numpy()
        assert P.shape == Pref.shape
        assert np.abs(P - Pref).max() <= REL * np.abs(Pref).max(), n


@pytest.mark.parametrize("kind,gname", [("sgd", "g1_sgd.npz"), ("adam", "g1_adam.npz"), ("sgd", "g1b_sgd.npz")])
def test_eigensolver_distance_is_the_references_own_conditioning(N, dev, golden_dir, kind, gname):
    """a5: the product's projector (rocSOLVER ``eigh`` in fp32 -> elbow -> HIP SYRK) differs from the
    reference's (LAPACK ``gesdd`` in fp32 via torch.svd) by up to 1e-4 of max|P|.  Both are fp32 decompositions of a matrix
    whose spectrum spans six decades; judged against an fp64 decomposition of the SAME covariance (the projector on the exact
    null space at the same rank), product and reference are BOTH within 1e-5 of max|P| on every layer (measured 1-4e-6 and
    0.7-2.4e-6; round 1's 1e-4 allowance was never needed -- the elbow sits at the head of the spectrum, where the gap to the
    tail makes the projector well conditioned).  Ranks (integers) are equal on every layer, from the product's spectrum, the
    reference's and the fp64 one."""
    from nsgp_repre_amd.optim.threshold import elbow_index
    g = np.load(os.path.join(golden_dir, gname))
    aligned = gname.startswith("g1b")
    names, _ = I.g1b_layers() if aligned else I.g1_layers()
    params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in (I.g1b_params() if aligned else I.g1_params())]
    opt = _make_opt(N, kind, params)
    opt.param_groups[0]["names"] = list(names)
    covs = I.g1b_covariances() if aligned else I.g1_covariances()
    opt.get_eigens({n: torch.from_numpy(c).to(dev) for n, c in covs.items()})
    opt.get_transforms(offset=I.G1_OFFSET)
    rule = "sgd" if kind == "sgd" else "adam"
    worst = []
    for n in (I.g1b_projected() if aligned else I.g1_projected()):
        C64 = torch.from_numpy(covs[n]).double().to(dev)
        lam, Q = torch.linalg.eigh(C64)
        sv64 = lam.abs().flip(0)
        V64 = Q.flip(1)
        first = int(elbow_index(opt.eigens[n]["eigen_value"].cpu().numpy(), I.G1_OFFSET, rule))
        assert first == int(elbow_index(g[f"sigma__{_key(n)}"], I.G1_OFFSET, rule)) == int(elbow_index(sv64.float().cpu().numpy(), I.G1_OFFSET, rule)), n
        P64 = V64[:, first:] @ V64[:, first:].t()
        if kind == "adam" or "backbone" in n:
            P64 = P64 / P64.norm()
        ours = (opt.transforms[n].double() - P64).abs().max().item() / P64.abs().max().item()
        ref = (torch.from_numpy(g[f"P__{_key(n)}"]).to(dev).double() - P64).abs().max().item() / P64.abs().max().item()
        worst.append((n, ours, ref))
        assert ours <= REL and ref <= REL, (n, ours, ref)          # both fp32 solvers sit inside the 1e-5 gate around the fp64 projector
    out_dir = os.environ.get("NSGP_REPORT_DIR")
    if out_dir:
        import json
        os.makedirs(out_dir, exist_ok=True)
        json.dump([dict(layer=n, product_vs_fp64=o, reference_vs_fp64=r) for n, o, r in worst],
                  open(os.path.join(out_dir, f"eigensolver_vs_fp64_{gname[:-4]}_{kind}.json"), "w"), indent=1)


def test_sgdna_pipeline_on_a_gapped_spectrum(N, dev):
    """SGDNSCLNA cuts at ``sigma <= sigma_min * thres`` (SGD_NSCL_NoAdaptive.py:157-158).  On the
    G1 covariances (spectrum over 6 decades) sigma_min sits at the fp32 noise floor of ANY
    solver, so the cut -- and P -- is not reproducible between LAPACK gesdd and rocSOLVER syevd
    (measured: max|dP| 0.06); the golden P for sgdna is therefore only used with the reference's
    own projector (test above).  Here the rule is checked end to end on a spectrum with a clean
    gap at the cut, against the oracle's torch.svd route."""
    D = 160
    rng = np.random.default_rng(12)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.concatenate([np.linspace(100, 10, 24), np.linspace(1.2, 1.0, D - 24)])
    C = torch.from_numpy(((Q * lam) @ Q.T).astype(np.float32))
    C = ((C + C.t()) / 2).contiguous()
    s, V = O.eigens(C)
    mask = O.na_threshold(s, 1.5)
    assert int(mask.sum()) == D - 24
    ref = O.build_projector(V, mask, True)
    p = torch.nn.Parameter(torch.zeros(8, D, 1, 1, device=dev))
    opt = N.SGDNSCLNA([p], lr=0.1, svd=True, thres=1.5)
    opt.param_groups[0]["names"] = ["backbone.x.weight"]
    opt.get_eigens({"backbone.x.weight": C.to(dev)})
    opt.get_transforms()
    assert _rel(opt.transforms["backbone.x.weight"], ref) <= 1e-4


@pytest.mark.parametrize("D,first,norm", [(128, 20, True), (256, 33, False), (147, 22, True), (324, 29, False), (40, 1, True), (256, 0, False), (256, 255, True)])
def test_build_projector_vs_oracle(N, dev, D, first, norm):
    from nsgp_repre_amd import ops
    C = torch.from_numpy(I.covariance_like(D, 9, rows_mult=2))
    _, V = O.eigens(C)
    mask = torch.zeros(D, dtype=torch.bool)
    mask[first:] = True
    ref = O.build_projector(V, mask, norm)
    P = ops.build_projector(V.contiguous().to(dev), first, norm)
    assert _rel(P, ref) <= REL
    assert torch.equal(P, P.t().contiguous())  # exactly symmetric by construction


# ------------------------------------------------------------------ a8/a9 covariance
def test_g3_covariance_vs_golden(N, dev, golden_dir):
    from nsgp_repre_amd import ops
    g = np.load(os.path.join(golden_dir, "g3_covariance.npz"))
    for ci, cfg in enumerate(I.g3_cases()):
        cov = None
        for x in I.g3_inputs(ci):
            xt = torch.from_numpy(x).to(dev)
            if cfg["kind"] == "conv":
                cov = ops.cov_accumulate_conv2d(xt, cfg["k"], cfg["s"], cfg["p"], cov)
            else:
                cov = ops.cov_accumulate_linear(xt, cov)
        assert _rel(cov, g[f"C_{ci}"]) <= REL, ci


@pytest.mark.parametrize("cin,k,s,p,hw,B", [(64, (1, 1), (1, 1), (0, 0), (50, 84), 2), (32, (3, 3), (1, 1), (1, 1), (25, 42), 1),
                                            (16, (3, 3), (2, 2), (1, 1), (50, 84), 2), (3, (7, 7), (2, 2), (3, 3), (64, 96), 1),
                                            (256, (1, 1), (1, 1), (0, 0), (13, 21), 1),
                                            (64, (3, 3), (1, 1), (1, 1), (25, 42), 2),      # D = 576 = 4.5 x 128: padded rows + a 128-row remainder tile
                                            (512, (1, 1), (1, 1), (0, 0), (13, 21), 1),     # D = 512, L = 273 (padded to 288)
                                            (128, (3, 3), (2, 2), (1, 1), (50, 84), 1),     # D = 1152, stride 2
                                            (192, (3, 1), (1, 2), (0, 1), (20, 30), 3)])    # D = 576, non-square kernel / stride / padding, batch 3
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_covariance_vs_oracle_mid_sizes(N, dev, cin, k, s, p, hw, B, mode):
    """All matrix-core paths of the SYRK: 0 = fp32 MFMA; 2 = two-term fp16 split forced (layers with D >= 512 take its second
    generation: materialised pre-tiled operand + LDS-DMA tiles, split-K); 3 = the split's first generation (gather kernel) for
    every layer.  The default picks by size."""
    from nsgp_repre_amd import ops
    x = torch.randn(B, cin, *hw, generator=torch.Generator().manual_seed(cin)).abs()
    ref = O.cov_conv2d(x, k, s, p)
    prev = ops.cov_set_split_mfma(mode)
    try:
        cov = ops.cov_accumulate_conv2d(x.to(dev), k, s, p)
        assert _rel(cov, ref) <= REL and _row_rel(cov, ref) <= REL
        assert torch.equal(cov, ops.cov_accumulate_conv2d(x.to(dev), k, s, p))     # no float atomics: bitwise repeatable
        cov = ops.cov_accumulate_conv2d(x.to(dev), k, s, p, cov)  # second batch accumulates
        assert _rel(cov, ref + ref) <= REL
        assert torch.equal(cov, cov.t().contiguous())
        # activations 1e+4 / 1e-6 times the usual size, and all zeros: the per-layer fp16 scale follows
        for f in (1e4, 1e-6, 0.0):
            c2 = ops.cov_accumulate_conv2d((x * f).to(dev), k, s, p)
            assert torch.isfinite(c2).all() and _rel(c2, ref * f * f) <= REL
    finally:
        ops.cov_set_split_mfma(prev)


_TRUE_SIZE = {   # (cin, kernel, stride, pad, H, W) at an 800 x 1344 padded image (bench.py r50_fpn_hooked_convs)
    "backbone.conv1": (3, 7, 2, 3, 800, 1344),            # L = 268,800 (the longest contraction), D = 147 (ragged)
    "neck.fpn_convs.0.conv": (256, 3, 1, 1, 200, 336),    # L = 67,200, D = 2304: 713 GFLOP, the largest SYRK
    "backbone.layer4.0.conv2": (512, 3, 2, 1, 50, 84),    # L = 1,050, D = 4608: the widest covariance
}
_cov_true = {}


def test_grouped_covariance_pass_vs_single_layer_path_and_oracle(N, dev):
    """The grouped pass (ops.CovGroupPlan: all layers of one forward in five launches, one tile table, no split-K, direct epilogue)
    on a mixed bag of geometries -- 1x1 / 3x3, strides 1 / 2, padding, batch 1 / 2 / 16 (batch mean first), D = 64 ... 1152, an L that is
    not a multiple of 32, a D that pads to the next 128 -- against the oracle's unfold + mm (1e-5 per row), bit-symmetric, bitwise
    reproducible run to run, accumulation over a second batch, and the layer the tile cannot take (D = 147) routed to the
    single-layer entry point."""
    from nsgp_repre_amd import ops
    geoms = [(1, 64, 40, 56, (1, 1), (1, 1), (0, 0)), (2, 64, 40, 56, (3, 3), (1, 1), (1, 1)), (1, 128, 20, 28, (3, 3), (2, 2), (1, 1)),
             (16, 192, 9, 13, (1, 1), (1, 1), (0, 0)), (1, 256, 20, 28, (1, 1), (2, 2), (0, 0)), (2, 320, 10, 14, (1, 1), (1, 1), (0, 0)),
             (1, 3, 64, 64, (7, 7), (2, 2), (3, 3))]
    g = torch.Generator().manual_seed(12)
    xs = [(torch.randn(b, c, h, w, generator=g).abs() * torch.pow(10.0, -2.0 * (torch.arange(c) % 5) / 4.0).view(1, c, 1, 1)) for b, c, h, w, *_ in geoms]
    plan = ops.CovGroupPlan(geoms, dev)
    assert plan.routes == [True] * 6 + [False] and plan.n_grouped == 6
    xd = [x.to(dev) for x in xs]
    covs = plan.run(xd, [None] * len(geoms))
    again = plan.run(xd, [None] * len(geoms))
    for i, (x, gm) in enumerate(zip(xs, geoms)):
        if not plan.routes[i]:
            assert covs[i] is None
            continue
        ref = O.cov_conv2d(x, gm[4], gm[5], gm[6])
        assert _row_rel(covs[i], ref) <= REL and _rel(covs[i], ref) <= REL, (i, _row_rel(covs[i], ref))
        assert torch.equal(covs[i], covs[i].t().contiguous()), i
        assert torch.equal(covs[i], again[i]), (i, "not reproducible")
    # second batch: accumulate in place (the first occurrence assigned)
    x2 = [(x * 0.5 + 0.1).to(dev) for x in xs]
    acc = plan.run(x2, [c.clone() if c is not None else None for c in covs])
    for i, (x, gm) in enumerate(zip(xs, geoms)):
        if plan.routes[i]:
            ref = O.cov_conv2d(x, gm[4], gm[5], gm[6]) + O.cov_conv2d(x * 0.5 + 0.1, gm[4], gm[5], gm[6])
            assert _rel(acc[i], ref) <= REL, i
    with pytest.raises(ValueError):
        plan.run(xd, [covs[0]] * len(geoms))        # the covariances of one run must be distinct buffers
    plan.close()
    with pytest.raises(RuntimeError):
        plan.run(xd, [None] * len(geoms))
    # through the hooks: grouped (default) and hook-time collectors agree with each other within the gate, both with the oracle
    import torch.nn as nn
    net = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 1), nn.ReLU(), nn.Conv2d(128, 64, 3, stride=2, padding=1)).to(dev)
    xin = torch.randn(2, 64, 24, 32, generator=g).abs().to(dev)
    res = {}
    for grouped in (True, False):
        col = N.runner.CovarianceCollector(net, [], grouped=grouped).register()
        with torch.no_grad():
            net(xin)
            col.flush()
            net(xin * 2)
        col.remove()
        col.close()
        res[grouped] = col.fea_in
    assert sorted(res[True]) == sorted(res[False]) == ["0.weight", "2.weight", "4.weight"]
    for k in res[True]:
        assert _rel(res[True][k], res[False][k]) <= REL, k
    # the same through the hooks with the correlation form forced for the 3x3 / 1 / 1 layer (the rule leaves a 24 x 32 map alone)
    prev = ops.cov_set_corr_mode(2)
    try:
        col = N.runner.CovarianceCollector(net, [], grouped=True).register()
        with torch.no_grad():
            net(xin)
            col.flush()
            assert any(p.n_correlation_form == 1 for p in col._plans.values())
            net(xin * 2)
        col.remove()
        col.close()
    finally:
        ops.cov_set_corr_mode(prev)
    for k in res[False]:
        assert _rel(col.fea_in[k], res[False][k]) <= REL, k
    # inputs of changing size (real loaders pad to many shapes): one plan per geometry, at most MAX_PLANS alive, ONE shared workspace
    col = N.runner.CovarianceCollector(net, [], grouped=True).register()
    col.MAX_PLANS = 2
    want = None
    with torch.no_grad():
        for hw in ((24, 32), (16, 40), (24, 32), (32, 24), (16, 40)):
            xi = torch.randn(1, 64, *hw, generator=g).abs().to(dev)
            net(xi)
            col.flush()
            assert len(col._plans) <= 2 and col._group_ws is not None
            ref = O.cov_conv2d(xi.cpu(), (3, 3), (1, 1), (1, 1))
            want = ref if want is None else want + ref
    col.remove()
    assert _rel(col.fea_in["0.weight"], want) <= REL
    col.close()
    assert col._plans == {} and col._group_ws is None


def test_grouped_covariance_correlation_form(N, dev):
    """The correlation form of the 3x3 / stride 1 / padding 1 layers (csrc/covariance.hip: Cov = the 25 shifted C x C correlations
    laid out over the 81 tap pairs, minus the covariance of the ring of positions the shifts add) FORCED on geometries the rule would
    leave alone: 64 channels (half of a 128-row tile is junk), 192 and 320 (a 3-block / 1-block remainder tile), widths that pad to
    the next 32 with and without room, a 6 x 6 map (mostly ring), a contraction long enough to be cut into ranges, batch mean first --
    against the oracle's unfold + mm at 1e-5 per row, bit-symmetric, bitwise reproducible, accumulating, and within the gate of
    the im2col form of the same plan."""
    from nsgp_repre_amd import ops
    k3 = ((3, 3), (1, 1), (1, 1))
    geoms = [(1, 64, 40, 56) + k3, (2, 64, 25, 42) + k3, (1, 128, 20, 28) + k3, (3, 192, 9, 13) + k3, (1, 256, 13, 21) + k3,
             (1, 320, 10, 14) + k3, (1, 64, 6, 6) + k3, (1, 64, 64, 200) + k3, (1, 128, 30, 60, (1, 1), (1, 1), (0, 0))]
    g = torch.Generator().manual_seed(31)
    xs = [(torch.randn(b, c, h, w, generator=g).abs() * torch.pow(10.0, -2.0 * (torch.arange(c) % 5) / 4.0).view(1, c, 1, 1)) for b, c, h, w, *_ in geoms]
    xd = [x.to(dev) for x in xs]
    refs = [O.cov_conv2d(x, gm[4], gm[5], gm[6]) for x, gm in zip(xs, geoms)]
    res = {}
    for mode in (2, 0):
        prev = ops.cov_set_corr_mode(mode)
        try:
            plan = ops.CovGroupPlan(geoms, dev)
        finally:
            ops.cov_set_corr_mode(prev)
        assert plan.routes == [True] * len(geoms)
        assert plan.n_correlation_form == (8 if mode == 2 else 0)
        covs = plan.run(xd, [None] * len(geoms))
        again = plan.run(xd, [None] * len(geoms))
        for i, ref in enumerate(refs):
            assert _row_rel(covs[i], ref) <= REL and _rel(covs[i], ref) <= REL, (mode, i, _row_rel(covs[i], ref))
            assert torch.equal(covs[i], covs[i].t().contiguous()), (mode, i)
            assert torch.equal(covs[i], again[i]), (mode, i, "not reproducible")
        x2 = [(x * 0.5 + 0.1).to(dev) for x in xs]
        acc = plan.run(x2, [c.clone() for c in covs])
        for i, (x, gm) in enumerate(zip(xs, geoms)):
            assert _rel(acc[i], refs[i] + O.cov_conv2d(x * 0.5 + 0.1, gm[4], gm[5], gm[6])) <= REL, (mode, i)
        # wide dynamic range and zeros: the per-layer scale follows
        for f in (1e4, 1e-6, 0.0):
            c2 = plan.run([x * f for x in xd], [None] * len(geoms))
            for i, ref in enumerate(refs):
                assert torch.isfinite(c2[i]).all() and _rel(c2[i], ref * f * f) <= REL, (mode, i, f)
        plan.close()
        res[mode] = covs
    for i in range(len(geoms)):
        assert _rel(res[2][i], res[0][i]) <= REL, i
    # the rule on its own: large maps take the form, small ones and everything that is not 3x3 / 1 / 1 do not
    plan = ops.CovGroupPlan([(1, 64, 100, 168) + k3, (1, 512, 7, 11) + k3, (1, 64, 100, 168, (3, 3), (2, 2), (1, 1))], dev)
    assert plan.n_correlation_form == 1
    plan.close()


def test_correlation_form_on_random_geometries(N, dev):
    """Seeded sweep: channels 64 ... 448 (every remainder of the 256-channel row tiles), maps 4 ... 40 on a side (widths on both sides of a
    multiple of 32 after the + 4 halo), batch 1 ... 3 -- the forced correlation form against the im2col form of the same plan (both within
    the gate of the oracle) and bit-symmetric."""
    from nsgp_repre_amd import ops
    rs = np.random.default_rng(77)
    k3 = ((3, 3), (1, 1), (1, 1))
    geoms = []
    for c in (64, 128, 192, 256, 320, 384, 448):
        h, w = int(rs.integers(4, 41)), int(rs.choice([4, 12, 27, 28, 29, 40, 59, 60, 61]))
        if h * w < 32:
            h = 8
        geoms.append((int(rs.integers(1, 4)), c, h, w) + k3)
    g = torch.Generator().manual_seed(78)
    xs = [torch.randn(b, c, h, w, generator=g).abs() for b, c, h, w, *_ in geoms]
    xd = [x.to(dev) for x in xs]
    res = {}
    for mode in (2, 0):
        prev = ops.cov_set_corr_mode(mode)
        try:
            plan = ops.CovGroupPlan(geoms, dev)
        finally:
            ops.cov_set_corr_mode(prev)
        assert plan.n_correlation_form == (len(geoms) if mode == 2 else 0)
        res[mode] = plan.run(xd, [None] * len(geoms))
        torch.cuda.synchronize()
        plan.close()
    for i, (x, gm) in enumerate(zip(xs, geoms)):
        ref = O.cov_conv2d(x, gm[4], gm[5], gm[6])
        assert _row_rel(res[2][i], ref) <= REL and _row_rel(res[0][i], ref) <= REL, (gm, _row_rel(res[2][i], ref), _row_rel(res[0][i], ref))
        assert torch.equal(res[2][i], res[2][i].t().contiguous()), gm


@pytest.mark.parametrize("mode", [0, 2, 3, 4])
@pytest.mark.parametrize("name", list(_TRUE_SIZE))
def test_covariance_at_true_size(N, dev, name, mode):
    """full-size covariance parity was unpinned (tests stopped at 64 x 96 inputs; the 7 x 7 stem differed from
    rocBLAS by 1.78e-5 with nobody knowing which side was off).  Three R-50-FPN layers at 800 x 1344, both matrix-core paths
    (0 = fp32 MFMA, 2 = two-term fp16 split: second generation for D >= 512, 3 = its first-generation gather kernel), channels spanning three decades, against (i) an fp64 product of the materialised
    unfold on the GPU -- tensor-max and PER ROW -- and (ii) the ORACLE (the reference's fp32 unfold + mm on the CPU), whose own
    distance from fp64 is measured beside ours; the adaptive elbow (an INTEGER that decides the projector's rank) must be
    the same for the fp32 and the fp16-split covariance.  NSGP_REPORT_DIR collects the numbers (profiles/r02/covariance_true_size.json)."""
    import json
    import torch.nn.functional as F
    from nsgp_repre_amd import ops
    from nsgp_repre_amd.optim.threshold import elbow_index
    cin, k, st, pd, H, W = _TRUE_SIZE[name]
    g = torch.Generator().manual_seed(cin + k)
    x = torch.randn(1, cin, H, W, generator=g).abs() * torch.pow(10.0, -3.0 * (torch.arange(cin) % 8) / 7.0).view(1, cin, 1, 1)
    xd = x.to(dev)
    if mode == 4:       # the grouped pass (what cal_fea_in runs since round 3): one tile table, no split-K, direct epilogue
        plan = ops.CovGroupPlan([(1, cin, H, W, (k, k), (st, st), (pd, pd))], dev)
        if not plan.routes[0]:
            assert (cin * k * k) % 64 != 0       # the 7x7 stem stays on the single-layer entry point
            plan.close()
            return
        assert plan.n_correlation_form == int(name == "neck.fpn_convs.0.conv")     # by rule: the 3x3 / 1 / 1 layer on the large map
        cov = plan.run([xd], [None])[0]
        torch.cuda.synchronize()
        plan.close()
    else:
        prev = ops.cov_set_split_mfma(mode)
        try:
            cov = ops.cov_accumulate_conv2d(xd, (k, k), (st, st), (pd, pd))
        finally:
            ops.cov_set_split_mfma(prev)
    X64 = F.unfold(xd.double(), k, padding=pd, stride=st)[0].t().contiguous()        # [L x D] fp64
    ref64 = X64.t() @ X64
    del X64
    key = (name, "ref")
    if key not in _cov_true:
        _cov_true[key] = O.cov_conv2d(x, (k, k), (st, st), (pd, pd))                   # the oracle, once per layer
    orc = _cov_true[key].to(dev)
    rec = dict(layer=name, path={0: "f32", 2: "f16x2", 3: "f16x2-gather", 4: "grouped"}[mode], L=int((H + 2 * pd - k) // st + 1) * int((W + 2 * pd - k) // st + 1), D=cin * k * k,
               ours_vs_fp64_tensor_rel=_rel(cov, ref64), ours_vs_fp64_row_rel=_row_rel(cov, ref64),
               oracle_vs_fp64_tensor_rel=_rel(orc, ref64), oracle_vs_fp64_row_rel=_row_rel(orc, ref64),
               ours_vs_oracle_tensor_rel=_rel(cov, orc), ours_vs_oracle_row_rel=_row_rel(cov, orc))
    assert torch.equal(cov, cov.t().contiguous())
    assert rec["ours_vs_fp64_tensor_rel"] <= REL and rec["ours_vs_fp64_row_rel"] <= REL, rec
    assert rec["ours_vs_oracle_row_rel"] <= REL + rec["oracle_vs_fp64_row_rel"], rec    # the oracle's own fp32 error is not ours to carry
    # the rank decision (an integer from an argmax over smoothed second differences of the spectrum): the same elbow from this
    # path's covariance as from the fp64 one -- wherever that argmax is decidable at fp32 precision at all, i.e. wherever
    # relative perturbations of 2e-6 (the level at which the ORACLE's own fp32 covariance differs from fp64) do not move the
    # fp64 elbow itself; where they do, two neighbouring indices tie and either is what the reference's fp32 path may produce
    sv = torch.linalg.eigvalsh(cov.double()).abs().flip(0).float().cpu().numpy()
    sv64d = torch.linalg.eigvalsh(ref64).abs().flip(0).cpu().numpy()
    sv_orc = torch.linalg.eigvalsh(orc.double()).abs().flip(0).float().cpu().numpy()
    e64 = int(elbow_index(sv64d.astype(np.float32), 0.0, "sgd"))
    rs = np.random.default_rng(5)
    wobble = {int(elbow_index((sv64d * (1 + 2e-6 * rs.standard_normal(sv64d.shape))).astype(np.float32), 0.0, "sgd")) for _ in range(16)}
    rec.update(elbow=int(elbow_index(sv, 0.0, "sgd")), elbow_fp64=e64, elbow_oracle=int(elbow_index(sv_orc, 0.0, "sgd")),
               elbow_fp64_under_2e6_perturbation=sorted(wobble | {e64}))
    if wobble <= {e64}:
        assert rec["elbow"] == e64, rec
    else:
        assert rec["elbow"] in (wobble | {e64}), rec
    if mode == 4:
        # the route a layer takes is decided by rule (D % 64 == 0 -> grouped pass), and its INTEGER must be the fp32 path's: the rank
        # of round 2's gather kernel (114 where fp32 / fp64 / the oracle say 115 on the rank-deficient layer4 case) cannot come back
        prev = ops.cov_set_split_mfma(0)
        try:
            cov32 = ops.cov_accumulate_conv2d(xd, (k, k), (st, st), (pd, pd))
        finally:
            ops.cov_set_split_mfma(prev)
        e32 = int(elbow_index(torch.linalg.eigvalsh(cov32.double()).abs().flip(0).float().cpu().numpy(), 0.0, "sgd"))
        rec["elbow_fp32_path"] = e32
        assert rec["elbow"] == e32, rec
    out_dir = os.environ.get("NSGP_REPORT_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        f = os.path.join(out_dir, "covariance_true_size.json")
        rows = json.load(open(f)) if os.path.exists(f) else []
        json.dump([r for r in rows if (r["layer"], r["path"]) != (rec["layer"], rec["path"])] + [rec], open(f, "w"), indent=1)


# ------------------------------------------------------------------ a15 prototypes
def test_g4_prototype_bank_vs_golden(N, dev, golden_dir):
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    g = np.load(os.path.join(golden_dir, "g4_prototypes.npz"))
    feats, cls = I.g4_rois()
    feats, cls = torch.from_numpy(feats).to(dev), torch.from_numpy(cls).to(dev)
    bank, labels, masks, centres = build_prototype_bank(feats, cls, I.G4_TASK_SPLIT, 2, I.G4_MAX_PROTO)
    np.testing.assert_array_equal(labels.cpu().numpy(), g["labels"])          # bit-exact
    for c in range(3):
        assert len(masks[c]) == int(g[f"nmask_{c}"])
        for j, m in enumerate(masks[c]):
            np.testing.assert_array_equal(m.numpy(), g[f"mask_{c}_{j}"])      # bit-exact
    assert _rel(bank, g["bank"]) <= REL


def test_sim_counts_vs_oracle_ragged(N, dev):
    """Counts and the bit matrix are integer results: bit-exact vs the oracle wherever no
    similarity sits within 1e-5 of the threshold (checked in fp64)."""
    from nsgp_repre_amd import ops
    # the last four reach the stream-K split (few tiles, long D; fast and guarded loads) and the direct kernel with
    # clamped fast loads on a ragged N
    for n, d, seed in ((1, 64, 1), (5, 100, 2), (64, 256, 3), (129, 224, 4), (300, 512, 5), (300, 4096, 6), (700, 1000, 7),
                       (1, 12544, 8), (3000, 64, 9)):
        Fc = torch.from_numpy(I.class_rois(n, d, seed, n_clusters=3))
        nrm = Fc / Fc.norm(dim=-1, keepdim=True)
        sim = nrm @ nrm.t()
        sim64 = Fc.double() / Fc.double().norm(dim=-1, keepdim=True)
        sim64 = sim64 @ sim64.t()
        ambiguous = (sim64 - 0.6).abs() < 1e-5
        counts, bitmask = ops.sim_counts(Fc.to(dev), 0.6)
        got = torch.stack([ops.unpack_bitmask_row(bitmask[i], n) for i in range(n)])
        ref = sim >= 0.6
        assert torch.equal(got | ambiguous, ref | ambiguous), (n, d)
        if not ambiguous.any():
            assert torch.equal(counts.cpu(), ref.long().sum(-1))
        assert torch.equal(got, got.t())


# ------------------------------------------------------------------ full-size properties
def test_full_size_step_properties(N, dev):
    """R-50-FPN layer shapes: (i) with P = I the projected step equals the un-projected one
    bit for bit; (ii) with a true projector, projecting twice equals projecting once
    (idempotence) within fp32 rounding; (iii) linearity in the update."""
    from nsgp_repre_amd import ops
    rows, cols = 512, 2304
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn(rows, cols, generator=g).to(dev)
    eye = torch.eye(cols, device=dev)
    assert torch.equal(ops.project(a, eye), a)
    Q, _ = torch.linalg.qr(torch.randn(cols, 300, generator=g).to(dev))
    P = (eye - Q @ Q.t()).contiguous()
    once = ops.project(a, P)
    twice = ops.project(once, P)
    assert _rel(twice, once) <= 5e-5
    b = torch.randn(rows, cols, generator=g).to(dev)
    lin = ops.project(a + b, P)
    assert _rel(lin, once + ops.project(b, P)) <= 1e-5
    ref = O.project_update(a.cpu(), P.cpu())        # the oracle (torch-CPU fp32 mm, the reference's arithmetic) at full size
    assert _row_rel(once, ref) <= REL


def _table_projectors(dev, layers):
    """One projector per (D, Frobenius-normalised?) pair of the table, built the way the product builds them in a run
    (C = X^T X, X = [4D x D] ~ N(0,1) diag(logspace(0,-3,D)); eigh -> elbow -> HIP projector kernel) and shared by
    the layers of that shape.  The step kernels are judged GIVEN P, so the oracle gets a CPU copy of the same matrices."""
    from nsgp_repre_amd import ops
    from nsgp_repre_amd.optim.threshold import elbow_index
    cache = {}
    for n, cout, D in layers:
        key = (D, "backbone" in n)
        if key in cache:
            continue
        gen = torch.Generator(device=dev).manual_seed(2000 + D)
        X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
        lam, Q = torch.linalg.eigh((X.t() @ X).contiguous())
        sv = lam.abs()
        order = torch.argsort(sv, descending=True, stable=True)
        first = elbow_index(sv[order].cpu().numpy(), 0.0, "sgd")
        cache[key] = ops.build_projector(Q[:, order].contiguous(), int(first), key[1])
        del X, lam, Q
    return cache


@pytest.mark.parametrize("split", [False, "f16x2"])
@pytest.mark.parametrize("depth", [50, 101])
def test_full_table_one_step_vs_oracle_per_row(N, dev, depth, split):
    """The complete projected-layer table of R-50-FPN (BASELINE configs[1]: 50 layers, 118.3 GFLOP) and R-101-FPN (configs[4]:
    101 layers, 175.9 GFLOP) in one plan, one SGDNSCL step, every MFMA path, against the ORACLE run over the same table on the
    CPU -- row by row under THE GATE.  Projected parameters start at zero (the step's result is the projected update itself);
    gradient rows span three decades.  With NSGP_REPORT_DIR set, the measured errors of every layer -- against the oracle and,
    for both, against an fp64 product -- are written there (profiles/r02/parity_*.json come from this test)."""
    import json
    layers = O.resnet_fpn_projected_layers(depth)
    Pc = _table_projectors(dev, layers)
    gen = torch.Generator(device="cpu").manual_seed(7 + depth)
    params, names, shapes, Ps = [], [], {}, {}
    for n, cout, D in layers:
        k = 3 if ("conv2" in n or "fpn_convs" in n) else 1
        shapes[n] = (cout, D // (k * k), k, k)
        params.append(torch.nn.Parameter(torch.zeros(shapes[n], device=dev)))
        names.append(n)
        Ps[n] = Pc[(D, "backbone" in n)]
    params.append(torch.nn.Parameter(torch.randn(1000, generator=gen).to(dev)))
    names.append("backbone.bn.weight")
    hp = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    opt = N.SGDNSCL(params, svd=True, **hp)
    opt.param_groups[0]["names"] = names
    opt.transforms.update(Ps)
    opt.split_mfma = split
    grads = []
    for n, p in zip(names, params):
        gr = torch.randn(p.shape, generator=gen) * 1e-3
        if n in Ps:
            gr *= torch.pow(10.0, -3.0 * (torch.arange(p.shape[0]) % 16) / 15.0).view(-1, 1, 1, 1)
        grads.append(gr)
    cpu_params = [p.detach().cpu().clone() for p in params]
    for p, gr in zip(params, grads):
        p.grad = gr.clone().to(dev)
    opt.step()
    torch.cuda.synchronize()
    flops, nbytes, ntiles, nproj = opt.plan_stats()
    fast, generic, v2 = opt.tile_counts()
    want_gf = {50: 118.3, 101: 175.9}[depth]
    assert nproj == len(layers) and abs(flops / 1e9 - want_gf) < 0.06 and generic == 0
    assert opt.uses_split_mfma() == split and ((v2 == {50: 836, 101: 1414}[depth] and fast == 0) if split == "f16x2" else (fast == {50: 1622, 101: 2778}[depth] and v2 == 0))
    tr_cpu = {k: v.cpu() for k, v in Pc.items()}
    O.sgd_nscl_step(names, cpu_params, [g.clone() for g in grads], [dict() for _ in names],
                    {n: tr_cpu[(shapes[n][1] * shapes[n][2] * shapes[n][3], "backbone" in n)] for n in Ps}, **hp)
    report, worst = [], 0.0
    for n, p, q, gr in zip(names, params, cpu_params, grads):
        if n not in Ps:
            assert _rel(p, q) <= 1e-6, n
            continue
        vs_oracle = _row_rel(p, q)
        worst = max(worst, vs_oracle)
        u64 = (-(0.02 * gr.to(dev))).double().view(gr.shape[0], -1) @ Ps[n].double()      # p0 = 0: weight decay adds nothing, buf = g
        rec = dict(layer=n, rows=gr.shape[0], D=Ps[n].shape[0], vs_oracle_row_rel=vs_oracle,
                   vs_fp64_row_rel=_row_rel(p, u64), vs_fp64_tensor_rel=_rel(p.detach().view(gr.shape[0], -1), u64),
                   oracle_vs_fp64_row_rel=_row_rel(q.to(dev), u64))
        report.append(rec)
        assert vs_oracle <= REL, rec
    out_dir = os.environ.get("NSGP_REPORT_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        summary = dict(table=f"R-{depth}-FPN", path=split or "f32", gate="max_n|ours-ref| <= 1e-5 * max_n|ref| per output row",
                       worst_row_rel_vs_oracle=worst, worst_row_rel_vs_fp64=max(r["vs_fp64_row_rel"] for r in report),
                       worst_tensor_rel_vs_fp64=max(r["vs_fp64_tensor_rel"] for r in report),
                       oracle_worst_row_rel_vs_fp64=max(r["oracle_vs_fp64_row_rel"] for r in report), layers=report)
        json.dump(summary, open(os.path.join(out_dir, f"parity_r{depth}_{split or 'f32'}.json"), "w"), indent=1)


# ------------------------------------------------------------------ low-rank form (the default for projectors the optimizer builds)
@pytest.mark.parametrize("kind", ["sgd", "adamw"])
def test_low_rank_form_matches_dense_form(N, dev, kind):
    """``low_rank=True`` (default) builds P = c (I - U U^T) and applies p += c*(u - (u U)U^T); ``low_rank=False`` builds
    V_tail V_tail^T and runs the dense GEMM u @ P (the reference's literal formula): same optimizer, same covariances, two
    parameter copies -> the applied updates must agree within the 1e-5 gate, and the layers that qualify (r <= 128,
    32-multiples) must really take the low-rank route."""
    shapes = {"backbone.a.weight": (256, 128, 3, 3), "neck.b.weight": (128, 512, 1, 1), "backbone.c.weight": (128, 256, 1, 1),
              "backbone.small.weight": (16, 8, 3, 3), "backbone.bn.weight": (256,)}
    gen = torch.Generator().manual_seed(5)
    init = {n: torch.randn(s, generator=gen) * 0.02 for n, s in shapes.items()}
    covs = {}
    for i, (n, s) in enumerate(shapes.items()):
        if len(s) == 4:
            D = s[1] * s[2] * s[3]
            covs[n] = torch.from_numpy(I.covariance_like(D, 40 + i, rows_mult=2)).to(dev)
    results = {}
    for low in (False, True):
        params = [torch.nn.Parameter(init[n].clone().to(dev)) for n in shapes]
        opt = (N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True) if kind == "sgd"
               else N.AdamWNSCL(params, lr=1e-3, weight_decay=0.05, svd=True))
        opt.param_groups[0]["names"] = list(shapes)
        opt.low_rank = low
        opt.get_eigens(covs)
        opt.get_transforms(offset=0.0)
        g2 = torch.Generator().manual_seed(9)
        for step in range(3):
            for p in params:
                p.grad = torch.randn(p.shape, generator=g2).to(dev)
            opt.step()
        torch.cuda.synchronize()
        n_lr, lr_flops, t1, t2 = opt.lowrank_stats()
        results[low] = ([p.detach().cpu() for p in params], n_lr, {n: opt._basis[n]["rank"] for n in opt._basis})
    (dense, n0, none), (lowr, n1, ranks) = results[False], results[True]
    assert n0 == 0 and none == {}
    qualifies = [n for n, s in shapes.items() if len(s) == 4 and s[0] % 32 == 0 and (s[1] * s[2] * s[3]) % 32 == 0
                 and 0 < ranks.get(n, 0) <= 128]
    assert n1 == len(qualifies) >= 3, (n1, qualifies, ranks)
    for n, a, b in zip(shapes, dense, lowr):
        upd = (a.double() - init[n].double())
        allowed = REL * upd.abs().max().item() + 2 * 2.0 ** -23 * a.abs().max().item()
        assert (a.double() - b.double()).abs().max().item() <= allowed, (kind, n)


def test_set_basis_builds_the_projector(N, dev):
    D, r = 256, 24
    Q, _ = torch.linalg.qr(torch.randn(D, D, generator=torch.Generator().manual_seed(3)))
    p = torch.nn.Parameter(torch.zeros(128, D, 1, 1, device=dev))
    opt = N.SGDNSCL([p], lr=0.1, svd=True)
    opt.param_groups[0]["names"] = ["backbone.w.weight"]
    opt.set_basis("backbone.w.weight", Q.contiguous().to(dev), r)
    ref = Q[:, r:] @ Q[:, r:].t()
    ref = ref / ref.norm()
    assert _rel(opt.transforms["backbone.w.weight"], ref) <= REL
    assert abs(float(opt._basis["backbone.w.weight"]["norm"]) - (D - r) ** 0.5) <= 1e-3


def test_low_rank_form_at_full_layer_size(N, dev):
    """The largest R-50 layer shape (512 x 4608): eigh on the GPU -> head-form projector P = c (I - U U^T); the low-rank
    step against the dense GEMM with the SAME P (fp16-split kernel), ONE SGD step from p = 0, row by row.  (V_tail V_tail^T and
    I - U U^T are only equal for exactly orthonormal V: with P built from the tail of a raw 4608-wide fp32 eigenbasis the two
    forms of the step differed by 1.06e-5 of max|update| in round 1 -- which is why the head form is built from U itself.)"""
    rows, D = 512, 4608
    g = torch.Generator(device=dev).manual_seed(21)
    X = torch.randn(2 * D, D, device=dev, generator=g) * torch.logspace(0, -3, D, device=dev)
    C = (X.t() @ X).contiguous()
    grad = torch.randn(rows, D // 9, 3, 3, device=dev, generator=g)
    res = []
    for low in (False, True):
        p = torch.nn.Parameter(torch.zeros(rows, D // 9, 3, 3, device=dev))
        opt = N.SGDNSCL([p], lr=0.02, momentum=0.9, svd=True)
        opt.param_groups[0]["names"] = ["backbone.layer4.0.conv2.weight"]
        opt.low_rank = True            # both runs build P from the polished basis ...
        opt.get_eigens({"backbone.layer4.0.conv2.weight": C})
        opt.get_transforms()
        opt.low_rank = low             # ... and differ only in the form of the step
        p.grad = grad.clone()
        opt.step()
        torch.cuda.synchronize()
        res.append((p.detach().clone(), opt.lowrank_stats()[0], opt._basis["backbone.layer4.0.conv2.weight"]["rank"]))
    (dense, n0, r), (lowr, n1, _) = res
    assert n0 == 0 and n1 == 1 and 0 < r <= 128, (n0, n1, r)
    assert _rel(lowr, dense) <= REL and _row_rel(lowr, dense) <= REL, (_rel(lowr, dense), _row_rel(lowr, dense))


@pytest.mark.parametrize("D,first,norm", [(128, 20, True), (256, 33, False), (2304, 35, True), (512, 64, True), (1024, 100, False), (4608, 128, True), (160, 1, True),
                                          (1024, 129, True), (2304, 256, False), (4608, 200, True)])
def test_build_projector_head_vs_oracle(N, dev, D, first, norm):
    """``nsgp_build_projector_head``: I - U U^T from the removed directions against the oracle's V_tail V_tail^T of the same
    (orthonormal) basis; bit-symmetric; and the complement of U to fp32 rounding (P U = 0)."""
    from nsgp_repre_amd import ops
    Q, _ = torch.linalg.qr(torch.randn(D, D, generator=torch.Generator().manual_seed(D + first), dtype=torch.float64))
    V = Q.float()
    mask = torch.zeros(D, dtype=torch.bool)
    mask[first:] = True
    ref = (Q[:, first:] @ Q[:, first:].t())
    if norm:
        ref = ref / ref.norm()
    rpad = 32 if first <= 32 else (64 if first <= 64 else (128 if first <= 128 else 256))
    U = torch.zeros(D, rpad)
    U[:, :first] = V[:, :first]
    P, nrm = ops.build_projector_head(U.to(dev), norm, return_norm=True)
    assert _rel(P, ref) <= 2e-6
    assert torch.equal(P, P.t().contiguous())
    if norm:
        assert abs(float(nrm) - (D - first) ** 0.5) <= 1e-3 * (D - first) ** 0.5
    assert (P.double() @ U.double().to(dev)).abs().max().item() <= 1e-6 * P.abs().max().item()
    with pytest.raises(RuntimeError):
        ops.build_projector_head(torch.zeros(100, 32, device=dev), False)      # D % 32 != 0


@pytest.mark.parametrize("kind", list(I.G1B_KINDS))
def test_g1b_default_pipeline_from_covariance_per_row(N, dev, golden_dir, kind):
    """EIGENSOLVER-DISTANCE statement (not the kernel gate: that is test_g1c_default_low_rank_launches_on_the_references_basis_per_row,
    which holds the same launches to the reference's output at 1e-5 given the reference's own basis).  The DEFAULT path end to end against the reference's own output (G1b, 128-aligned layers): covariance -> ``get_eigens``
    (eigh on the GPU) -> elbow -> head-form projector -> low-rank step, next to the reference's literal formula on the same
    optimizer (``low_rank = False``: V_tail V_tail^T and the dense GEMM).  Three statements, first step from p = 0, row by row:
    (i) the two forms of the product agree under THE GATE; (ii) against what the reference's ``step()`` produced with ITS
    projector (LAPACK gesdd via torch.svd) both sit at the distance between two fp32 EIGENSOLVERS -- measured 2e-6 .. 1.6e-5 of a
    row's maximum (an AdamW first step is sign(g)-like: every entry of the row carries the same weight into the projector's
    1-4e-6 difference, test_eigensolver_distance_...) -- so the low-rank form must be no further from the reference than 1.5 x the
    dense form (or inside the gate), and (iii) inside 5e-5 absolutely."""
    g = np.load(os.path.join(golden_dir, f"g1b_{kind}.npz"))
    names, _ = I.g1b_layers()
    got = {}
    for low in (True, False):
        params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in I.g1b_params()]
        opt = _make_opt(N, kind, params)
        opt.param_groups[0]["names"] = list(names)
        opt.low_rank = low
        opt.get_eigens({n: torch.from_numpy(c).to(dev) for n, c in I.g1b_covariances().items()})
        opt.get_transforms(offset=I.G1_OFFSET)
        prev = [torch.from_numpy(a) for a in I.g1b_params()]
        for step in range(I.G1B_STEPS):
            for p, a in zip(params, I.g1b_grads(step)):
                p.grad = torch.from_numpy(a).to(dev)
            opt.step()
            torch.cuda.synchronize()
            if low:
                assert opt.lowrank_stats()[0] == len(I.g1b_projected()) and opt.tile_counts() == (0, 0, 0)
            else:
                assert opt.lowrank_stats()[0] == 0 and opt.tile_counts()[2] > 0
            if step == 0:
                got[low] = {n: p.detach().clone() for n, p in zip(names, params)}
            for n, p, p0 in zip(names, params, prev):
                ref = torch.from_numpy(g[f"p_step{step}__{_key(n)}"])
                upd_ref = ref.double() - p0.double()
                upd = p.detach().cpu().double() - p0.double()
                allowed = 5 * REL * upd_ref.abs().max().item() + 2 * 2.0 ** -23 * ref.abs().max().item()
                assert (upd - upd_ref).abs().max().item() <= allowed, (kind, low, n, step)
            prev = [torch.from_numpy(g[f"p_step{step}__{_key(n)}"]) for n in names]
        opt.close()
    for n in I.g1b_projected():
        ref = torch.from_numpy(g[f"p_step0__{_key(n)}"])
        d_low, d_dense = _row_rel(got[True][n], ref), _row_rel(got[False][n], ref)
        assert _row_rel(got[True][n], got[False][n]) <= REL, (kind, n, "the two forms of the product")           # (i)
        assert d_low <= max(REL, 1.5 * d_dense), (kind, n, d_low, d_dense)                                          # (ii)
        assert d_low <= 5e-5 and d_dense <= 5e-5, (kind, n, d_low, d_dense)                                         # (iii)


@pytest.mark.parametrize("polish", [True, False])
@pytest.mark.parametrize("kind", list(I.G1C_KINDS))
def test_g1c_default_low_rank_launches_on_the_references_basis_per_row(N, dev, golden_dir, kind, polish):
    """THE GATE on the DEFAULT step kernels against reference output, given the reference's own basis (kernel parity is
    judged given identical (U, r)).  G1c stores U = eigen_vector[:, :r] of the reference's torch.svd and what its ``step()`` --
    ``torch.mm(update, V_tail V_tail^T [/ ||.||_F])``, SGD_NSCL.py:82-94,270-285 -- made of p0 = 0 and gradient rows over six
    decades, on layers that reach every path of the low-rank launches: r = 21 / 23 / 24 (rpad 32, one K range), r = 48 at D = 1152
    (rpad 64, 5 K ranges: slabs + nsgp_lr_reduce_kernel), r = 99 at D = 2304 (rpad 128, 9 K ranges; backbone: Frobenius scale).
    ``set_basis(n, U_ref, r)`` installs exactly that basis; every projected layer must run on nsgp_update_lr_kernel /
    nsgp_lr_apply_kernel; every output row within 1e-5 of its own maximum, nothing added -- except that for the two backbone
    layers the reference's inexact fp32 CPU ``torch.norm`` (one scalar per layer, see below) is taken out first.
    SGD (momentum; Nesterov) and AdamW."""
    g = np.load(os.path.join(golden_dir, f"g1c_{kind}.npz"))
    gU = np.load(os.path.join(golden_dir, "g1c_sgd.npz"))
    names, _ = I.g1c_layers()
    params = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in I.g1c_params()]
    opt = _make_opt(N, kind, params)
    opt.param_groups[0]["names"] = list(names)
    opt.polish_basis = polish
    assert opt.low_rank is True
    for n in I.g1c_projected():
        r = int(g[f"rank__{_key(n)}"])
        U = torch.from_numpy(gU[f"U__{_key(n)}"]).to(dev)
        assert U.shape[1] == r
        opt.set_basis(n, U, r)
    worst, scales = {}, {}
    for step in range(I.G1C_STEPS[kind]):
        for p, a in zip(params, I.g1c_grads(step)):
            p.grad = torch.from_numpy(a).to(dev)
        opt.step()
        torch.cuda.synchronize()
        assert opt.lowrank_stats()[0] == len(I.g1c_projected()) == 5 and opt.tile_counts() == (0, 0, 0)
        for n, p in zip(names, params):
            if n not in I.g1c_projected():
                continue
            ref = torch.from_numpy(g[f"p_step{step}__{_key(n)}"])
            s = 1.0
            if "backbone" in n:
                # The Frobenius scale.  The reference divides by ITS fp32 `torch.norm(transform)` (SGD_NSCL.py:282-283); on the CPU
                # that reduction keeps one fp32 vector accumulator over all D^2 squares and drops the small off-diagonal ones: for
                # this D = 2304 projector it returns 46.9290 where ||P||_F = sqrt(D - r) = 46.9574 (fixture: refnorm / refnorm64),
                # 6.05e-4 low -- a uniform factor on the layer's update, which the reference's own GPU path (a tree reduction) does
                # not share.  The product uses the correctly rounded norm (fp64 accumulation, checked here against sqrt(D - r));
                # the row-wise gate is then held on the update with that ONE scalar per layer taken out.
                D, r = ref[0].numel(), int(g[f"rank__{_key(n)}"])
                ours, theirs = float(opt._basis[n]["norm"]), float(g[f"refnorm__{_key(n)}"])
                assert abs(ours / (D - r) ** 0.5 - 1.0) <= 2e-6, (n, ours, (D - r) ** 0.5)
                assert abs(float(g[f"refnorm64__{_key(n)}"]) / (D - r) ** 0.5 - 1.0) <= 2e-6
                s = ours / theirs
                scales[n] = s
                assert abs(s - 1.0) <= 1e-3, (n, s)
            if step == 0:       # p0 = 0: p IS the projected update -- the gate itself
                worst[n] = _row_rel(p.detach() * s, ref)
                assert worst[n] <= REL, (kind, n, worst[n])
            else:               # p != 0: the fp32 add of the update into p rounds at ulp(p) on both sides
                assert _rel(p.detach() * s, ref) <= REL, (kind, n, step)
    out_dir = os.environ.get("NSGP_REPORT_DIR")
    if out_dir:
        import json
        os.makedirs(out_dir, exist_ok=True)
        json.dump(dict(fixture=f"g1c_{kind}", polish_basis=polish, gate="max_n|ours-ref| <= 1e-5 * max_n|ref| per output row, p0 = 0",
                       worst_row_rel_vs_reference=worst, frobenius_norm_ours_over_reference_cpu_fp32=scales, ranks={n: int(g[f"rank__{_key(n)}"]) for n in I.g1c_projected()}),
                  open(os.path.join(out_dir, f"parity_g1c_{kind}_{'polished' if polish else 'raw'}.json"), "w"), indent=1)
    opt.close()


@pytest.mark.parametrize("depth", [50, 101])
def test_full_table_low_rank_default_vs_oracle_per_row(N, dev, depth):
    """The complete projected-layer tables on the DEFAULT path: projectors built by ``set_basis`` from SURVEY 8d's seeded
    covariances (eigh -> elbow -> head form), one SGDNSCL step, against the ORACLE's dense ``u @ P`` with the product's own
    ``transforms[name]`` on the CPU -- row by row under THE GATE.  Every layer whose rank is <= 128 must take the low-rank
    launches.  With NSGP_REPORT_DIR set the per-layer errors (also against an fp64 product) are written there."""
    import json
    from nsgp_repre_amd.optim.threshold import elbow_index
    layers = O.resnet_fpn_projected_layers(depth)
    basis = {}
    for n, cout, D in layers:
        if D in basis:
            continue
        gen = torch.Generator(device=dev).manual_seed(2000 + D)
        X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
        lam, Q = torch.linalg.eigh((X.t() @ X).contiguous())
        sv = lam.abs()
        order = torch.argsort(sv, descending=True, stable=True)
        basis[D] = (Q[:, order].contiguous(), int(elbow_index(sv[order].cpu().numpy(), 0.0, "sgd")))
        del X, lam, Q
    gen = torch.Generator(device="cpu").manual_seed(11 + depth)
    params, names, shapes = [], [], {}
    for n, cout, D in layers:
        k = 3 if ("conv2" in n or "fpn_convs" in n) else 1
        shapes[n] = (cout, D // (k * k), k, k)
        params.append(torch.nn.Parameter(torch.zeros(shapes[n], device=dev)))
        names.append(n)
    params.append(torch.nn.Parameter(torch.randn(1000, generator=gen).to(dev)))
    names.append("backbone.bn.weight")
    hp = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    opt = N.SGDNSCL(params, svd=True, **hp)
    opt.param_groups[0]["names"] = names
    assert opt.low_rank is True
    for n, cout, D in layers:
        opt.set_basis(n, *basis[D])
    grads = []
    for n, p in zip(names, params):
        gr = torch.randn(p.shape, generator=gen) * 1e-3
        if n in shapes:
            gr *= torch.pow(10.0, -3.0 * (torch.arange(p.shape[0]) % 16) / 15.0).view(-1, 1, 1, 1)
        grads.append(gr)
    cpu_params = [p.detach().cpu().clone() for p in params]
    for p, gr in zip(params, grads):
        p.grad = gr.clone().to(dev)
    opt.step()
    torch.cuda.synchronize()
    n_lr, lr_flops, t1, t2 = opt.lowrank_stats()
    want = [n for n, cout, D in layers if 0 < basis[D][1] <= 128]
    assert n_lr == len(want) >= len(layers) - 8, (n_lr, len(want), {D: b[1] for D, b in basis.items()})
    tr_cpu = {n: opt.transforms[n].cpu() for n in shapes}
    O.sgd_nscl_step(names, cpu_params, [g.clone() for g in grads], [dict() for _ in names], tr_cpu, **hp)
    report, worst = [], 0.0
    for n, p, q, gr in zip(names, params, cpu_params, grads):
        if n not in shapes:
            assert _rel(p, q) <= 1e-6, n
            continue
        vs_oracle = _row_rel(p, q)
        worst = max(worst, vs_oracle)
        u64 = (-(0.02 * gr.to(dev))).double().view(gr.shape[0], -1) @ opt.transforms[n].double()
        rec = dict(layer=n, rows=gr.shape[0], D=tr_cpu[n].shape[0], removed_directions=basis[tr_cpu[n].shape[0]][1],
                   low_rank=n in want, vs_oracle_row_rel=vs_oracle, vs_fp64_row_rel=_row_rel(p, u64),
                   oracle_vs_fp64_row_rel=_row_rel(q.to(dev), u64))
        report.append(rec)
        assert vs_oracle <= REL, rec
    out_dir = os.environ.get("NSGP_REPORT_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        summary = dict(table=f"R-{depth}-FPN", path="low_rank (default)", gate="max_n|ours-ref| <= 1e-5 * max_n|ref| per output row",
                       layers_on_low_rank=n_lr, worst_row_rel_vs_oracle=worst, worst_row_rel_vs_fp64=max(r["vs_fp64_row_rel"] for r in report),
                       oracle_worst_row_rel_vs_fp64=max(r["oracle_vs_fp64_row_rel"] for r in report), layers=report)
        json.dump(summary, open(os.path.join(out_dir, f"parity_r{depth}_low_rank.json"), "w"), indent=1)


def test_low_rank_takes_misaligned_gradient_views_and_every_rank_class(N, dev):
    """Gradients that are views into a flat bucket at 4-byte-aligned offsets (DDP's gradient_as_bucket_view) on the low-rank
    launches, for layers of each rank class (U padded to 32, 64, 128 columns, and the WIDE class of 129 .. 256 removed directions with its own
    pair of launches: one K range and five), SGD (Nesterov) / AdamW / AdamW-AMSGrad, against the oracle; a layer with 257 removed directions
    and one whose row count is not a 32-multiple take the dense GEMM with their head-form projector."""
    ranks = {"backbone.r20.weight": 20, "neck.r33.weight": 33, "backbone.r96.weight": 96, "neck.r128.weight": 128, "backbone.r129.weight": 129,
             "neck.r200.weight": 200, "backbone.r256.weight": 256, "backbone.r257.weight": 257, "neck.rows48.weight": 10}
    shapes = {"backbone.r20.weight": (64, 160), "neck.r33.weight": (96, 32, 3, 3), "backbone.r96.weight": (32, 512, 1, 1),
              "neck.r128.weight": (160, 384), "backbone.r129.weight": (128, 256), "neck.r200.weight": (96, 128, 3, 3),
              "backbone.r256.weight": (64, 2304), "backbone.r257.weight": (64, 512), "neck.rows48.weight": (48, 64), "x.bias": (7,)}
    for kind in ("sgd", "adamw", "adamw_amsgrad"):
        gen = torch.Generator().manual_seed(31)
        init = {n: torch.randn(s, generator=gen) * 0.05 for n, s in shapes.items()}
        params = {n: torch.nn.Parameter(init[n].clone().to(dev)) for n in shapes}
        opt = (N.SGDNSCL(list(params.values()), lr=0.02, momentum=0.9, weight_decay=1e-4, nesterov=True, svd=True) if kind == "sgd"
               else N.AdamWNSCL(list(params.values()), lr=1e-3, weight_decay=0.05, svd=True, amsgrad=kind.endswith("amsgrad")))
        opt.param_groups[0]["names"] = list(shapes)
        for n, r in ranks.items():
            D = int(np.prod(shapes[n][1:]))
            Q, _ = torch.linalg.qr(torch.randn(D, D, generator=gen))
            opt.set_basis(n, Q.contiguous().to(dev), r)
        tr_cpu = {n: opt.transforms[n].cpu() for n in ranks}
        cpu = {n: init[n].clone() for n in shapes}
        states = [dict() for _ in shapes]
        for step in range(2):
            grads = {n: torch.randn(s, generator=gen) for n, s in shapes.items()}
            flat = torch.zeros(sum(v.numel() + 1 for v in grads.values()) + 1, device=dev)
            off = 1
            for n in shapes:
                view = flat[off:off + grads[n].numel()].view(shapes[n])
                view.copy_(grads[n].to(dev))
                params[n].grad = view
                off += grads[n].numel() + 1
            opt.step()
            if kind == "sgd":
                O.sgd_nscl_step(list(shapes), [cpu[n] for n in shapes], [grads[n].clone() for n in shapes], states, tr_cpu,
                                lr=0.02, momentum=0.9, weight_decay=1e-4, nesterov=True)
            else:
                O.adamw_nscl_step(list(shapes), [cpu[n] for n in shapes], [grads[n].clone() for n in shapes], states, tr_cpu,
                                  lr=1e-3, weight_decay=0.05, amsgrad=kind.endswith("amsgrad"))
        torch.cuda.synchronize()
        assert opt.lowrank_stats()[0] == 7, opt.lowrank_stats()          # r = 257 and the 48-row layer (head-form P, not a 32-multiple) take the dense GEMM
        _check(params, cpu, init, kind)


# ------------------------------------------------------------------ param groups, odd paths
def _run_both(N, dev, groups_spec, shapes, transforms_cpu, steps=2, grad_views=None, kind="sgd", split_mfma=None):
    """Run the HIP optimizer and the oracle on the same tensors; return (params_gpu, params_cpu)."""
    gen = torch.Generator().manual_seed(17)
    init = {n: torch.randn(s, generator=gen) * 0.05 for n, s in shapes.items()}
    params = {n: torch.nn.Parameter(init[n].clone().to(dev)) for n in shapes}
    groups = [dict(params=[params[n] for n in names], **hp) for names, hp in groups_spec]
    opt = {"sgd": lambda: N.SGDNSCL(groups, lr=0.01, svd=True), "adamw": lambda: N.AdamWNSCL(groups, lr=1e-3, svd=True),
           "adam": lambda: N.AdamNSCL(groups, lr=1e-3, svd=True)}[kind]()
    for g, (names, _) in zip(opt.param_groups, groups_spec):
        g["names"] = list(names)
    for n, P in transforms_cpu.items():
        opt.transforms[n] = P.to(dev)
    if split_mfma is not None:
        opt.split_mfma = split_mfma
    cpu = {n: init[n].clone() for n in shapes}
    states = {n: dict() for n in shapes}
    for step in range(steps):
        grads = {n: torch.randn(s, generator=gen) for n, s in shapes.items()}
        if grad_views is None:
            for n in shapes:
                params[n].grad = grads[n].clone().to(dev)
        else:   # all grads are views into ONE flat buffer at deliberately odd (4-byte aligned only) offsets
            total = sum(v.numel() + 1 for v in grads.values()) + 1
            flat = torch.zeros(total, device=dev)
            off = 1
            for n in shapes:
                view = flat[off:off + grads[n].numel()].view(shapes[n])
                view.copy_(grads[n].to(dev))
                params[n].grad = view
                off += grads[n].numel() + 1
        opt.step()
        for (names, hp), g in zip(groups_spec, opt.param_groups):
            full = {k: v for k, v in g.items() if k in ("lr", "momentum", "dampening", "weight_decay", "nesterov")}
            if kind == "sgd":
                O.sgd_nscl_step(list(names), [cpu[n] for n in names], [grads[n].clone() for n in names],
                                [states[n] for n in names], transforms_cpu, **full)
            else:
                (O.adamw_nscl_step if kind == "adamw" else O.adam_nscl_step)(
                    list(names), [cpu[n] for n in names], [grads[n].clone() for n in names], [states[n] for n in names], transforms_cpu,
                    lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], amsgrad=g["amsgrad"])
    torch.cuda.synchronize()
    return opt, params, cpu, init


def _check(params, cpu, init, tag):
    for n in cpu:
        upd = cpu[n].double() - init[n].double()
        allowed = REL * upd.abs().max().item() + 4 * 2.0 ** -23 * cpu[n].abs().max().item()
        assert (params[n].detach().cpu().double() - cpu[n].double()).abs().max().item() <= allowed, (tag, n)


def _proj_for(shapes, names):
    tr = {}
    for i, n in enumerate(names):
        s = shapes[n]
        D = int(np.prod(s[1:]))
        sv, V = O.eigens(torch.from_numpy(I.covariance_like(D, 70 + i, rows_mult=2)))
        tr[n] = O.build_projector(V, O.adaptive_threshold(sv, 0.0), "backbone" in n)
    return tr


# ------------------------------------------------------------------ two-term fp16 split of the projection GEMM
@pytest.mark.parametrize("split", ["f16x2"])
@pytest.mark.parametrize("kind", ["sgd", "sgd_nomomentum", "sgd_nesterov", "adamw", "adam_amsgrad"])
def test_split_mfma_steps_vs_oracle(N, dev, kind, split):
    """Both split projection paths against the CPU oracle under the same 1e-5 gate as the fp32 path: aligned layers run
    the split kernel, a ragged layer in the same plan stays on the generic fp32 tiles, a plain tensor is untouched.  The SGD
    flavours differ in WHICH array is the projection's A operand (momentum buffer / mutated gradient), i.e. in what the
    per-tensor fp16 scale is measured on."""
    shapes = {"backbone.a.weight": (128, 256), "backbone.b.weight": (256, 128, 3, 3), "neck.c.weight": (256, 256, 3, 3),
              "backbone.ragged.weight": (100, 36, 3, 3), "rpn_head.x.weight": (64, 40)}
    names = list(shapes)
    tr = _proj_for(shapes, names[:4])
    hp = {"sgd": dict(lr=0.02, momentum=0.9, weight_decay=1e-4), "sgd_nomomentum": dict(lr=0.02, momentum=0.0, weight_decay=1e-4),
          "sgd_nesterov": dict(lr=0.02, momentum=0.9, weight_decay=1e-4, nesterov=True), "adamw": dict(lr=1e-3, weight_decay=0.05),
          "adam_amsgrad": dict(lr=1e-3, weight_decay=1e-4, amsgrad=True)}[kind]
    okind = {"adamw": "adamw", "adam_amsgrad": "adam"}.get(kind, "sgd")
    opt, params, cpu, init = _run_both(N, dev, [(names, hp)], shapes, tr, steps=3, kind=okind, split_mfma=split)
    assert opt.uses_split_mfma() == split
    _check(params, cpu, init, f"{split}-{kind}")
    # the same run on the fp32 MFMA path lands within the same gate of the split run
    opt2, params2, cpu2, init2 = _run_both(N, dev, [(names, hp)], shapes, tr, steps=3, kind=okind, split_mfma=False)
    assert not opt2.uses_split_mfma()
    for n in shapes:
        upd = (cpu[n] - init[n]).abs().max().item()
        assert (params[n] - params2[n]).abs().max().item() <= REL * upd + 4 * 2.0 ** -23 * cpu[n].abs().max().item(), n


@pytest.mark.parametrize("split", ["f16x2"])
def test_split_mfma_with_misaligned_gradients_and_projector_edits(N, dev, split):
    """(i) gradient views at odd offsets: the bf16 kernel's tile falls back to the guarded fp32 loader for that operand, the
    fp16 path reads them with scalar loads in the elementwise launch that writes the split copy;
    (ii) an in-place edit of a projector invalidates its cached split (tensor version in the plan key)."""
    shapes = {"backbone.a.weight": (128, 256), "neck.c.weight": (256, 128, 3, 3)}
    names = list(shapes)
    tr = _proj_for(shapes, names)
    hp = dict(lr=0.05, momentum=0.0, weight_decay=0.0)
    opt, params, cpu, init = _run_both(N, dev, [(names, hp)], shapes, tr, steps=2, grad_views=True, split_mfma=split)
    assert opt.uses_split_mfma() == split
    _check(params, cpu, init, "split-misaligned")
    n = names[0]
    before = params[n].detach().clone()
    opt.transforms[n].mul_(0.5)                                  # in place: same storage, new version
    g = torch.randn(shapes[n], generator=torch.Generator().manual_seed(9))
    for m in names:
        params[m].grad = (g if m == n else torch.zeros(shapes[m])).to(dev)
    opt.step()
    want = before.cpu() + O.project_update(-(0.05 * g), 0.5 * tr[n])
    assert _rel(params[n].detach() - before, want - before.cpu()) <= REL


def test_f16x2_split_survives_a_wide_dynamic_range(N, dev):
    """fp16 has a 5-bit exponent: the two-term path scales every ROW of the update and every COLUMN of the projector by its own
    power of two.  Updates 1e+4 and 1e-6 times the usual size, rows 1e-6 times smaller than their neighbours, projector columns
    over six decades, a tiny projector and an all-zero gradient: every output row must meet THE GATE against ITS OWN maximum
    (round 1 judged the small rows against the large rows' maximum and could not see a per-tensor scale), and no inf / nan
    may appear."""
    from nsgp_repre_amd import ops
    D, rows = 256, 256
    g = torch.Generator().manual_seed(21)
    sv, V = O.eigens(torch.from_numpy(I.covariance_like(D, 33, rows_mult=2)))
    P = O.build_projector(V, O.adaptive_threshold(sv, 0.0), True)
    col_decades = torch.pow(10.0, -6.0 * (torch.arange(D) % 24) / 23.0)
    for gscale, row_scale, pscale, cols in ((1e4, 1.0, 1.0, False), (1e-6, 1.0, 1.0, False), (1.0, 1e-6, 1.0, False),
                                            (1.0, 1.0, 1e-7, False), (1.0, 1e-6, 1.0, True), (0.0, 1.0, 1.0, False)):
        grad = torch.randn(rows, D, generator=g) * gscale
        grad[::2] *= row_scale                                   # every other row: neighbours inside one MFMA block differ
        Pc = P * pscale * (col_decades[None, :] if cols else 1.0)
        p = torch.nn.Parameter(torch.zeros(rows, D, device=dev))
        opt = N.SGDNSCL([p], lr=1.0, momentum=0.0, svd=True)
        opt.param_groups[0]["names"] = ["backbone.w.weight"]
        opt.transforms["backbone.w.weight"] = Pc.to(dev)
        opt.split_mfma = "f16x2"
        p.grad = grad.clone().to(dev)
        opt.step()
        assert opt.uses_split_mfma() == "f16x2" and opt.tile_counts() == (0, 0, 2)
        got = p.detach().cpu()
        assert torch.isfinite(got).all()
        want = (-(grad.double()) @ Pc.double())
        assert _row_rel(got, want) <= REL, (gscale, row_scale, pscale, cols, _row_rel(got, want))
        if cols:    # and column by column: a column-scaled projector keeps fp32-level accuracy in its small columns
            assert _row_rel(got.t().contiguous(), want.t().contiguous()) <= REL
    buf = ops.split_projector_f16(P.to(dev))
    terms, c, cinv = ops.unpack_split_f16(buf, D)
    colmax = P.abs().amax(0).to(dev)
    assert bool(((colmax * c >= 2.0 ** 13) & (colmax * c < 2.0 ** 14)).all()) and bool((c * cinv == 1).all())
    back = (terms[0].double() + terms[1].double()) * cinv.double()[:, None]          # [n][k] = P^T
    assert (back.t() - P.to(dev).double()).abs().max().item() <= 2.0 ** -21 * float(P.abs().max())


def test_param_groups_with_different_hyperparameters(N, dev):
    shapes = {"backbone.a.weight": (128, 128, 1, 1), "backbone.a.bias": (128,), "neck.b.weight": (40, 12, 3, 3),
              "neck.b.bias": (40,), "roi_head.fc.weight": (24, 33), "backbone.bn.weight": (128,)}
    tr = _proj_for(shapes, ["backbone.a.weight", "neck.b.weight", "roi_head.fc.weight"])
    spec = [(["backbone.a.weight", "backbone.a.bias"], dict(lr=0.002, momentum=0.9, weight_decay=1e-4)),
            (["neck.b.weight", "neck.b.bias", "roi_head.fc.weight"], dict(lr=0.02, momentum=0.8, dampening=0.2, weight_decay=0.0, nesterov=True)),
            (["backbone.bn.weight"], dict(lr=0.02, momentum=0.0, weight_decay=0.0))]
    opt, params, cpu, init = _run_both(N, dev, spec, shapes, tr, steps=3)
    _check(params, cpu, init, "groups")
    assert len(opt._plans) == 1 and len(opt._plans[0]["groups"]) == 3


def test_momentum_zero_and_weight_decay_uses_the_gradient_as_source(N, dev):
    """momentum = 0: the update source of the projection GEMM is the (weight-decayed) gradient itself."""
    shapes = {"backbone.a.weight": (128, 256, 1, 1), "neck.b.weight": (20, 12, 3, 3), "x.bias": (7,)}
    tr = _proj_for(shapes, ["backbone.a.weight", "neck.b.weight"])
    spec = [(list(shapes), dict(lr=0.05, momentum=0.0, weight_decay=0.01))]
    opt, params, cpu, init = _run_both(N, dev, spec, shapes, tr, steps=2)
    _check(params, cpu, init, "momentum0")
    opt.mutate_grad = False          # the decayed gradient must still reach the GEMM
    opt2, params2, cpu2, init2 = _run_both(N, dev, spec, shapes, tr, steps=2)
    _check(params2, cpu2, init2, "momentum0-nomutate")


def test_misaligned_gradient_views(N, dev):
    """Gradients that are 4-byte-aligned views of a flat bucket take the guarded loader (A operand) and the
    scalar elementwise path; results must not change."""
    shapes = {"backbone.a.weight": (128, 128, 1, 1), "backbone.c.weight": (256, 128, 1, 1), "neck.b.weight": (20, 12, 3, 3), "x.bias": (7,)}
    tr = _proj_for(shapes, ["backbone.a.weight", "backbone.c.weight", "neck.b.weight"])
    for hp in (dict(lr=0.05, momentum=0.0, weight_decay=0.01), dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3),
               dict(lr=0.05, momentum=0.9, weight_decay=1e-3)):
        opt, params, cpu, init = _run_both(N, dev, [(list(shapes), hp)], shapes, tr, steps=2, grad_views=True)
        _check(params, cpu, init, f"misaligned {hp}")


def test_more_param_groups_than_one_plan_holds(N, dev):
    """mmengine's paramwise constructor makes one group per parameter: 40 groups -> two plans of <= 32."""
    shapes = {f"backbone.l{i}.weight": (8, 16, 1, 1) for i in range(40)}
    tr = _proj_for(shapes, list(shapes)[:5])
    spec = [([n], dict(lr=0.01 * (1 + i % 3), momentum=0.9, weight_decay=1e-4 * (i % 2))) for i, n in enumerate(shapes)]
    opt, params, cpu, init = _run_both(N, dev, spec, shapes, tr, steps=2)
    _check(params, cpu, init, "40 groups")
    assert len(opt._plans) == 2


def test_adamw_param_groups(N, dev):
    shapes = {"backbone.a.weight": (128, 128, 1, 1), "neck.b.weight": (40, 12, 3, 3), "backbone.bn.weight": (128,)}
    tr = _proj_for(shapes, ["backbone.a.weight", "neck.b.weight"])
    spec = [(["backbone.a.weight"], dict(lr=1e-3, weight_decay=0.1)),
            (["neck.b.weight", "backbone.bn.weight"], dict(lr=1e-4, weight_decay=0.0, betas=(0.8, 0.99), amsgrad=True))]
    opt, params, cpu, init = _run_both(N, dev, spec, shapes, tr, steps=3, kind="adamw")
    _check(params, cpu, init, "adamw groups")


# ------------------------------------------------------------------ the other BASELINE configs as parity cases
def test_coco_40_40_sized_bank_vs_oracle(N, dev):
    """configs[3] (COCO 40+40 task 2): 40 old classes -> K <= 400 prototypes; ragged class sizes incl. a
    class with 2 rows.  Masks and labels bit-exact vs the oracle, bank within 1e-5."""
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    D = 1024
    feats, cls = [], []
    for c in range(40):
        n = 2 if c == 7 else 20 + (c * 7) % 50
        feats.append(I.class_rois(n, D, 9000 + c, n_clusters=3 + c % 4))
        cls.append(np.full(n, c, dtype=np.int64))
    feats, cls = torch.from_numpy(np.concatenate(feats)), torch.from_numpy(np.concatenate(cls))
    bank_ref, lab_ref, masks_ref, _ = O.build_bank(feats, cls, [0, 40, 80], 2, 10)
    bank, lab, masks, _ = build_prototype_bank(feats.to(dev), cls.to(dev), [0, 40, 80], 2, 10)
    # a similarity within 1e-5 of the threshold may legitimately flip: skip such classes (none expected)
    assert torch.equal(lab.cpu(), lab_ref) and bank.shape[0] <= 400
    for c in range(40):
        assert len(masks[c]) == len(masks_ref[c])
        for a, b in zip(masks[c], masks_ref[c]):
            assert torch.equal(a, b), c
    assert _rel(bank, bank_ref) <= REL


def test_coco_scale_class_3000_rois_vs_oracle(N, dev):
    """configs[3]/[4]: one COCO-sized class at the true feature width, N_c = 3000 x 12544 (226 GFLOP of similarity): counts and
    the bit matrix against the oracle (bit-exact wherever no similarity sits within 1e-5 of the threshold, checked in fp64), the
    greedy cover's masks / centres and the prototypes themselves."""
    from nsgp_repre_amd import ops
    from nsgp_repre_amd.roi_heads.prototype_bank import select_class_prototypes
    n, d = 3000, 12544
    Fc = torch.from_numpy(I.class_rois(n, d, 77, n_clusters=5))
    Fd = Fc.to(dev)
    n64 = Fd.double() / Fd.double().norm(dim=-1, keepdim=True)
    sim64 = n64 @ n64.t()
    ambiguous = ((sim64 - 0.6).abs() < 1e-5).cpu()
    counts, bitmask = ops.sim_counts(Fd, 0.6)
    words = bitmask.cpu().numpy().view(np.uint64)
    got = torch.from_numpy(np.unpackbits(words.view(np.uint8).reshape(n, -1), axis=1, bitorder="little")[:, :n].astype(bool))
    nrm = Fc / Fc.norm(dim=-1, keepdim=True)
    ref = (nrm @ nrm.t()) >= 0.6                                   # the oracle's arithmetic (torch CPU fp32)
    assert torch.equal(got | ambiguous, ref | ambiguous)
    assert torch.equal(got, got.t())
    assert torch.equal(counts.cpu(), got.long().sum(-1))           # counts = popcounts of the rows
    if not bool(ambiguous.any()):
        co, fine, masks, cen, _ = O.prototype_select(Fc, 10)
        co2, fine2, masks2, cen2 = select_class_prototypes(Fd, 10)
        assert cen == cen2 and len(masks) == len(masks2) and all(torch.equal(a, b) for a, b in zip(masks, masks2))
        assert _rel(co2, co) <= REL and all(_rel(a, b) <= REL for a, b in zip(fine2, fine))


def test_coco_stress_class_20000_rois_properties(N, dev):
    """configs[4]'s "large prototype bank": N_c = 20,000 x 12544 (1 GB of features, 10 TFLOP of similarity, a 50 MB bit
    matrix instead of the reference's 1.6 GB fp32 + 3.2 GB int64).  No CPU oracle at this size: size-independent properties --
    the bit matrix is symmetric with a set diagonal, counts are the row popcounts, sampled entries agree with an fp64
    similarity, every fine prototype is the mean of the rows its mask selects, and the build is bitwise repeatable."""
    from nsgp_repre_amd import ops
    from nsgp_repre_amd.roi_heads.prototype_bank import select_class_prototypes
    n, d = 20000, 12544
    g = torch.Generator(device=dev).manual_seed(3)
    centres = torch.relu(torch.randn(6, d, device=dev, generator=g))
    which = torch.randint(0, 6, (n,), device=dev, generator=g)
    F = torch.empty(n, d, device=dev)
    for lo in range(0, n, 2500):
        F[lo:lo + 2500] = torch.relu(centres[which[lo:lo + 2500]] + 0.6 * torch.randn(2500, d, device=dev, generator=g))
    counts, bitmask = ops.sim_counts(F, 0.6)
    words = bitmask.cpu().numpy().view(np.uint64)
    bits = np.unpackbits(words.view(np.uint8).reshape(n, -1), axis=1, bitorder="little")[:, :n].astype(bool)
    assert np.array_equal(bits, bits.T) and bits.diagonal().all()
    assert np.array_equal(counts.cpu().numpy(), bits.sum(1))
    rs = np.random.default_rng(0)
    ii, jj = rs.integers(0, n, 4000), rs.integers(0, n, 4000)
    Fi, Fj = F[torch.from_numpy(ii).to(dev)].double(), F[torch.from_numpy(jj).to(dev)].double()
    s64 = ((Fi * Fj).sum(1) / (Fi.norm(dim=1) * Fj.norm(dim=1))).cpu().numpy()
    clear = np.abs(s64 - 0.6) > 1e-5
    assert np.array_equal(bits[ii, jj][clear], (s64 >= 0.6)[clear])
    co, fine, masks, cen = select_class_prototypes(F, 10)
    assert 1 <= len(fine) <= 9 and len(set(cen)) == len(cen)
    assert _rel(co, F.double().mean(0, keepdim=True)) <= REL
    for f, m in zip(fine, masks):
        assert _rel(f, F[m.to(dev)].double().mean(0, keepdim=True)) <= REL
    co2, fine2, masks2, cen2 = select_class_prototypes(F, 10)
    assert cen2 == cen and torch.equal(co2, co) and all(torch.equal(a, b) for a, b in zip(fine2, fine))


def test_step_is_bitwise_deterministic(N, dev):
    """No float atomics anywhere on the path: two runs from the same state give the same bits
    (dense and low-rank form, covariance, prototype selection)."""
    from nsgp_repre_amd import ops
    shapes = {"backbone.a.weight": (256, 128, 3, 3), "neck.b.weight": (128, 512, 1, 1), "backbone.bn.weight": (300,)}
    cov = {n: torch.from_numpy(I.covariance_like(int(np.prod(s[1:])), 90 + i, rows_mult=2)).to(dev)
           for i, (n, s) in enumerate(shapes.items()) if len(s) == 4}
    for low in (False, True):
        outs = []
        for rep in range(2):
            gen = torch.Generator().manual_seed(3)
            params = [torch.nn.Parameter((torch.randn(s, generator=gen) * 0.02).to(dev)) for s in shapes.values()]
            opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
            opt.param_groups[0]["names"] = list(shapes)
            opt.low_rank = low
            opt.get_eigens(cov)
            opt.get_transforms()
            for step in range(3):
                for p in params:
                    p.grad = torch.randn(p.shape, generator=gen).to(dev)
                opt.step()
            torch.cuda.synchronize()
            outs.append([p.detach().clone() for p in params] + [opt.transforms[n].clone() for n in cov])
        for a, b in zip(*outs):
            assert torch.equal(a, b), low
    x = torch.randn(2, 64, 40, 56, generator=torch.Generator().manual_seed(1)).to(dev)
    c1 = ops.cov_accumulate_conv2d(x, (3, 3), (1, 1), (1, 1))
    c2 = ops.cov_accumulate_conv2d(x, (3, 3), (1, 1), (1, 1))
    assert torch.equal(c1, c2)
    F_ = torch.from_numpy(I.class_rois(200, 512, 8, n_clusters=5)).to(dev)
    (k1, b1), (k2, b2) = ops.sim_counts(F_), ops.sim_counts(F_)
    assert torch.equal(k1, k2) and torch.equal(b1, b2)


def test_state_dict_round_trip_keeps_stepping_correctly(N, dev):
    """Optimizer.load_state_dict replaces every state tensor; the step plans must follow (a stale
    momentum-buffer pointer would silently train on freed memory)."""
    shapes = {"backbone.a.weight": (128, 128, 1, 1), "x.bias": (9,)}
    tr = _proj_for(shapes, ["backbone.a.weight"])
    hp = dict(lr=0.05, momentum=0.9, weight_decay=1e-3)
    gen = torch.Generator().manual_seed(2)
    init = {n: torch.randn(s, generator=gen) * 0.05 for n, s in shapes.items()}
    grads = [{n: torch.randn(s, generator=gen) for n, s in shapes.items()} for _ in range(4)]

    def make():
        params = {n: torch.nn.Parameter(init[n].clone().to(dev)) for n in shapes}
        opt = N.SGDNSCL(list(params.values()), svd=True, **hp)
        opt.param_groups[0]["names"] = list(shapes)
        for n, P in tr.items():
            opt.transforms[n] = P.to(dev)
        return params, opt

    def run(params, opt, steps):
        for g in steps:
            for n in shapes:
                params[n].grad = g[n].clone().to(dev)
            opt.step()
    pa, oa = make()
    run(pa, oa, grads)                        # 4 uninterrupted steps
    pb, ob = make()
    run(pb, ob, grads[:2])
    import copy
    sd = copy.deepcopy(ob.state_dict())       # state_dict() hands out the live tensors: copy, or oc would share ob's buffers
    pc, oc = make()
    with torch.no_grad():
        for n in shapes:
            pc[n].copy_(pb[n])
    oc.load_state_dict(sd)                    # fresh optimizer, state restored
    oc.param_groups[0]["names"] = list(shapes)
    run(pc, oc, grads[2:])
    ob.load_state_dict(ob.state_dict())       # and a self round trip on a live optimizer (plans must be rebuilt)
    run(pb, ob, grads[2:])
    torch.cuda.synchronize()
    for n in shapes:
        assert torch.equal(pa[n], pc[n]) and torch.equal(pa[n], pb[n]), n


# ------------------------------------------------------------------ projector caches keyed on identity, explicit teardown
@pytest.mark.parametrize("split", ["f16x2", False, "low_rank"])
def test_rebuilding_projectors_on_a_stepped_optimizer(N, dev, split):
    """Task t -> t+1 inside one process: ``get_eigens`` / ``get_transforms`` run again on an optimizer that has already
    stepped.  ``set_basis`` frees the old projector of a layer just before the next layer's is allocated, so with several
    layers of equal D the allocator hands old addresses to new projectors -- the split copies must follow the tensor
    objects, not the addresses.  The oracle steps with the product's own current projectors (copied to the CPU), which
    isolates the caching from the eigensolver."""
    shapes = {"backbone.a.weight": (128, 256), "backbone.b.weight": (128, 256), "neck.c.weight": (256, 256),
              "neck.d.weight": (128, 256, 1, 1), "backbone.e.weight": (256, 128, 1, 1), "neck.f.weight": (128, 128)}
    names = list(shapes)
    gen = torch.Generator().manual_seed(5)
    init = {n: torch.randn(s, generator=gen) * 0.05 for n, s in shapes.items()}
    params = {n: torch.nn.Parameter(init[n].clone().to(dev)) for n in names}
    opt = N.SGDNSCL([params[n] for n in names], lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    opt.param_groups[0]["names"] = names
    low = split == "low_rank"         # the default: head-form projectors applied as c (u - (u U) U^T); the bases must follow too
    opt.low_rank = low
    if not low:
        opt.split_mfma = split
    cpu = {n: init[n].clone() for n in names}
    states = [dict() for _ in names]
    hp = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    for task in range(3):
        fea_in = {n: torch.from_numpy(I.covariance_like(int(np.prod(shapes[n][1:])), 300 + 17 * task + i, rows_mult=2)).to(dev)
                  for i, n in enumerate(names)}
        opt.get_eigens(fea_in)
        opt.get_transforms(offset=0.0)
        tr_cpu = {n: opt.transforms[n].cpu() for n in names}
        for _ in range(2):
            grads = {n: torch.randn(shapes[n], generator=gen) for n in names}
            for n in names:
                params[n].grad = grads[n].clone().to(dev)
            opt.step()
            O.sgd_nscl_step(names, [cpu[n] for n in names], [grads[n].clone() for n in names], states, tr_cpu, **hp)
        if low:
            assert opt.lowrank_stats()[0] == len(names) and opt.tile_counts() == (0, 0, 0)
            assert all(opt._basis[n]["P"] is opt.transforms[n] for n in names)
        else:
            assert opt.uses_split_mfma() == split and opt.lowrank_stats()[0] == 0
        torch.cuda.synchronize()
        _check(params, cpu, init, f"task{task}")
    if split and not low:   # every derived copy belongs to the projector object that is installed NOW
        assert all(opt._splits[n]["P"] is opt.transforms[n] for n in names)
    opt.close()
    assert opt._plans == [] and opt._splits == {}
    # still usable after close(): the next step builds fresh plans
    for n in names:
        params[n].grad = torch.zeros(shapes[n], device=dev)
    opt.step()


def test_raw_pointer_rewrite_of_a_projector_is_seen(N, dev):
    """``ops.build_projector(out=P)`` writes P through its raw pointer; the wrapper bumps the tensor's version counter so
    the cached fp16 split is redone (the version of such a tensor used to stay 0 for ever)."""
    from nsgp_repre_amd import ops
    D, rows = 256, 128
    sv, V = O.eigens(torch.from_numpy(I.covariance_like(D, 41, rows_mult=2)))
    Vd = V.to(dev).contiguous()
    p = torch.nn.Parameter(torch.zeros(rows, D, device=dev))
    opt = N.SGDNSCL([p], lr=1.0, momentum=0.0, svd=True)
    opt.param_groups[0]["names"] = ["neck.w.weight"]
    P = ops.build_projector(Vd, 40, False)
    opt.transforms["neck.w.weight"] = P
    g = torch.randn(rows, D, generator=torch.Generator().manual_seed(2))
    p.grad = g.clone().to(dev)
    opt.step()
    v0 = P._version
    ops.build_projector(Vd, 200, False, out=P)              # same tensor, same address, new contents
    assert P._version > v0
    before = p.detach().clone()
    p.grad = g.clone().to(dev)
    opt.step()
    want = O.project_update(-g, P.cpu())
    assert _rel(p.detach() - before, want) <= REL


def test_collected_optimizers_release_their_plans_later(N, dev):
    """``__del__`` makes no HIP call: the handles of a garbage-collected optimizer are parked and destroyed (with checked
    return codes) at the next explicit entry point."""
    import gc
    from nsgp_repre_amd.optim import base
    base.release_collected_plans()
    p = torch.nn.Parameter(torch.zeros(64, 32, device=dev))
    opt = N.SGDNSCL([p], lr=0.1, svd=True)
    opt.param_groups[0]["names"] = ["x.weight"]
    p.grad = torch.ones_like(p)
    opt.step()
    assert len(opt._plans) == 1
    del opt
    gc.collect()
    assert len(base._GRAVEYARD) == 1
    assert base.release_collected_plans() == 1 and base._GRAVEYARD == []
