"""Thin torch-tensor wrappers over the C ABI (device pointers + the current HIP stream).

torch is plumbing here: device memory, streams.  All arithmetic runs in the HIP
kernels of ``csrc/``.  Every wrapper raises on CPU tensors -- no fallback.
"""
import ctypes as C

import torch

from . import _lib


def _stream():
    # the raw handle of torch's current stream (torch.cuda.current_stream().cuda_stream builds a Stream object first: ~10x the cost,
    # and the replay pass is host-bound)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _touched(t: torch.Tensor) -> torch.Tensor:
    """A kernel wrote ``t`` through its raw pointer: bump the tensor's version counter, as an in-place torch op would, so
    that whoever caches something derived from it (the optimizers' split projectors) sees the edit."""
    torch.autograd.graph.increment_version(t)
    return t


def _dev(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"nsgp_repre_amd: `{name}` must be a GPU tensor (there is no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"nsgp_repre_amd: `{name}` must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"nsgp_repre_amd: `{name}` must be contiguous")
    return C.c_void_p(t.data_ptr())


def project(a: torch.Tensor, proj: torch.Tensor, scale: float = 1.0, out: torch.Tensor = None,
            accumulate: bool = False) -> torch.Tensor:
    """``out (+)= scale * (a.view(rows, -1) @ proj)`` -- the `torch.mm(update.view(Cout,-1), P)` of
    mmdet/engine/optimizers/SGD_NSCL.py:85-90 in isolation."""
    lib = _lib.load_library()
    rows = a.shape[0]
    cols = a.numel() // rows
    if proj.shape != (cols, cols):
        raise ValueError(f"projector shape {tuple(proj.shape)} does not match update [{rows} x {cols}]")
    fresh = out is None
    if fresh:
        out = torch.empty_like(a)
    _lib.check(lib.nsgp_project(_dev(a, "a"), _dev(proj, "proj"), _dev(out, "out"), rows, cols, float(scale),
                                int(accumulate), _stream()), "nsgp_project")
    return out if fresh else _touched(out)


def split_projector_f16(P: torch.Tensor) -> torch.Tensor:
    """Pre-tiled two-term fp16 split of ``diag(c) P^T`` with one power-of-two scale ``c[n]`` per projector column (largest
    |entry| of the column into [2^13, 2^14); fp16 overflows at 65504), as one uint8 buffer:
    ``[n/64][k/8][term][n%64][8 fp16]`` (4 bytes per element of P), then ``c`` and ``1/c`` as fp32 ``[D]`` each.  Two launches
    per projector per task; ``unpack_split_f16`` turns the buffer back into tensors for inspection."""
    lib = _lib.load_library()
    D = P.shape[0]
    if P.shape != (D, D):
        raise ValueError("projector must be square")
    if D % 64 != 0:
        raise ValueError("the fp16 split needs D to be a multiple of 64")
    out = torch.empty(lib.nsgp_split_projector_f16_bytes(D), dtype=torch.uint8, device=P.device)
    _lib.check(lib.nsgp_split_projector_f16(_dev(P, "P"), D, C.c_void_p(out.data_ptr()), _stream()), "nsgp_split_projector_f16")
    return out


def unpack_split_f16(buf: torch.Tensor, D: int):
    """(terms [2 x D x D] fp16 as [term][n][k], c [D], 1/c [D]) from the buffer ``split_projector_f16`` returns."""
    planes = buf[:D * D * 4].view(torch.float16).view(D // 64, D // 8, 2, 64, 8)       # [n/64][k/8][term][n%64][8]
    terms = planes.permute(2, 0, 3, 1, 4).reshape(2, D, D)
    scales = buf[D * D * 4:].view(torch.float32)
    return terms, scales[:D], scales[D:2 * D]


def build_projector(V: torch.Tensor, first_col: int, normalise: bool, out: torch.Tensor = None,
                    return_norm: bool = False):
    """``P = V[:, first_col:] @ V[:, first_col:].T`` (``/ ||P||_F`` if normalise) --
    mmdet/engine/optimizers/SGD_NSCL.py:270-285."""
    lib = _lib.load_library()
    D = V.shape[0]
    if V.shape != (D, D):
        raise ValueError("V must be square")
    fresh = out is None
    if fresh:
        out = torch.empty_like(V)
    nbytes = lib.nsgp_projector_scratch_bytes(D)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=V.device)
    _lib.check(lib.nsgp_build_projector(_dev(V, "V"), D, int(first_col), int(bool(normalise)), _dev(out, "P"),
                                        C.c_void_p(scratch.data_ptr()), nbytes, _stream()), "nsgp_build_projector")
    if not fresh:
        _touched(out)
    if return_norm:   # ||P||_F before the division (fp32, slot NORM_BLOCKS of the scratch), as a 0-d GPU tensor
        norm = scratch[1024 * 8:1024 * 8 + 4].view(torch.float32)[0] if normalise else torch.ones((), device=V.device)
        return out, norm
    return out


def build_projector_head(U: torch.Tensor, normalise: bool, out: torch.Tensor = None, return_norm: bool = False):
    """``P = I - U U^T`` (``/ ||P||_F`` if normalise) from the REMOVED directions ``U`` [D x rpad] (orthonormal columns,
    zero-padded to a multiple of 32 up to 128, or to 256) -- the projector of SGD_NSCL.py:270-285 written from the other side; the
    form the low-rank step applies (``nsgp_build_projector_head``)."""
    lib = _lib.load_library()
    D, rpad = U.shape
    fresh = out is None
    if fresh:
        out = torch.empty(D, D, dtype=torch.float32, device=U.device)
    nbytes = lib.nsgp_projector_scratch_bytes(D)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=U.device)
    _lib.check(lib.nsgp_build_projector_head(_dev(U, "U"), D, rpad, int(bool(normalise)), _dev(out, "P"),
                                             C.c_void_p(scratch.data_ptr()), nbytes, _stream()), "nsgp_build_projector_head")
    if not fresh:
        _touched(out)
    if return_norm:
        norm = scratch[1024 * 8:1024 * 8 + 4].view(torch.float32)[0] if normalise else torch.ones((), device=U.device)
        return out, norm
    return out


def cov_accumulate_conv2d(x: torch.Tensor, kernel_size, stride, padding, cov: torch.Tensor = None,
                          workspace: torch.Tensor = None) -> torch.Tensor:
    """``C (+)= X^T X`` with X the implicit unfold of the batch mean --
    mmdet/engine/runner/nsrunner_roi_replay.py:908-913, 930-934.  ``cov=None`` is the
    reference's first call (assign); passing ``cov`` accumulates in place."""
    lib = _lib.load_library()
    B, cin, H, W = x.shape
    kh, kw = kernel_size
    sh, sw = stride
    ph, pw = padding
    D = cin * kh * kw
    nbytes = lib.nsgp_cov_workspace_bytes(cin, H, W, kh, kw, sh, sw, ph, pw)
    if nbytes == 0:
        raise ValueError("bad convolution geometry")
    if workspace is None or workspace.numel() * workspace.element_size() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    accumulate = cov is not None
    if cov is None:
        cov = torch.empty(D, D, dtype=torch.float32, device=x.device)
    elif cov.shape != (D, D):
        raise ValueError(f"cov shape {tuple(cov.shape)} != ({D},{D})")
    _lib.check(lib.nsgp_cov_accumulate_conv2d(_dev(x, "x"), B, cin, H, W, kh, kw, sh, sw, ph, pw, _dev(cov, "cov"),
                                              int(accumulate), C.c_void_p(workspace.data_ptr()),
                                              workspace.numel() * workspace.element_size(), _stream()),
               "nsgp_cov_accumulate_conv2d")
    return _touched(cov) if accumulate else cov


def cov_set_split_mfma(mode: int) -> int:
    """Select the covariance SYRK's matrix-core path: 0 = fp32 MFMA, 1 = auto (default: the two-term fp16 split for layers
    large enough to repay its extra launches), 2 = always the split, 3 = always the split but only its first-generation
    (gather) kernel -- by default layers with D >= 512 take the second generation (materialised pre-tiled operand, LDS-DMA
    tiles, split-K).  Returns the previous setting."""
    return int(_lib.load_library().nsgp_cov_set_split_mfma(int(mode)))


def cov_set_corr_mode(mode: int) -> int:
    """Correlation form of the 3x3 / stride 1 / padding 1 layers in ``CovGroupPlan`` (read when a plan is created): 0 never,
    1 where it saves tile-steps (default), 2 wherever it applies.  Returns the previous mode."""
    return int(_lib.load_library().nsgp_cov_set_corr_mode(int(mode)))


def cov_workspace_bytes(cin, H, W, kernel_size, stride, padding) -> int:
    lib = _lib.load_library()
    return lib.nsgp_cov_workspace_bytes(cin, H, W, kernel_size[0], kernel_size[1], stride[0], stride[1],
                                        padding[0], padding[1])


class CovGroupPlan:
    """All hooked convolutions of ONE forward in a handful of launches (``nsgp_cov_plan_*``, csrc/covariance.hip "grouped pass";
    the 3x3 / stride 1 / padding 1 layers on large maps in the correlation form: ``n_correlation_form``, ``cov_set_corr_mode``).

    ``geoms``: one ``(batch, cin, h, w, kernel_size, stride, padding)`` per layer.  ``routes[i]`` says whether layer i rides in the
    grouped launches (True) or has to go through ``cov_accumulate_conv2d`` (False: D not a multiple of 64).  ``run(xs, covs)``
    assigns (``covs[i] is None``: a fresh [D x D] tensor is made) or accumulates in place, and returns the list of covariances
    (entries of route-False layers are passed through untouched).  ``close()`` releases the plan's device tables on the calling
    thread; nothing is released from a finaliser."""

    def __init__(self, geoms, device):
        lib = _lib.load_library()
        self.geoms = [tuple(g) for g in geoms]
        self.n = len(self.geoms)
        arr = (_lib.CovGeom * self.n)()
        self.D = []
        for a, (b, cin, h, w, k, s, p) in zip(arr, self.geoms):
            a.batch, a.cin, a.h, a.w, a.kh, a.kw, a.sh, a.sw, a.ph, a.pw = b, cin, h, w, k[0], k[1], s[0], s[1], p[0], p[1]
            self.D.append(cin * k[0] * k[1])
        self._handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.nsgp_cov_plan_create(C.byref(self._handle), arr, self.n), "nsgp_cov_plan_create")
        r = (C.c_int * self.n)()
        _lib.check(lib.nsgp_cov_plan_routes(self._handle, r, self.n), "nsgp_cov_plan_routes")
        self.routes = [bool(v) for v in r]
        self.workspace_bytes = int(lib.nsgp_cov_plan_workspace_bytes(self._handle))
        ng, nt, fl = C.c_int(), C.c_int(), C.c_double()
        _lib.check(lib.nsgp_cov_plan_stats(self._handle, C.byref(ng), C.byref(nt), C.byref(fl)), "nsgp_cov_plan_stats")
        self.n_grouped, self.n_tiles, self.upper_flops = ng.value, nt.value, fl.value
        nc, ts = C.c_int(), C.c_double()
        _lib.check(lib.nsgp_cov_plan_forms(self._handle, C.byref(nc), C.byref(ts)), "nsgp_cov_plan_forms")
        self.n_correlation_form, self.tile_steps = nc.value, ts.value
        self.device = device
        self._ws = None

    def run(self, xs, covs, workspace=None):
        """``workspace``: an optional uint8 GPU tensor of at least ``workspace_bytes`` (callers that keep several plans alive -- one
        per input geometry -- share one); by default the plan allocates and keeps its own."""
        if self._handle is None:
            raise RuntimeError("CovGroupPlan is closed")
        lib = _lib.load_library()
        if len(xs) != self.n or len(covs) != self.n:
            raise ValueError("one input and one covariance slot per layer of the plan")
        xp, cp, acc = (C.c_void_p * self.n)(), (C.c_void_p * self.n)(), (C.c_int * self.n)()
        out, seen = list(covs), set()
        for i, (x, c, g, ok) in enumerate(zip(xs, covs, self.geoms, self.routes)):
            if not ok:
                continue
            if tuple(x.shape) != (g[0], g[1], g[2], g[3]):
                raise ValueError(f"layer {i}: input shape {tuple(x.shape)} does not match the plan's {g[:4]}")
            xp[i] = _dev(x, "x").value
            acc[i] = int(c is not None)
            if c is None:
                c = out[i] = torch.empty(self.D[i], self.D[i], dtype=torch.float32, device=x.device)
            elif tuple(c.shape) != (self.D[i], self.D[i]):
                raise ValueError(f"layer {i}: cov shape {tuple(c.shape)} != ({self.D[i]},{self.D[i]})")
            if c.data_ptr() in seen:
                raise ValueError("the covariances of one grouped run must be distinct buffers")
            seen.add(c.data_ptr())
            cp[i] = _dev(c, "cov").value
        ws = workspace
        if ws is None:
            if self._ws is None or self._ws.numel() < self.workspace_bytes:
                self._ws = torch.empty(max(self.workspace_bytes, 16), dtype=torch.uint8, device=self.device)
            ws = self._ws
        elif not ws.is_cuda or ws.dtype != torch.uint8 or ws.numel() < self.workspace_bytes:
            raise ValueError(f"workspace must be a uint8 GPU tensor of at least {self.workspace_bytes} bytes")
        _lib.check(lib.nsgp_cov_plan_run(self._handle, xp, cp, acc, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "nsgp_cov_plan_run")
        for i, (c, ok) in enumerate(zip(covs, self.routes)):
            if ok and c is not None:
                _touched(c)
        return out

    def close(self):
        if self._handle is not None:
            h, self._handle = self._handle, None
            _lib.check(_lib.load_library().nsgp_cov_plan_destroy(h), "nsgp_cov_plan_destroy")
            self._ws = None


def cov_accumulate_linear(x: torch.Tensor, cov: torch.Tensor = None) -> torch.Tensor:
    """Linear branch, runner:901-902: ``X = mean(x, 0, keepdim)``; ``C (+)= X^T X``."""
    lib = _lib.load_library()
    if x.dim() != 2:
        raise ValueError("linear covariance expects a [B x F] input (torch.mm in the reference needs 2-D)")
    B, Fd = x.shape
    accumulate = cov is not None
    if cov is None:
        cov = torch.empty(Fd, Fd, dtype=torch.float32, device=x.device)
    _lib.check(lib.nsgp_cov_accumulate_linear(_dev(x, "x"), B, Fd, _dev(cov, "cov"), int(accumulate), _stream()),
               "nsgp_cov_accumulate_linear")
    return _touched(cov) if accumulate else cov


def sim_counts(feats: torch.Tensor, thr: float = 0.6):
    """Row-normalised Gram >= thr -> (counts int64 [N], bitmask uint64-as-int64 [N x words]) --
    mmdet/models/roi_heads/standard_roi_replay_head.py:417-423."""
    lib = _lib.load_library()
    N, D = feats.shape
    words = (N + 63) // 64
    norms = torch.empty(N, dtype=torch.float32, device=feats.device)
    counts = torch.empty(N, dtype=torch.int64, device=feats.device)
    bitmask = torch.empty(N, words, dtype=torch.int64, device=feats.device)
    nbytes = lib.repre_sim_workspace_bytes(N, D)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=feats.device)
    _lib.check(lib.repre_sim_counts(_dev(feats, "feats"), N, D, float(thr), _dev(norms, "norms"),
                                    _dev(counts, "counts", torch.int64), _dev(bitmask, "bitmask", torch.int64),
                                    C.c_void_p(ws.data_ptr()), nbytes, _stream()), "repre_sim_counts")
    return counts, bitmask


def masked_mean(feats: torch.Tensor, rowmask_words: torch.Tensor = None, n_selected: int = None) -> torch.Tensor:
    """Mean of the rows selected by a bit mask (int64 words) -> [1 x D] --
    standard_roi_replay_head.py:413 (all rows) and :443 (``mean(F[m])``)."""
    lib = _lib.load_library()
    N, D = feats.shape
    if rowmask_words is None:
        n_selected = N
        mptr = C.c_void_p(0)
    else:
        if n_selected is None:
            raise ValueError("n_selected is required with a mask")
        mptr = _dev(rowmask_words, "rowmask", torch.int64)
    out = torch.empty(1, D, dtype=torch.float32, device=feats.device)
    nbytes = lib.repre_masked_mean_workspace_bytes(N, D)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
    _lib.check(lib.repre_masked_mean(_dev(feats, "feats"), N, D, mptr, int(n_selected), _dev(out, "out"),
                                     C.c_void_p(ws.data_ptr()), nbytes, _stream()), "repre_masked_mean")
    return out


class _DoubleSoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, labels):
        lib = _lib.load_library()
        K, Cn = scores.shape
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        _lib.check(lib.repre_replay_ce_forward(_dev(scores, "scores"), _dev(labels, "labels", torch.int64), K, Cn,
                                               C.c_void_p(loss.data_ptr()), _stream()), "repre_replay_ce_forward")
        ctx.save_for_backward(scores, labels)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        scores, labels = ctx.saved_tensors
        lib = _lib.load_library()
        K, Cn = scores.shape
        go = grad_out.detach().reshape(()).float().contiguous()
        grad = torch.empty_like(scores)
        _lib.check(lib.repre_replay_ce_backward(_dev(scores, "scores"), _dev(labels, "labels", torch.int64), K, Cn,
                                                C.c_void_p(go.data_ptr()), _dev(grad, "grad"), _stream()), "repre_replay_ce_backward")
        return grad, None


def double_softmax_cross_entropy(scores: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``F.cross_entropy(scores.softmax(-1), labels)`` -- the replay classifier loss of
    standard_roi_replay_head.py:499 -- as one fused forward and one fused backward launch."""
    return _DoubleSoftmaxCE.apply(scores.float().contiguous(), labels.contiguous())


class _ReplayHeadLoss(torch.autograd.Function):
    """bank -> relu(fc) -> relu(fc) -> class rows -> CE(softmax(.)) in 6 launches, its backward in 5 (csrc/replay_head.hip).  The pass
    is host-bound (the GPU work is ~0.2 ms): one allocation per direction, raw pointers, the class heads read in place."""

    @staticmethod
    def forward(ctx, bank, labels, w1, b1, w2, b2, *heads):
        lib = _lib.load_library()
        K, fin = bank.shape
        hidden = w1.shape[0]
        hw, hb = heads[0::2], heads[1::2]
        nh = len(hw)
        rows = [int(w.shape[0]) for w in hw]
        C_ = sum(rows)
        if w1.shape != (hidden, fin) or w2.shape != (hidden, hidden) or nh == 0 or nh != len(hb) or any(w.shape[1] != hidden for w in hw):
            raise ValueError(f"replay head shapes do not chain: bank {tuple(bank.shape)}, w1 {tuple(w1.shape)}, w2 {tuple(w2.shape)}, "
                             f"class heads {[tuple(w.shape) for w in hw]}")
        nbytes = lib.repre_replay_head_workspace_bytes(K, fin, hidden, C_)
        if nbytes == 0 or nh > 16:
            raise ValueError(f"replay head: unsupported size (rows {K} <= 512, class columns {C_} <= 256, heads {nh} <= 16)")
        _dev(bank, "bank"), _dev(labels, "labels", torch.int64)
        for i, t in enumerate((w1, b1, w2, b2) + tuple(heads)):
            _dev(t, f"weight {i}")
        # one buffer: [workspace | h1 | h2 | scores | loss], every piece 256-byte aligned
        n_h, n_s = K * hidden, (K * C_ + 63) // 64 * 64
        ws_f = (nbytes + 255) // 256 * 64
        buf = torch.empty(ws_f + 2 * n_h + n_s + 64, dtype=torch.float32, device=bank.device)
        base = buf.data_ptr()
        scores = buf[ws_f + 2 * n_h:ws_f + 2 * n_h + K * C_].view(K, C_)
        loss = buf[ws_f + 2 * n_h + n_s]
        wp, bp, rp = (C.c_void_p * nh)(*[w.data_ptr() for w in hw]), (C.c_void_p * nh)(*[b.data_ptr() for b in hb]), (C.c_int * nh)(*rows)
        _lib.check(lib.repre_replay_head_forward(bank.data_ptr(), K, fin, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                 wp, bp, rp, nh, hidden, C_, labels.data_ptr(),
                                                 base + 4 * ws_f, base + 4 * (ws_f + n_h), scores.data_ptr(), loss.data_ptr(),
                                                 base, ws_f * 4, _stream()), "repre_replay_head_forward")
        ctx.save_for_backward(bank, labels, w2, buf, *heads)
        ctx.dims = (K, fin, hidden, C_, ws_f, n_h, n_s, rows)
        ctx.mark_non_differentiable(scores)
        return loss, scores

    @staticmethod
    def backward(ctx, grad_loss, _grad_scores):
        bank, labels, w2, buf, *heads = ctx.saved_tensors
        lib = _lib.load_library()
        K, fin, hidden, C_, ws_f, n_h, n_s, rows = ctx.dims
        hw, hb = heads[0::2], heads[1::2]
        nh = len(rows)
        go = grad_loss.detach().reshape(()).float().contiguous()
        # one buffer for all gradients, every piece 256-byte aligned
        sizes = [hidden * fin, hidden, hidden * hidden, hidden]
        for r in rows:
            sizes += [r * hidden, r]
        offs, tot = [], 0
        for n_ in sizes:
            offs.append(tot)
            tot += (n_ + 63) // 64 * 64
        gbuf = torch.empty(tot, dtype=torch.float32, device=bank.device)
        gp, base = gbuf.data_ptr(), buf.data_ptr()
        wp, bp, rp = (C.c_void_p * nh)(*[w.data_ptr() for w in hw]), (C.c_void_p * nh)(*[b.data_ptr() for b in hb]), (C.c_int * nh)(*rows)
        gwp = (C.c_void_p * nh)(*[gp + 4 * offs[4 + 2 * h] for h in range(nh)])
        gbp = (C.c_void_p * nh)(*[gp + 4 * offs[5 + 2 * h] for h in range(nh)])
        _lib.check(lib.repre_replay_head_backward(bank.data_ptr(), K, fin, w2.data_ptr(), wp, bp, rp, nh, hidden, C_, labels.data_ptr(),
                                                  base + 4 * ws_f, base + 4 * (ws_f + n_h), base + 4 * (ws_f + 2 * n_h), go.data_ptr(),
                                                  gp + 4 * offs[0], gp + 4 * offs[1], gp + 4 * offs[2], gp + 4 * offs[3], gwp, gbp,
                                                  base, ws_f * 4, _stream()), "repre_replay_head_backward")
        pieces = [gbuf[o:o + n_] for o, n_ in zip(offs, sizes)]
        grads = [pieces[0].view(hidden, fin), pieces[1], pieces[2].view(hidden, hidden), pieces[3]]
        for h, r in enumerate(rows):
            grads += [pieces[4 + 2 * h].view(r, hidden), pieces[5 + 2 * h]]
        return (None, None, *grads)


def replay_head_loss(bank: torch.Tensor, labels: torch.Tensor, w1, b1, w2, b2, head_weights, head_biases):
    """The per-step replay pass of standard_roi_replay_head.py:468-501 over the two shared FCs and the kept class rows of
    ``Shared2FCBBoxHeadTask`` (convfc_bbox_head_task.py:235-276), fused: ``(loss, scores)`` with
    ``scores = relu(relu(bank w1^T + b1) w2^T + b2) wc^T + bc`` and ``loss = F.cross_entropy(scores.softmax(-1), labels)``, where
    ``wc`` / ``bc`` are the rows of ``head_weights`` / ``head_biases`` (the per-task fc_cls heads seen so far, then the background
    head) read in place.  The bank is a constant (no gradient); gradients flow to every weight / bias.  fp32, GPU only."""
    flat = []
    for w, b in zip(head_weights, head_biases):
        flat += [w.float().contiguous(), b.float().contiguous()]
    return _ReplayHeadLoss.apply(bank.detach().float().contiguous(), labels.contiguous(), w1.float().contiguous(), b1.float().contiguous(),
                                 w2.float().contiguous(), b2.float().contiguous(), *flat)


def pseudo_label_filter(boxes: torch.Tensor, scores: torch.Tensor, gt_boxes: torch.Tensor, rpn_thresh: float,
                        roi_thresh: float, iou_thresh: float = 0.7):
    """Sequential teacher pseudo-label filter of one image -> (add_to_rpn bool[P], add_to_roi bool[P]) --
    mmdet/models/detectors/faster_rcnn_roi_replay.py:78-108."""
    lib = _lib.load_library()
    P, G = boxes.shape[0], gt_boxes.shape[0]
    add_rpn = torch.zeros(P, dtype=torch.uint8, device=boxes.device)
    add_roi = torch.zeros(P, dtype=torch.uint8, device=boxes.device)
    if P == 0:
        return add_rpn.bool(), add_roi.bool()
    gptr = _dev(gt_boxes, "gt_boxes") if G > 0 else C.c_void_p(0)
    _lib.check(lib.repre_pseudo_label_filter(_dev(boxes, "boxes"), _dev(scores, "scores"), P, gptr, G, float(iou_thresh),
                                             float(rpn_thresh), float(roi_thresh), _dev(add_rpn, "add_rpn", torch.uint8),
                                             _dev(add_roi, "add_roi", torch.uint8), _stream()), "repre_pseudo_label_filter")
    return add_rpn.bool(), add_roi.bool()


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_thr: float, idxs: torch.Tensor = None, max_keep: int = None):
    """Greedy NMS -> kept indices into ``boxes`` in descending score order (mmcv.ops.nms / batched_nms, which
    the teacher's per-step ``predict`` of det:72-74 calls).  ``idxs`` makes it class-/level-aware the way
    batched_nms does: boxes of different groups are shifted apart so they never overlap."""
    lib = _lib.load_library()
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes.float()
    if idxs is not None:
        b = b + (idxs.to(b) * (b.max() + 1))[:, None]
    b = b[order].contiguous()
    max_keep = n if max_keep is None else min(int(max_keep), n)
    keep = torch.empty(max_keep, dtype=torch.int64, device=boxes.device)
    n_keep = torch.empty(1, dtype=torch.int32, device=boxes.device)
    nbytes = lib.repre_nms_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=boxes.device)
    _lib.check(lib.repre_nms(_dev(b, "boxes"), n, float(iou_thr), max_keep, _dev(keep, "keep", torch.int64),
                             _dev(n_keep, "n_keep", torch.int32), C.c_void_p(ws.data_ptr()), nbytes, _stream()), "repre_nms")
    return order[keep[:int(n_keep.item())]]


def unpack_bitmask_row(words: torch.Tensor, n: int) -> torch.Tensor:
    """int64 words (little-endian bit order) -> bool[n] (host-side glue for mask.pth)."""
    w = words.cpu().numpy().view("uint64")
    import numpy as np
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n]
    return torch.from_numpy(bits.astype(bool))


def pack_bool_mask(mask: torch.Tensor) -> torch.Tensor:
    """bool[n] -> int64 words, the inverse of unpack_bitmask_row."""
    import numpy as np
    m = mask.cpu().numpy().astype(np.uint8)
    n = m.shape[0]
    pad = (-n) % 64
    if pad:
        m = np.concatenate([m, np.zeros(pad, np.uint8)])
    return torch.from_numpy(np.packbits(m, bitorder="little").view(np.int64).copy())
