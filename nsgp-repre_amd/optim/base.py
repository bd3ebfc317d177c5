"""Common machinery of the four NSGP optimizers.

Interface mirror of the reference's ``SGDNSCL`` / ``AdamWNSCL`` / ``AdamNSCL`` /
``SGDNSCLNA`` (mmdet/engine/optimizers/*.py): ``torch.optim.Optimizer`` subclasses
with the extra ``get_eigens`` / ``get_transforms`` / ``adaptive_threshold`` methods,
the public ``eigens`` / ``transforms`` dicts and the extra param-group key
``names`` (wired by the runner, nsrunner_roi_replay.py:473-484).

What differs is *how* a step runs: instead of a Python loop of ~160 x 5 tiny
elementwise launches plus 50 `torch.mm`, ``step()`` hands one table to
``nsgp_plan_step`` (C ABI) on the current stream: one multi-tensor HIP kernel, then -- for projectors this
optimizer built itself (``get_transforms`` / ``set_basis``) with at most 256 removed directions -- the low-rank
form ``p += c (u - (u U) U^T)`` (the layers' update fused with ``T = u U``, an ordered slab reduce, the apply launch:
exact fp32 MFMA, HBM-bound), and for every other projector one
grouped dense MFMA GEMM ``u @ P`` (two-term fp16 split by default, fp32 MFMA with ``split_mfma = False``).
"""
import ctypes as C
import logging
from operator import is_ as _is
from collections import defaultdict

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from .. import _lib, ops
from .threshold import elbow_index

logger = logging.getLogger("nsgp_repre_amd")

#: largest number of removed directions the low-rank form of the step takes (csrc/projected_step.hip: LR_MAX_RANK)
LOW_RANK_MAX = 256

#: default of ``optimizer.split_mfma`` (see NSCLOptimizerBase.__init__): False | "f16x2" (True = "f16x2")
SPLIT_MFMA_DEFAULT = "f16x2"
_SPLIT_KINDS = {False: 0, None: 0, "f32": 0, "f16x2": 2, True: 2}

#: plan handles of optimizers that were garbage-collected without ``close()``.  ``__del__`` may run wherever the
#: interpreter happens to collect -- on an autograd thread, in the middle of another test's backward -- so it makes NO HIP
#: call: it parks the handles here and the next explicit, main-thread entry point (``step`` building plans, ``close``,
#: ``release_collected_plans``) destroys them.
_GRAVEYARD = []


def release_collected_plans():
    """Destroy the plans parked by ``__del__`` (device tables, pinned ring, events).  Returns how many were released;
    a failing ``hipFree`` / ``hipEventDestroy`` is reported, not swallowed."""
    if not _GRAVEYARD:
        return 0
    lib = _lib.load_library()
    n = 0
    while _GRAVEYARD:
        handle = _GRAVEYARD.pop()
        rc = lib.nsgp_plan_destroy(handle)
        if rc != 0:
            msg = lib.nsgp_last_error()
            logger.error("nsgp_plan_destroy failed with code %d: %s", rc, msg.decode() if msg else "")
        n += 1
    return n


class NSCLOptimizerBase(Optimizer):
    _kind = _lib.NSGP_OPT_SGD   # which C-ABI update rule
    _threshold_rule = "sgd"     # offset rule of adaptive_threshold
    _normalise_all = False      # Adam_NSCL.py:183 normalises every projector

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self.eigens = defaultdict(dict)
        self.transforms = defaultdict(dict)
        self.count = 0
        #: mirror the reference's in-place mutation of ``p.grad`` (weight decay / Nesterov add)
        self.mutate_grad = True
        #: projectors built by get_transforms / set_basis with r <= LOW_RANK_MAX removed directions (and D % 32 == 0) are built
        #: in the HEAD form P = c (I - U U^T) from the orthonormalised top-r eigenvectors U, and the step applies them as
        #: p += c (u - (u U) U^T): 4*Cout*D*r FLOP and no projector traffic instead of 2*Cout*D^2 FLOP + D^2 projector bytes.
        #: ``transforms[name]`` still holds the dense P (the reference's public attribute); dense u @ P and the low-rank
        #: form are the same function of (u, U) up to the rounding of P's entries.  False = always build V_tail V_tail^T and
        #: always run the dense GEMM (the reference's literal formula).  Projectors assigned from outside
        #: (``transforms[name] = P``) always take the dense GEMM.
        self.low_rank = True
        #: orthonormalise the r head vectors (one Newton-Schulz step U <- U (1.5 I - 0.5 U^T U): an fp32 eigh leaves them
        #: orthonormal to ~1e-6, afterwards to ~1e-12) before the head-form projector is built from them
        self.polish_basis = True
        #: dense projection on the fp16 matrix cores with fp32 accumulation and fp32-level error per output row:
        #: "f16x2" (default) -- two fp16 terms per operand, three MFMAs per fp32-equivalent product, one power-of-two scale
        #: per row of the update and per column of the projector (csrc/gemm_f16x2_v2.hpp; 4 extra bytes per projector element).
        #: The split copy of a projector is made once and redone if the projector tensor is replaced or modified in place.
        #: False = fp32 MFMA.  (A three-term bf16 split existed until round 2: slower on every table, removed.)
        self.split_mfma = SPLIT_MFMA_DEFAULT
        #: get_eigens: at most this many equal-width covariances per batched eigh call (1 = one call per layer)
        self.eigh_batch = 16
        #: get_eigens: host threads (each with its own HIP stream) the solver calls are issued from; 1 = the calling thread only
        self.eigh_threads = 4
        self._splits = {}
        self._basis = {}
        self._plans = []
        self._plan_key = None
        self._fast = None
        self._workspaces = []

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("svd", False)
            group.setdefault("names", [])
        # load_state_dict() replaces every state tensor: the plans hold the old pointers
        if getattr(self, "_plans", None):
            self._destroy_plans()
        self._plans, self._plan_key, self._workspaces, self._fast = [], None, [], None
        if not hasattr(self, "_basis"):
            self._basis, self.low_rank, self.mutate_grad, self.polish_basis = {}, True, True, True
            self.eigh_batch, self.eigh_threads = 16, 4
            self.split_mfma, self._splits = SPLIT_MFMA_DEFAULT, {}

    def close(self):
        """Release the GPU resources of this optimizer's plans (device tables, pinned upload ring, events) NOW, on the
        calling thread, with every HIP return code checked.  The runner calls it when a task's optimizer is done; the
        optimizer stays usable (the next ``step`` builds fresh plans)."""
        self._destroy_plans()
        self._splits = {}
        release_collected_plans()

    def __del__(self):
        # no HIP call from a finaliser (see _GRAVEYARD): hand the raw handles over and forget them
        try:
            for plan in getattr(self, "_plans", None) or []:
                _GRAVEYARD.append(plan["handle"])
            self._plans = []
        except Exception:
            pass

    # ------------------------------------------------------------------ spectrum -> projector
    def adaptive_threshold(self, svals: torch.Tensor, offset: float = 0):
        """Bool mask [D], True from the elbow index on (the small-sigma tail that spans the
        null space).  Host-side like the reference (SGD_NSCL.py:98-177: numpy + scipy)."""
        i_thres = elbow_index(svals.detach().cpu().numpy(), offset, self._threshold_rule)
        mask = torch.zeros(svals.shape[0], dtype=torch.bool, device=svals.device)
        mask[i_thres:] = True
        return mask

    def _svd_named(self):
        for group in self.param_groups:
            if group["svd"] is False:
                continue
            for n, p in zip(group["names"], group["params"]):
                yield group, n, p

    def get_eigens(self, fea_in, distinguisher=None):
        """Spectrum of every covariance present in ``fea_in`` (SGD_NSCL.py:360-380 runs
        ``torch.svd(C, some=False)``).  C is symmetric PSD, so its SVD is its
        eigendecomposition: ``eigh`` (rocSOLVER on the GPU) gives the same singular values
        (|lambda|, sorted descending) and right singular vectors up to sign -- and the
        projector built from them is sign-invariant."""
        names = [n for _, n, _p in self._svd_named() if n in fea_in.keys()]
        # layers of equal width go to the solver as ONE batched call (R-50-FPN: ten 2304-wide, seven 1024-wide, ... covariances):
        # the same per-matrix algorithm, far fewer launches and host round trips
        by_width = {}
        for n in names:
            by_width.setdefault((fea_in[n].shape[0], fea_in[n].device), []).append(n)
        chunks = []
        for (_d, _dev), group in by_width.items():
            for lo in range(0, len(group), self.eigh_batch):
                chunks.append(group[lo:lo + self.eigh_batch])
        chunks.sort(key=lambda c: -len(c) * fea_in[c[0]].shape[0] ** 3)      # longest first

        def solve(chunk):
            if len(chunk) == 1:
                lam, Q = torch.linalg.eigh(fea_in[chunk[0]])
                lam, Q = lam[None], Q[None]
            else:
                lam, Q = torch.linalg.eigh(torch.stack([fea_in[n] for n in chunk]))
            out = []
            for i, n in enumerate(chunk):
                s_ = lam[i].abs()
                order = torch.argsort(s_, descending=True, stable=True)
                out.append((n, s_[order].contiguous(), Q[i][:, order].contiguous()))
            return out
        on_gpu = bool(chunks) and all(fea_in[c[0]].is_cuda for c in chunks)
        n_threads = min(int(self.eigh_threads), len(chunks)) if on_gpu else 1
        if n_threads <= 1:
            results = [solve(c) for c in chunks]
        else:
            # The sweep is HOST-bound: rocSOLVER's syevd is ~1,200 small launches per matrix, issued from the calling thread (~60 k
            # for the 50 R-50-FPN layers, 0.5 s, GPU mostly idle).  The calls are independent, so they are issued from several host
            # threads, each on its own HIP stream (torch releases the GIL inside the solver call); per-matrix arithmetic is unchanged.
            import concurrent.futures
            dev = fea_in[chunks[0][0]].device
            cur = torch.cuda.current_stream(dev)
            streams = [torch.cuda.Stream(device=dev) for _ in range(n_threads)]
            for st in streams:
                st.wait_stream(cur)
            import queue
            free = queue.SimpleQueue()
            for st in streams:
                free.put(st)

            def work(chunk):
                st = free.get()
                try:
                    with torch.cuda.device(dev), torch.cuda.stream(st):
                        return solve(chunk)
                finally:
                    free.put(st)
            with concurrent.futures.ThreadPoolExecutor(max_workers=n_threads) as pool:
                results = list(pool.map(work, chunks))
            for st in streams:
                cur.wait_stream(st)
            for res in results:
                for _n, sv, V in res:
                    sv.record_stream(cur)
                    V.record_stream(cur)
        for res in results:
            for n, sv, V in res:
                eigen = self.eigens[n]
                eigen["eigen_value"], eigen["eigen_vector"] = sv, V
        if distinguisher is not None:
            self.plot_sval_figures(self.eigens, distinguisher)

    def plot_sval_figures(self, svals_dict, distinguisher=None, offset=0.0):
        """Spectrum plots, kept / null-space parts in two colours, one panel per layer, written to
        ``./figures/svals_task1_<distinguisher>.png`` (SGD_NSCL.py:180-201; host-side matplotlib)."""
        import os
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        keys = list(svals_dict.keys())
        fig, axes = plt.subplots(len(keys) // 4 + 1, 4, squeeze=False)
        fig.set_figheight(60)
        fig.set_figwidth(15)
        for i, k in enumerate(keys):
            sv = svals_dict[k]["eigen_value"]
            pts = sv.detach().cpu().numpy()
            i_thres = elbow_index(pts, offset, self._threshold_rule)
            ax = axes[i // 4, i % 4]
            ax.plot(np.arange(i_thres + 1), pts[:i_thres + 1], color="blue")
            ax.plot(np.arange(i_thres, len(pts)), pts[i_thres:], color="red")
            ax.set_title(k)
        os.makedirs("./figures", exist_ok=True)
        fig.tight_layout()
        fig.savefig(os.path.join("./figures", f"svals_task{1}_{distinguisher}.png"))
        plt.close(fig)

    def _null_space_start(self, group, n):
        """Index of the first basis column kept (the mask is always a suffix)."""
        mask = self.adaptive_threshold(self.eigens[n]["eigen_value"], offset=self._offset)
        return int(mask.to(torch.int8).argmax().item())

    def _normalise(self, n):
        return self._normalise_all or ("backbone" in n)

    def get_transforms(self, offset=0.0):
        """P = V_tail V_tail^T (/ ||P||_F for backbone layers) per named parameter
        (SGD_NSCL.py:203-290), built by the HIP SYRK kernel ``nsgp_build_projector``."""
        self._offset = offset
        for group, n, _p in self._svd_named():
            if n not in self.eigens.keys():
                logger.info("missing keys: %s", n)
                continue
            sv = self.eigens[n]["eigen_value"]
            first = self._null_space_start(group, n)
            kept = sv.shape[0] - first
            logger.info("%s: reserving basis %d/%d; cond: %s, radio:%s", n, kept, sv.shape[0],
                        float(sv[0] / sv[first]), float(sv[first:].sum() / sv.sum()))
            self.set_basis(n, self.eigens[n]["eigen_vector"], first)

    def set_basis(self, name: str, V: torch.Tensor, rank: int, normalise=None):
        """Install the projector of one parameter from an orthonormal eigenbasis ``V`` (columns in descending eigenvalue
        order) whose first ``rank`` columns are removed: ``transforms[name] = V[:, rank:] V[:, rank:]^T``
        (Frobenius-normalised per the optimizer's rule, SGD_NSCL.py:270-285).

        With ``low_rank`` (default) and ``0 < rank <= LOW_RANK_MAX`` the SAME projector is built from the other side,
        ``I - U U^T`` with ``U`` the orthonormalised ``V[:, :rank]`` (HIP kernel ``nsgp_build_projector_head``), and
        ``U`` is remembered so that ``step`` can apply it as ``c (u - (u U) U^T)``; for an orthonormal ``V`` the two
        constructions are the same matrix (an fp32 eigensolver's ``V`` is orthonormal to ~1e-6: the two differ by that
        much, both sit at the same distance from the fp64 projector -- tests/test_gpu_parity.py).  Otherwise the HIP SYRK
        kernel builds ``V_tail V_tail^T`` and the step runs the dense GEMM."""
        normalise = self._normalise(name) if normalise is None else normalise
        D, rank = V.shape[0], int(rank)
        if self.low_rank and 0 < rank <= LOW_RANK_MAX and D % 32 == 0:
            U = V[:, :rank]
            if self.polish_basis:       # on the r head columns only: an r x r Gram matrix, once per layer per task
                U = 1.5 * U - 0.5 * (U @ (U.t() @ U))
            rpad = 32 if rank <= 32 else (64 if rank <= 64 else (128 if rank <= 128 else 256))     # U's padded width (csrc: lr_rpad)
            U_rm = torch.zeros(D, rpad, dtype=torch.float32, device=V.device)
            U_rm[:, :rank] = U
            U_kq = U_rm.view(D // 4, 4, rpad).permute(0, 2, 1).contiguous()     # k-quads [D/4][rpad][4]
            P, norm = ops.build_projector_head(U_rm, normalise, return_norm=True)
            # the caches hold the projector OBJECT (not its address: the allocator hands a freed [D x D] block straight
            # to the next layer's projector) and compare identity + version
            self._basis[name] = dict(U_rm=U_rm, U_kq=U_kq, rank=rank, norm=norm, c=None, P=P, P_version=P._version)
        else:
            if self.low_rank:        # the default form does not apply: say so (the dense GEMM costs ~3 x the step time of such a layer)
                why = ("no direction removed" if rank <= 0 else f"{rank} removed directions > {LOW_RANK_MAX}" if rank > LOW_RANK_MAX
                       else f"D = {D} is not a multiple of 32")
                logger.info("%s: projector applied as a dense GEMM, not in the low-rank form (%s)", name, why)
            P, norm = ops.build_projector(V, rank, normalise, return_norm=True)
            self._basis.pop(name, None)
        self.transforms[name] = P.detach_()
        self._splits.pop(name, None)
        self._plan_key = None  # new projector buffers -> new plan
        self._fast = None

    def invalidate_projector(self, name=None):
        """Forget every derived copy (fp16 / bf16 split, low-rank basis link) of ``transforms[name]`` -- of all
        projectors when ``name`` is None.  Needed only after rewriting a projector's memory behind torch's back (a raw
        pointer write that does not bump the tensor's version counter); assigning a new tensor, an in-place torch op
        or ``set_basis`` are detected on their own."""
        for n in ([name] if name is not None else list(self._splits)):
            self._splits.pop(n, None)
        for n in ([name] if name is not None else list(self._basis)):
            self._basis.pop(n, None)
        self._fast = None

    # ------------------------------------------------------------------ step
    def _init_state(self, p, state, group):
        raise NotImplementedError

    def _state_tensors(self, state, group):
        raise NotImplementedError

    def _fill_hyper(self, h, group, step):
        raise NotImplementedError

    def _destroy_plans(self):
        """Explicit, caller's-thread teardown (never from ``__del__``): each plan drains its own upload events before its
        buffers go, and a failing HIP call raises."""
        plans, self._plans = self._plans, []
        self._fast = None
        if plans:
            lib = _lib.load_library()
            for plan in plans:
                _lib.check(lib.nsgp_plan_destroy(plan["handle"]), "nsgp_plan_destroy")
        self._workspaces = []       # after the plans: nsgp_plan_destroy has synchronised every launch that used them

    def _build_plans(self, entries):
        """entries: list of (group_index, name, p, state).  One plan per <= NSGP_MAX_HYPER groups."""
        lib = _lib.load_library()
        self._destroy_plans()
        release_collected_plans()
        # drop the split copies of projectors that were replaced or edited since they were made
        self._splits = {n: c for n, c in self._splits.items()
                        if self.transforms.get(n) is c["P"] and c["P"]._version == c["version"]}
        groups = sorted({gi for gi, *_ in entries})
        for lo in range(0, len(groups), _lib.NSGP_MAX_HYPER):
            gset = groups[lo:lo + _lib.NSGP_MAX_HYPER]
            slot = {gi: k for k, gi in enumerate(gset)}
            sub = [e for e in entries if e[0] in slot]
            descs = (_lib.TensorDesc * len(sub))()
            for d, (gi, n, p, st) in zip(descs, sub):
                group = self.param_groups[gi]
                s0, s1, s2 = self._state_tensors(st, group)
                d.param = p.data.data_ptr()
                d.state0 = s0.data_ptr()
                d.state1 = s1.data_ptr() if s1 is not None else None
                d.state2 = s2.data_ptr() if s2 is not None else None
                d.numel = p.numel()
                d.hyper = slot[gi]
                P = self.transforms.get(n) if (group["svd"] and len(self.transforms) > 0 and n in self.transforms.keys()) else None
                if P is not None:
                    if p.dim() not in (2, 4):
                        raise RuntimeError(f"{n}: projected parameters must be 2-D or 4-D (SGD_NSCL.py:83-90)")
                    rows = p.shape[0]
                    cols = p.numel() // rows
                    if tuple(P.shape) != (cols, cols) or not P.is_cuda or P.dtype != torch.float32 or not P.is_contiguous():
                        raise RuntimeError(f"{n}: transform must be a contiguous fp32 GPU [{cols}x{cols}] tensor, got "
                                           f"{tuple(P.shape)} {P.dtype} {P.device}")
                    d.proj = P.data_ptr()
                    d.rows, d.cols = rows, cols
                    b = self._basis.get(n)
                    if (self.low_rank and b is not None and b["P"] is P and b["P_version"] == P._version
                            and rows % 32 == 0 and cols % 32 == 0):
                        # only for head-form projectors this optimizer built itself and nobody touched since
                        if b["c"] is None:
                            b["c"] = 1.0 / float(b["norm"])
                        d.basis, d.basis_rows = b["U_kq"].data_ptr(), b["U_rm"].data_ptr()
                        d.rank = int(b["rank"])
                        d.basis_scale = b["c"]
                    else:
                        kind = _SPLIT_KINDS[self.split_mfma]
                        if kind == 2 and rows % 128 == 0 and cols % 128 == 0:
                            sp, sc = self._split_of(n, P, kind)
                            d.proj_split, d.split_kind, d.split_scale = sp.data_ptr(), kind, sc
                else:
                    d.proj = None
            nbytes = lib.nsgp_plan_workspace_bytes(descs, len(sub), self._kind)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=sub[0][2].device)
            handle = C.c_void_p()
            _lib.check(lib.nsgp_plan_create(C.byref(handle), descs, len(sub), self._kind,
                                            C.c_void_p(ws.data_ptr()), nbytes), "nsgp_plan_create")
            self._workspaces.append(ws)
            self._plans.append(dict(handle=handle, entries=sub, groups=gset,
                                    grads=(C.c_void_p * len(sub))(), hyper=(_lib.Hyper * len(gset))()))

    def _split_of(self, name: str, P: torch.Tensor, kind: int):
        """(split copy of ``P^T``, scale) of the given kind for ``transforms[name]``.  Cached per parameter name together
        with the projector tensor itself: a hit needs the SAME tensor object at the SAME version (holding the reference
        also keeps its address from being recycled while the copy is alive)."""
        c = self._splits.get(name)
        if c is None or c["P"] is not P or c["version"] != P._version or c["kind"] != kind:
            split, scale = ops.split_projector_f16(P), 0.0
            c = self._splits[name] = dict(P=P, version=P._version, kind=kind, split=split, scale=scale)
        return c["split"], c["scale"]

    def uses_split_mfma(self):
        """The split kind every current plan runs its dense projection launch on: False or "f16x2"."""
        lib = _lib.load_library()
        kinds = {lib.nsgp_plan_uses_split_mfma(p["handle"]) for p in self._plans}
        return {2: "f16x2"}.get(kinds.pop(), False) if len(kinds) == 1 else False

    def plan_stats(self):
        """(gemm_flops, algorithmic_bytes, n_tiles, n_projected) summed over the current plans."""
        lib = _lib.load_library()
        tot = [0.0, 0.0, 0, 0]
        for plan in self._plans:
            f, b, t, n = C.c_double(), C.c_double(), C.c_int(), C.c_int()
            _lib.check(lib.nsgp_plan_stats(plan["handle"], C.byref(f), C.byref(b), C.byref(t), C.byref(n)))
            tot = [tot[0] + f.value, tot[1] + b.value, tot[2] + t.value, tot[3] + n.value]
        return tuple(tot)

    def tile_counts(self):
        """(fast 128x128 tiles, guarded 128x128 tiles, fp16-split 256x128 tiles) of the dense projection, over the current plans."""
        lib = _lib.load_library()
        tot = [0, 0, 0]
        for plan in self._plans:
            a, b, c = C.c_int(), C.c_int(), C.c_int()
            _lib.check(lib.nsgp_plan_tile_counts(plan["handle"], C.byref(a), C.byref(b), C.byref(c)))
            tot = [tot[0] + a.value, tot[1] + b.value, tot[2] + c.value]
        return tuple(tot)

    def lowrank_stats(self):
        """(layers on the low-rank form, their FLOPs 4*Cout*D*r, workgroups of the T = u U launch, wave units of the apply
        launch) over the current plans."""
        lib = _lib.load_library()
        tot = [0, 0.0, 0, 0]
        for plan in self._plans:
            n, f, a, b = C.c_int(), C.c_double(), C.c_int(), C.c_int()
            _lib.check(lib.nsgp_plan_lowrank_stats(plan["handle"], C.byref(n), C.byref(f), C.byref(a), C.byref(b)))
            tot = [tot[0] + n.value, tot[1] + f.value, tot[2] + a.value, tot[3] + b.value]
        return tuple(tot)

    def launch_shape(self):
        """Workgroups per launch of a step, summed over the current plans: (multi-tensor update launch, low-rank units of the fused
        update + T launch, chunks of the un-projected tensors riding in that launch, slab reduce, apply launch)."""
        lib = _lib.load_library()
        tot = [0] * 5
        for plan in self._plans:
            sh = (C.c_int * 5)()
            _lib.check(lib.nsgp_plan_launch_shape(plan["handle"], sh), "nsgp_plan_launch_shape")
            tot = [a + b for a, b in zip(tot, sh)]
        return tuple(tot)

    def profile_begin(self, max_steps=1024):
        """Record HIP events around both launches of the next ``max_steps`` steps (measurement)."""
        lib = _lib.load_library()
        for plan in self._plans:
            _lib.check(lib.nsgp_plan_profile_begin(plan["handle"], max_steps), "nsgp_plan_profile_begin")

    def profile_end(self):
        """(n_steps, avg elementwise-launch ms, avg projection-GEMM ms), summed over plans."""
        lib = _lib.load_library()
        n_, u_, g_ = 0, 0.0, 0.0
        for plan in self._plans:
            n, u, g = C.c_int(), C.c_float(), C.c_float()
            _lib.check(lib.nsgp_plan_profile_end(plan["handle"], C.byref(n), C.byref(u), C.byref(g)), "nsgp_plan_profile_end")
            n_, u_, g_ = max(n_, n.value), u_ + u.value, g_ + g.value
        return n_, u_, g_

    def profile_detail(self):
        """The last ``profile_end`` launch by launch, ms summed over plans: (multi-tensor elementwise launch, fused update + T
        launch of the low-rank layers, dense GEMM launches, slab reduce, low-rank apply launch)."""
        lib = _lib.load_library()
        tot = [0.0] * 5
        for plan in self._plans:
            ms = (C.c_float * 5)()
            _lib.check(lib.nsgp_plan_profile_detail(plan["handle"], ms), "nsgp_plan_profile_detail")
            tot = [a + b for a, b in zip(tot, ms)]
        return tuple(tot)

    def _validate(self, n, p, group):
        if p.grad is None:  # the reference dereferences p.grad.data unconditionally (:75)
            raise AttributeError(f"{n}: 'NoneType' object has no attribute 'data' (parameter has no grad)")
        if p.grad.is_sparse:
            raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
        if not p.is_cuda:
            raise RuntimeError(f"{n}: nsgp_repre_amd optimizers run on the GPU only (no CPU fallback)")
        if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
            raise TypeError(f"{n}: parameters and gradients must be fp32")
        if not p.is_contiguous() or not p.grad.is_contiguous():
            raise RuntimeError(f"{n}: parameter/gradient must be contiguous")

    def _structure_unchanged(self) -> bool:
        """Per-step check that everything the plans hold raw pointers to is still where it was: the same Parameter objects on the
        same storage, the same state dicts (``load_state_dict`` swaps them), the same projector tensors at the same version (an
        in-place edit invalidates the split copy), the same names / ``svd`` flags.  The training step is host-bound, so this runs
        on flat lists built once per plan (list comparisons in C) instead of a Python loop over 162 tensors: ~35 us."""
        f = self._fast
        if f is None or f["flags"] != (bool(self.low_rank), _SPLIT_KINDS[self.split_mfma]) or len(self.param_groups) != len(f["groups"]):
            return False
        st_get = self.state.get
        for group, (params, names, svd) in zip(self.param_groups, f["groups"]):
            cur = group["params"]       # the same objects in the same order (identity: `==` on tensors is elementwise), same names, same flag
            if group["svd"] != svd or len(cur) != len(params) or not all(map(_is, cur, params)) or group["names"] != names:
                return False
        plist = f["plist"]
        if [p.data_ptr() for p in plist] != f["ptrs"] or not all(map(_is, map(st_get, plist), f["states"])):
            return False
        tr = self.transforms
        if len(tr) != f["n_transforms"]:
            return False
        for n, rP, rver, rptr in f["proj"]:
            P = tr.get(n)
            if P is not rP or P._version != rver or P.data_ptr() != rptr:
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        """One optimization step (SGD_NSCL.py:59-96 semantics, every listed (name, p) pair).

        Host side per step: one pass to fingerprint the (name, parameter, projector) structure, and
        -- while it is unchanged -- only the gradient pointers and the per-group scalars are refreshed
        before the single ``nsgp_plan_step`` call per plan."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load_library()
        if not self._structure_unchanged():
            entries, groups = [], []
            for gi, group in enumerate(self.param_groups):
                for n, p in zip(group["names"], group["params"]):
                    self._validate(n, p, group)
                    state = self.state[p]
                    if len(state) == 0:
                        self._init_state(p, state, group)
                    entries.append((gi, n, p, state))
            if not entries:
                return loss
            self._build_plans(entries)
            # what the plans point at, for the per-step check: flat lists (parameters, their storages, their state dicts) and the
            # projectors that are in use (tensor, version, storage); a projected name whose projector appears or disappears changes
            # len(transforms) or one of the recorded identities
            tr = self.transforms
            plist, proj = [], []
            for group in self.param_groups:
                svd = group["svd"]
                groups.append((list(group["params"]), list(group["names"]), svd))
                plist += group["params"]
                for n in group["names"]:
                    if svd and n in tr:
                        P = tr[n]
                        proj.append((n, P, P._version, P.data_ptr()))
            self._fast = dict(flags=(bool(self.low_rank), _SPLIT_KINDS[self.split_mfma]), groups=groups, plist=plist,
                              ptrs=[p.data_ptr() for p in plist], states=[self.state[p] for p in plist], proj=proj, n_transforms=len(tr))
            for plan in self._plans:     # per plan: its parameters, states and names as flat lists, and the index range of each of its groups
                ents = plan["entries"]
                plan["params"], plan["states"], plan["names"] = [e[2] for e in ents], [e[3] for e in ents], [e[1] for e in ents]
                plan["ranges"] = []
                for gi in plan["groups"]:
                    idx = [i for i, e in enumerate(ents) if e[0] == gi]
                    plan["ranges"].append((gi, idx[0], idx[-1] + 1))       # entries are in group order: a contiguous run
        elif not self._plans:
            return loss
        stream = C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        mutate = int(self.mutate_grad)
        for plan in self._plans:
            glist = [p.grad for p in plan["params"]]
            for i, g in enumerate(glist):      # identity, not `None in glist`: `in` falls back to Tensor.__eq__(None), an op per tensor
                if g is None:
                    raise AttributeError(f"{plan['names'][i]}: 'NoneType' object has no attribute 'data' (parameter has no grad)")
            plan["grads"][:] = [g.data_ptr() for g in glist]
            steps = [st["step"] for st in plan["states"]]
            hyper = plan["hyper"]
            for k, (gi, lo, hi) in enumerate(plan["ranges"]):
                t = steps[lo]
                if steps[lo:hi].count(t) != hi - lo:
                    raise RuntimeError("parameters of one param group carry different step counts")
                self._fill_hyper(hyper[k], self.param_groups[gi], t + 1)
                hyper[k].write_grad = mutate
            _lib.check(lib.nsgp_plan_step(plan["handle"], plan["grads"], hyper, len(plan["groups"]), stream), "nsgp_plan_step")
            for st in plan["states"]:        # the launches are queued: only now do the step counters advance
                st["step"] += 1
        return loss
