"""CPU tests of the host layer: C-ABI surface, registry names, rank selection, argument
checking and bit-mask glue.  No kernel is launched here (there is no GPU in this container)."""
import os
import re

import numpy as np
import pytest
import torch

import inputs as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def N():
    import nsgp_repre_amd
    return nsgp_repre_amd


def test_library_loads_and_exports_every_declared_symbol(N):
    from nsgp_repre_amd import _lib
    lib = N.load_library()
    header = open(os.path.join(ROOT, "include", "nsgp_repre.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b((?:nsgp|repre)_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nsgp_abi_version() == 8
    assert lib.nsgp_device_count() >= 0


def test_struct_layout_matches_header(N):
    import ctypes as C
    from nsgp_repre_amd import _lib
    assert C.sizeof(_lib.TensorDesc) == 5 * 8 + 8 + 4 * 4 + 8 + 4 + 4 + 8 + 4 + 4 + 8  # + proj_split, split_scale (ABI 5), basis_rows (ABI 7)
    assert C.sizeof(_lib.Hyper) == 16 * 4


def test_argument_validation_without_gpu(N):
    """Error paths that return before any HIP call."""
    import ctypes as C
    from nsgp_repre_amd import _lib
    lib = N.load_library()
    assert lib.nsgp_plan_create(None, None, 0, 0, None, 0) == -1
    assert b"null" in lib.nsgp_last_error()
    assert lib.nsgp_project(None, None, None, 4, 4, 1.0, 0, None) == -1
    assert lib.nsgp_cov_workspace_bytes(0, 4, 4, 3, 3, 1, 1, 1, 1) == 0
    assert lib.nsgp_cov_workspace_bytes(256, 200, 336, 3, 3, 1, 1, 1, 1) > 256 * 202 * 338 * 4
    assert lib.nsgp_projector_scratch_bytes(4608) >= 8
    assert lib.repre_masked_mean_workspace_bytes(300, 12544) >= 12544 * 4
    assert lib.nsgp_plan_step(None, None, None, 1, None) == -1
    d = (_lib.TensorDesc * 1)()
    assert lib.nsgp_plan_workspace_bytes(d, 1, _lib.NSGP_OPT_SGD) == 0


def test_registry_names(N):
    reg = N.registry
    for name in ("SGDNSCL", "AdamWNSCL", "AdamNSCL", "SGDNSCLNA"):
        assert reg.OPTIMIZERS.get(name) is getattr(N, name)
    opt = reg.OPTIMIZERS.build(dict(type="SGDNSCL", lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True),
                               params=[torch.nn.Parameter(torch.zeros(3))])
    assert isinstance(opt, N.SGDNSCL) and opt.defaults["svd"] is True


def test_optimizer_surface_and_validation(N):
    p = [torch.nn.Parameter(torch.zeros(4, 4))]
    with pytest.raises(ValueError):
        N.SGDNSCL(p, lr=-1)
    with pytest.raises(ValueError):
        N.AdamWNSCL(p, betas=(1.0, 0.9))
    with pytest.raises(ValueError):
        N.AdamNSCL(p, eps=-1)
    opt = N.SGDNSCL(p, lr=0.1)
    assert opt.defaults == dict(lr=0.1, momentum=0, dampening=0, nesterov=False, weight_decay=0, svd=False, thres=1.001)
    assert N.AdamNSCL(p).defaults["thres"] == 0.99 and N.AdamWNSCL(p).defaults["thres"] == 1.001
    assert hasattr(opt, "eigens") and hasattr(opt, "transforms") and hasattr(opt, "get_eigens") and hasattr(opt, "get_transforms")
    # param groups carry no 'names' until the runner wires them; __setstate__ defaults it to []
    sd = opt.state_dict()
    opt2 = N.SGDNSCL(p, lr=0.1)
    opt2.__setstate__({"state": {}, "param_groups": [dict(g, params=p) for g in sd["param_groups"]], "defaults": opt.defaults})
    assert opt2.param_groups[0]["names"] == [] and opt2.param_groups[0]["svd"] is False
    # no names wired -> step is a no-op, exactly like the reference's zip(names, params)
    opt.param_groups[0]["names"] = []
    p[0].grad = torch.ones(4, 4)
    opt.step()
    assert torch.equal(p[0].detach(), torch.zeros(4, 4))


def test_step_refuses_cpu_tensors_and_missing_grads(N):
    p = [torch.nn.Parameter(torch.zeros(4, 4))]
    opt = N.SGDNSCL(p, lr=0.1)
    opt.param_groups[0]["names"] = ["w"]
    with pytest.raises(AttributeError):
        opt.step()  # p.grad is None: the reference raises AttributeError too (SGD_NSCL.py:75)
    p[0].grad = torch.ones(4, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        opt.step()


def test_elbow_index_matches_reference_goldens(N, golden_dir):
    from nsgp_repre_amd.optim.threshold import elbow_index
    g = np.load(os.path.join(golden_dir, "g2_thresholds.npz"))
    for si, s in enumerate(I.g2_spectra()):
        for oi, off in enumerate(I.G2_OFFSETS):
            assert elbow_index(s, off, "sgd") == int(g[f"sgd_{si}_{oi}"]), (si, off)
            assert elbow_index(s, off, "adam") == int(g[f"adam_{si}_{oi}"]), (si, off)


def test_adaptive_threshold_mask_is_suffix(N):
    opt = N.SGDNSCL([torch.nn.Parameter(torch.zeros(1))])
    s = torch.from_numpy(I.g2_spectra()[7])
    m = opt.adaptive_threshold(s, 0.0)
    assert m.dtype == torch.bool and m.shape == s.shape
    i = int(m.to(torch.int8).argmax())
    assert m[i:].all() and not m[:i].any()
    optA = N.AdamWNSCL([torch.nn.Parameter(torch.zeros(1))])
    assert int(optA.adaptive_threshold(s, 0.3).to(torch.int8).argmax()) != int(opt.adaptive_threshold(s, 0.3).to(torch.int8).argmax())


def test_bitmask_pack_roundtrip(N):
    from nsgp_repre_amd import ops
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 300):
        m = torch.from_numpy(rng.random(n) < 0.3)
        w = ops.pack_bool_mask(m)
        assert w.dtype == torch.int64 and w.numel() == (n + 63) // 64
        assert torch.equal(ops.unpack_bitmask_row(w, n), m)


def test_ops_refuse_cpu(N):
    from nsgp_repre_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.build_projector(torch.eye(8), 2, True)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.sim_counts(torch.ones(4, 8))
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.cov_accumulate_conv2d(torch.ones(1, 2, 4, 4), (3, 3), (1, 1), (1, 1))


def test_spectrum_plot_is_written(N, tmp_path, monkeypatch):
    """get_eigens(..., distinguisher=...) plots like the reference (SGD_NSCL.py:383-384); CPU tensors are
    fine for this host-side part."""
    monkeypatch.chdir(tmp_path)
    p = torch.nn.Parameter(torch.zeros(4, 16))
    opt = N.SGDNSCL([p], svd=True)
    opt.param_groups[0]["names"] = ["fc.weight"]
    C = torch.from_numpy(I.covariance_like(16, 3))
    opt.get_eigens({"fc.weight": C}, distinguisher="unit")
    assert (tmp_path / "figures" / "svals_task1_unit.png").exists()
    sv = opt.eigens["fc.weight"]["eigen_value"]
    assert bool((sv[:-1] >= sv[1:]).all())
    V = opt.eigens["fc.weight"]["eigen_vector"]
    assert torch.allclose(V @ torch.diag(sv) @ V.t(), C, rtol=1e-3, atol=1e-3 * float(C.abs().max()))
