"""a task-1 work dir WRITTEN BY THE REFERENCE'S OWN CODE (tests/golden/handoff/, produced by make_golden.py handoff:
the reference's cal_fea_in / cal_rois / calculate_save_importance and its task-2 head's mask.pth) consumed by this package.

CPU part: the files load through the package's loader without executing anything from them (two of them are pickled
``defaultdict`` objects that ``torch.load(weights_only=True)`` refuses), and the ORACLE makes of them what the reference's own
task-2 start made of them (expected.npz).  The GPU part (product task-2 start on the HIP path) is at the bottom, ``-m gpu``."""
import os
import pickle
import shutil
import zipfile

import numpy as np
import pytest
import torch

import inputs as I
import nsgp_oracle as O


@pytest.fixture()
def handoff(golden_dir, tmp_path):
    """A private copy (the task-2 head writes mask.pth next to it)."""
    dst = tmp_path / "handoff"
    shutil.copytree(os.path.join(golden_dir, "handoff"), dst)
    return str(dst)


def _key(n):
    return n.replace(".", "_")


def test_reference_written_files_load_without_executing_them(handoff):
    from nsgp_repre_amd.runner.safe_load import load_handoff
    E = np.load(os.path.join(handoff, "expected.npz"))
    w1 = os.path.join(handoff, "work_1")
    # what the reference wrote: a defaultdict at the top (covariance) / inside (EWC terms) -- refused by the weights-only loader
    assert str(E["cov_type"]) == "defaultdict" and list(E["ewc_types"]) == ["dict", "defaultdict"]
    for f in ("covariance.pth", "ewc_reg_terms_ewc.pth"):
        with pytest.raises(pickle.UnpicklingError, match="defaultdict"):
            torch.load(os.path.join(w1, f), weights_only=True)
    cov = load_handoff(os.path.join(w1, "covariance.pth"))
    assert type(cov) is dict and sorted(cov) == list(E["cov_keys"])
    assert all(v.dtype == torch.float32 and v.shape[0] == v.shape[1] for v in cov.values())
    terms = load_handoff(os.path.join(w1, "ewc_reg_terms_ewc.pth"))
    assert type(terms["importance"]) is dict and sorted(terms["importance"]) == ["backbone.bn1.bias", "backbone.bn1.weight"]
    assert all(len(v) == 1 and v[0].shape == (1, 8) for v in terms["task_param"].values())
    rois = load_handoff(os.path.join(w1, "rois_etc.pth"))           # plain lists: the weights-only loader takes them
    assert [tuple(t.shape) for t in rois] == [(38, 12544), (38,), (38,), (38, 4), (38, 4), (38, 5)]
    masks = load_handoff(os.path.join(handoff, "work_2", "mask.pth"))
    assert len(masks) == 3 and all(m.dtype == torch.bool for ml in masks for m in ml)
    ck = load_handoff(os.path.join(w1, "best_task1.pth"))
    assert set(ck) == {"meta", "state_dict"}


def test_allow_list_loader_resolves_nothing_else(tmp_path):
    """A pickle that names any other global (here os.system through a defaultdict factory, and a plain reduce) is rejected before
    anything is called."""
    from nsgp_repre_amd.runner.safe_load import load_handoff
    import collections

    def archive(payload, name):
        path = tmp_path / name
        with zipfile.ZipFile(path, "w") as zf:
            zf.writestr("archive/data.pkl", payload)
            zf.writestr("archive/byteorder", "little")
            zf.writestr("archive/version", "3\n")
        return str(path)
    marker = tmp_path / "executed"
    evil = b"\x80\x02ccollections\ndefaultdict\nq\x00cos\nsystem\nq\x01\x85q\x02Rq\x03."      # defaultdict(os.system)
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        load_handoff(archive(evil, "a.pth"))

    class Boom:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    with pytest.raises(pickle.UnpicklingError):
        load_handoff(archive(pickle.dumps({"x": collections.defaultdict(list), "y": Boom()}, protocol=2), "b.pth"))
    assert not marker.exists()


def test_oracle_task2_start_from_the_references_files(handoff):
    """covariance.pth -> spectra, ranks, projectors, one SGDNSCL step; rois_etc.pth (+ the reference's mask.pth) -> prototype bank;
    ewc_reg_terms_ewc.pth -> EWC loss: the oracle lands on what the reference's own classes made of the same files."""
    from nsgp_repre_amd.runner.safe_load import load_handoff
    from nsgp_repre_amd.runner.nullspace import full_ignore_keys, should_ignore
    E = np.load(os.path.join(handoff, "expected.npz"))
    w1 = os.path.join(handoff, "work_1")
    cov = load_handoff(os.path.join(w1, "covariance.pth"))
    ignore = full_ignore_keys(I.HANDOFF_IGNORE_KEYS)
    fea_in = {k: v for k, v in cov.items() if not should_ignore(k, ignore)}
    net = I.handoff_net()
    names = [n for n, _ in net.named_parameters()]
    eig, tr = O.get_transforms(names, fea_in, 0.0, "sgd")
    assert sorted(tr) == sorted(n for n in names if f"P__{_key(n)}" in E.files)
    for n in tr:
        np.testing.assert_allclose(eig[n]["eigen_value"].numpy(), E[f"sigma__{_key(n)}"], rtol=1e-5, atol=1e-6 * float(E[f"sigma__{_key(n)}"][0]))
        mask = O.adaptive_threshold(eig[n]["eigen_value"], 0.0, "sgd")
        assert int(mask.to(torch.int8).argmax()) == int(E[f"rank__{_key(n)}"])
        assert np.abs(tr[n].numpy() - E[f"P__{_key(n)}"]).max() <= 1e-5 * np.abs(E[f"P__{_key(n)}"]).max()
    params = [p.detach().clone() for _, p in net.named_parameters()]
    grads = [torch.from_numpy(I.handoff_step_inputs()[n].copy()) for n in names]
    O.sgd_nscl_step(names, params, grads, [dict() for _ in names], tr, **I.G1_HYPER["sgd"])
    for n, p in zip(names, params):
        ref = E[f"p_step0__{_key(n)}"]
        assert np.abs(p.numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), n
    rois = load_handoff(os.path.join(w1, "rois_etc.pth"))
    bank, labels, masks, _ = O.build_bank(rois[0], rois[1], I.HANDOFF_TASK_SPLIT, 2, 10)
    np.testing.assert_array_equal(labels.numpy(), E["labels"])
    assert np.abs(bank.numpy() - E["bank"]).max() <= 1e-6 * np.abs(E["bank"]).max()
    ref_masks = load_handoff(os.path.join(handoff, "work_2", "mask.pth"))
    assert [len(m) for m in masks] == [len(m) for m in ref_masks]
    assert all(torch.equal(a, b) for ml, rl in zip(masks, ref_masks) for a, b in zip(ml, rl))
    terms = load_handoff(os.path.join(w1, "ewc_reg_terms_ewc.pth"))
    theta = {n: torch.from_numpy(v).requires_grad_(True) for n, v in I.handoff_theta().items()}
    loss = O.ewc_loss(theta, terms["importance"], terms["task_param"])
    np.testing.assert_allclose(loss.item(), E["ewc_loss"], rtol=1e-6)


@pytest.mark.gpu
def test_product_task2_start_from_the_references_files(handoff):
    """The product's task-2 start on the HIP path, from the reference-written task-1 directory: checkpoint by keyword,
    ``update_optim_transforms`` (allow-list loader -> eigh -> elbow -> projectors) and one projected step, the
    ``StandardMultiPrototypeReplayHead`` constructor (bank + the mask.pth it writes), ``load_importance`` + ``EWCHook`` -- each
    against what the reference's own task-2 start made of the same files."""
    import nsgp_repre_amd as N
    from nsgp_repre_amd.runner.safe_load import load_handoff
    dev = torch.device("cuda:0")
    E = np.load(os.path.join(handoff, "expected.npz"))
    w1, w2 = os.path.join(handoff, "work_1"), os.path.join(handoff, "work_2")
    ref_masks = load_handoff(os.path.join(w2, "mask.pth"))
    os.remove(os.path.join(w2, "mask.pth"))                  # the product's head has to write its own
    net = I.handoff_net().to(dev)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(1.0)                                      # so that loading the task-1 checkpoint is visible
    opt = N.SGDNSCL(net.parameters(), svd=True, **I.G1_HYPER["sgd"])
    runner = N.runner.BRNullSpaceRunner(net, opt, w2, task_id=2, previous_dir=w1, ckpt_keywords="best", ignore_keys=I.HANDOFF_IGNORE_KEYS,
                                        train_task_split=I.HANDOFF_TASK_SPLIT)
    assert runner.load_previous_checkpoint(net).endswith("best_task1.pth")
    ref_net = I.handoff_net()
    assert all(torch.equal(a.cpu(), b) for (_, a), (_, b) in zip(net.named_parameters(), ref_net.named_parameters()))
    runner.wire_param_names(opt, net)
    runner.update_optim_transforms(opt, net)
    names = [n for n, _ in net.named_parameters()]
    projected = [n for n in names if f"P__{_key(n)}" in E.files]
    assert sorted(opt.transforms) == sorted(projected)
    for n in projected:
        P = opt.transforms[n].cpu().numpy()
        D, r, sig = P.shape[0], int(E[f"rank__{_key(n)}"]), E[f"sigma__{_key(n)}"].astype(np.float64)
        np.testing.assert_allclose(opt.eigens[n]["eigen_value"].cpu().numpy(), sig, rtol=2e-5, atol=2e-6 * sig[0])
        kept = round(float(np.trace(P)) ** 2 / float((P ** 2).sum()))          # rank of a (scaled) projector
        assert D - kept == r, n                                                 # the INTEGER is exact
        # P itself: two fp32 eigensolvers (rocSOLVER syevd here, LAPACK gesdd in the reference) agree on the invariant subspace up
        # to ~eps * sigma_max / gap (Davis-Kahan).  conv1 / layer1 cut at a wide gap: 1e-5.  The neck layer's elbow sits between
        # sigma_16 = 5.84 and sigma_17 = 5.31 under sigma_max = 1783: the SUBSPACE is conditioned to ~4e-4, whichever solver is used
        # (the reference's CPU and GPU paths differ from each other by as much).
        cond = 2.0 ** -23 * sig[0] / (sig[r - 1] - sig[r])
        tol = max(1e-5, 4.0 * cond)
        assert tol <= 1e-5 or n == "neck.conv.weight", (n, tol)
        assert np.abs(P - E[f"P__{_key(n)}"]).max() <= tol * np.abs(E[f"P__{_key(n)}"]).max(), (n, tol)
    # the step itself is held at 1e-5 with the reference's projectors installed (kernel parity is judged given identical P)
    for n in projected:
        opt.transforms[n] = torch.from_numpy(E[f"P__{_key(n)}"]).to(dev)
    grads = I.handoff_step_inputs()
    for n, p in net.named_parameters():
        p.grad = torch.from_numpy(grads[n].copy()).to(dev)
    opt.step()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        ref = E[f"p_step0__{_key(n)}"]
        assert np.abs(p.detach().cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), n
    opt.close()
    # RePRE: the bank from the reference's rois_etc.pth; the mask.pth the product writes equals the reference's
    bbox_head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=64, roi_feat_size=7, num_classes=5,
                                                  task_split=I.HANDOFF_TASK_SPLIT, task_id=2).to(dev)
    head = N.roi_heads.StandardMultiPrototypeReplayHead(bbox_head=bbox_head, previous_path=w1, task_id=2, task_split=I.HANDOFF_TASK_SPLIT,
                                                        max_prototype=10)
    assert head.replay
    np.testing.assert_array_equal(head.tmp_label.cpu().numpy(), E["labels"])
    assert np.abs(head.bbox_featss.cpu().numpy() - E["bank"]).max() <= 1e-5 * np.abs(E["bank"]).max()
    mine = torch.load(os.path.join(w2, "mask.pth"), weights_only=True)
    assert [len(m) for m in mine] == [len(m) for m in ref_masks]
    assert all(torch.equal(a, b) for ml, rl in zip(mine, ref_masks) for a, b in zip(ml, rl))
    assert torch.isfinite(head.add_replay_loss({})["replay_loss_cls"])
    # EWC: the reference's defaultdict file through load_importance + EWCHook
    runner.load_importance(net)
    with torch.no_grad():
        for n, v in I.handoff_theta().items():
            dict(net.named_parameters())[n].copy_(torch.from_numpy(v))
    net.loss = lambda *a, **k: {"loss_cls": torch.ones((), device=dev)}
    hook = runner.wrap_loss_with_ewc(net)
    net.zero_grad(set_to_none=True)
    res = hook()
    np.testing.assert_allclose(res["ewc_loss"].item(), E["ewc_loss"], rtol=1e-6)
    res["ewc_loss"].backward()
    for n in I.handoff_theta():
        gref = E[f"ewc_grad__{_key(n)}"]
        np.testing.assert_allclose(dict(net.named_parameters())[n].grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-6 * np.abs(gref).max())
