"""``BRNullSpaceRunner`` -- task orchestration around the hot path
(mmdet/engine/runner/nsrunner_roi_replay.py:111-1031).

Two faces of one class name:

* with MMEngine installed the registered ``BRNullSpaceRunner`` subclasses ``mmengine.runner.Runner``
  and reads the same top-level config keys as the reference (``task_id, train_task_split, offset,
  ignore_keys, previous_dir, ckpt_keywords, reserve_per_class, is_trained``; runner:288-362,
  416-418) so ``tools/train.py``'s ``RUNNERS.build(cfg)`` picks it up unchanged;
* without MMEngine (this image) it is a small stand-alone driver over plain torch objects with the
  same method names, which is what the tests exercise.

Both delegate every NSGP/RePRE step to ``runner/nullspace.py``; the EWC regulariser lives in
``runner/ewc.py``.
"""
import os
import os.path as osp
from typing import Callable, Iterable, Optional, Sequence

import torch

from ..registry import HAVE_MMENGINE, RUNNERS, register
from . import ewc as EWC
from . import nullspace as NS


class NullSpaceTaskMixin:
    """The fork's additions to the runner, framework-agnostic."""

    def init_task_state(self, work_dir: str, task_id: int = 1, train_task_split: Optional[Sequence[int]] = None,
                        previous_dir: Optional[str] = None, ckpt_keywords: str = "best",
                        ignore_keys: Optional[Sequence[str]] = None, offset: float = 0.0, reserve_per_class: int = 0,
                        is_trained: bool = False, fea_in_load_path: Optional[str] = None,
                        rr_thresh: Optional[Sequence[float]] = None, cov_grouped: bool = True, cov_streams: int = 4):
        self.cov_grouped, self.cov_streams = bool(cov_grouped), int(cov_streams)      # how cal_fea_in issues the covariance launches
        self.task_id = task_id if task_id is not None else 1
        self.task_split = train_task_split
        self.previous_dir = previous_dir if self.task_id != 1 else None
        self.ckpt_keywords = ckpt_keywords
        if self.previous_dir is None or not osp.exists(self.previous_dir):
            assert self.task_id == 1, "Error, previous task dir should be fed into the runner."      # runner:292
        self.fea_in_save_path = osp.join(work_dir, "covariance.pth")                                  # runner:346
        self.fea_in_load_path = fea_in_load_path or (osp.join(self.previous_dir, "covariance.pth")
                                                     if self.previous_dir is not None else None)
        self.ignore_keys = NS.full_ignore_keys(ignore_keys)                                           # runner:355
        self.offset = offset or 0.0
        self.reserve_per_class = reserve_per_class or 0
        self.is_trained = bool(is_trained)
        self._task_work_dir = work_dir
        self.rr_thresh = list(rr_thresh) if rr_thresh else [0.5, 0.5]                               # runner:356
        self.reg_params, self.ewc_reg_terms = {}, {}

    # -- start of task t >= 2 ------------------------------------------------------------------
    def wire_param_names(self, optimizer, model):
        NS.wire_param_names(optimizer, model)

    def update_optim_transforms(self, optimizer, model):
        """runner:635-662 (the duplicate ``update_model_transforms`` :665-692 is folded in)."""
        dev = next(NS.unwrap(model).parameters()).device
        return NS.update_optim_transforms(optimizer, self.fea_in_load_path, self.ignore_keys, self.offset, dev)

    def attach_teacher(self, model):
        """runner:527-547: a frozen deep copy of the (checkpoint-loaded) model answers for the old tasks -- its
        head is switched to task ``t-1`` -- and the RoI head gets a handle on the teacher's RoI head."""
        import copy
        ori = NS.unwrap(model)
        if hasattr(ori, "teacher_model"):
            del ori.teacher_model
        ori.teacher_model = copy.deepcopy(ori)
        ori.teacher_model.roi_head.bbox_head.task_id = self.task_id - 1
        ori.roi_head.teacher_model = ori.teacher_model.roi_head
        for name, param in ori.named_parameters():
            if "teacher" in name:
                param.requires_grad_(False)
        NS.guard_conv_weights(ori)           # the deep copy allocated new weights (stock MIOpen over-read, see the guard)
        return ori.teacher_model

    def set_pseudo_label_thresholds(self, model):
        """runner:439-441: the detector's teacher pseudo-label thresholds come from the config's ``rr_thresh``."""
        ori = NS.unwrap(model)
        ori.rpn_thresh, ori.roi_thresh = self.rr_thresh[0], self.rr_thresh[1]

    # -- EWC on the BatchNorm parameters (runner:558-565, 591, 946-1031) ---------------------------
    def load_importance(self, model):
        """runner:996-999: register the BN parameters and load the previous tasks' ``ewc_reg_terms_ewc.pth``."""
        ori = NS.unwrap(model)
        self.reg_params = EWC.register_params(ori)
        dev = next(ori.parameters()).device
        self.ewc_reg_terms = EWC.load_importance(osp.join(self.previous_dir, "ewc_reg_terms_ewc.pth"), dev)
        return self.ewc_reg_terms

    def wrap_loss_with_ewc(self, model):
        """runner:559-565: ``model.loss`` becomes an ``EWCHook`` that adds ``ewc_loss`` to the loss dict (not for 'joint' runs)."""
        ori = NS.unwrap(model)
        if "joint" in self._task_work_dir or not hasattr(ori, "loss"):
            return None
        ori.loss = EWC.EWCHook(module=ori, reg_params=self.reg_params, ewc_reg_terms=self.ewc_reg_terms)
        return ori.loss

    def calculate_save_importance(self, model, batches, loss_of: Callable, zero_grad: Optional[Callable] = None):
        """runner:946-990: Fisher diagonal of the registered parameters over the train set, in ``eval()`` mode, appended to the
        running lists and written to ``<work_dir>/ewc_reg_terms_ewc.pth``.  ``loss_of(model, batch)`` returns the scalar loss of
        one batch (the reference: ``parse_losses(model._run_forward(data, mode='loss'))``); ``len(batch) / len(batches)`` is the
        reference's weight, taken literally (``len(data_batch)`` of an MMEngine batch dict is its number of keys)."""
        ori = NS.unwrap(model)
        self.reg_params = EWC.register_params(ori)
        importance = {n: p.clone().detach().fill_(0) for n, p in self.reg_params.items()}
        ori.eval()
        batches = list(batches) if not hasattr(batches, "__len__") else batches
        for batch in batches:
            loss = loss_of(model, batch)
            loss.backward()
            EWC.accumulate_importance(importance, self.reg_params, len(batch), len(batches))
            if zero_grad is not None:
                zero_grad()
            else:
                for p in ori.parameters():
                    p.grad = None
        self.ewc_reg_terms = EWC.save_importance(self._task_work_dir, self.ewc_reg_terms, importance, self.reg_params)
        return self.ewc_reg_terms

    # -- checkpoints by keyword (runner:295-299, 710-716, 782-786) -----------------------------------
    def save_checkpoint(self, model, filename: str) -> str:
        """Stand-alone counterpart of MMEngine's CheckpointHook: ``{'state_dict': ...}`` under ``work_dir/filename``."""
        path = osp.join(self._task_work_dir, filename)
        torch.save({"state_dict": {k: v.detach().clone() for k, v in NS.unwrap(model).state_dict().items() if "teacher" not in k}}, path)
        return path

    def _load_state(self, model, path: str) -> str:
        ori = NS.unwrap(model)
        ckpt = torch.load(path, map_location=next(ori.parameters()).device, weights_only=True)
        ori.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt, strict=False)
        NS.guard_conv_weights(ori)
        return path

    def load_previous_checkpoint(self, model) -> Optional[str]:
        """runner:295-299: task t >= 2 starts from the previous task's checkpoint whose name contains ``ckpt_keywords``."""
        if self.previous_dir is None:
            return None
        try:
            return self._load_state(model, NS.find_checkpoint(self.previous_dir, self.ckpt_keywords))
        except FileNotFoundError:
            return None

    def reload_task_checkpoint(self, model) -> Optional[str]:
        """runner:710-716: BEFORE the covariance pass (and so before the RoI dump, runner:782-786, whose own reload is a no-op
        because ``_has_loaded`` is already set) the model goes back to this task's ``ckpt_keywords`` checkpoint -- the files
        ``covariance.pth`` / ``rois_etc.pth`` describe the model the next task will load, not the last iteration's weights."""
        try:
            return self._load_state(model, NS.find_checkpoint(self._task_work_dir, self.ckpt_keywords))
        except FileNotFoundError:
            return None

    # -- end of task t --------------------------------------------------------------------------
    def cal_fea_in(self, model, batches: Iterable, forward: Optional[Callable] = None):
        return NS.cal_fea_in(model, batches, self.ignore_keys, self.fea_in_save_path,
                             self.fea_in_load_path if self.task_id != 1 else None, self.task_id, forward,
                             grouped=self.cov_grouped, n_streams=self.cov_streams)

    def cal_rois(self, model, batches: Iterable, forward: Optional[Callable] = None, num_classes: int = 20):
        prev = osp.join(self.previous_dir, "rois_etc.pth") if (self.task_id != 1 and self.previous_dir) else None
        return NS.cal_rois(model, batches, osp.join(self._task_work_dir, "rois_etc.pth"), prev, self.task_id,
                           self.reserve_per_class, num_classes, forward)


if HAVE_MMENGINE:  # pragma: no cover - exercised only where mmengine is installed
    from mmengine.runner import Runner as _Runner

    @register(RUNNERS)
    class BRNullSpaceRunner(NullSpaceTaskMixin, _Runner):
        @classmethod
        def from_cfg(cls, cfg):
            runner = super().from_cfg(cfg)
            runner.init_task_state(runner.work_dir, cfg.get("task_id"), cfg.get("train_task_split"),
                                   cfg.get("previous_dir"), cfg.get("ckpt_keywords", "best"), cfg.get("ignore_keys"),
                                   cfg.get("offset"), cfg.get("reserve_per_class"), cfg.get("is_trained"),
                                   cfg.get("fea_in_load_path"), cfg.get("rr_thresh"), cfg.get("cov_grouped", True), cfg.get("cov_streams", 4))
            return runner

        def train(self):
            model = NS.unwrap(self.model)
            NS.guard_conv_weights(model)         # stock MIOpen 1x1 backward-data over-read (see the guard); again after loads / teacher copy
            self.set_pseudo_label_thresholds(model)                                                   # runner:439-441
            self._train_loop = self.build_train_loop(self._train_loop)
            self.optim_wrapper = self.build_optim_wrapper(self.optim_wrapper)
            self.wire_param_names(self.optim_wrapper.optimizer, model)                               # runner:473-484
            self.scale_lr(self.optim_wrapper, self.auto_scale_lr)
            if self.param_schedulers is not None:
                self.param_schedulers = self.build_param_scheduler(self.param_schedulers)
            if self._val_loop is not None:
                self._val_loop = self.build_val_loop(self._val_loop)
            self.call_hook("before_run")
            self._init_model_weights()
            self.load_or_resume()
            NS.guard_conv_weights(model)
            if self.task_id != 1 and not self.is_trained:
                if "joint" not in self.work_dir:
                    self.attach_teacher(self.model)
                assert self._resume is False                                                          # runner:551
                NullSpaceTaskMixin.update_optim_transforms(self, self.optim_wrapper.optimizer, model)
                NullSpaceTaskMixin.load_importance(self, self.model)                                  # runner:558
                self.wrap_loss_with_ewc(self.model)                                                   # runner:559-565
            self.optim_wrapper.initialize_count_status(self.model, self._train_loop.iter, self._train_loop.max_iters)
            if not self.is_trained:
                self.train_loop.run()
            self.call_hook("after_run")
            self._has_loaded = False

            def loss_of(net, data_batch):
                ori = NS.unwrap(net)
                data = ori.data_preprocessor(data_batch, True)
                parsed, _ = ori.parse_losses(net._run_forward(data, mode="loss"))
                return self.optim_wrapper.scale_loss(parsed)
            NullSpaceTaskMixin.calculate_save_importance(self, self.model, self.train_dataloader, loss_of,
                                                         self.optim_wrapper.zero_grad)               # runner:591
            # runner:710-716: back to the ckpt_keywords checkpoint of THIS task before the covariance pass
            self._load_from = NS.find_checkpoint(self.work_dir, self.ckpt_keywords)
            self.load_or_resume()

            def fwd_ns(net, data_batch):
                data = net.data_preprocessor(data_batch, True)
                net(data["inputs"], data["data_samples"], mode="nullspace")

            def fwd_roi(net, data_batch):
                data = net.data_preprocessor(data_batch, True)
                return net(data["inputs"], data["data_samples"], mode="roi_replay")
            self.cal_fea_in(self.model, self.train_dataloader, fwd_ns)
            self.cal_rois(self.model, self.train_dataloader, fwd_roi)
            return self.model
else:

    @register(RUNNERS)
    class BRNullSpaceRunner(NullSpaceTaskMixin):
        """Stand-alone driver: ``model`` + NSGP ``optimizer`` + an iterable of batches.

        ``train(step_fn, batches)`` wires the names, builds the projectors for task >= 2, runs
        ``step_fn(model, batch) -> loss`` / ``backward`` / ``optimizer.step`` per batch, then does the
        end-of-task covariance pass and (if a RoI forward is given) the RoI dump."""

        def __init__(self, model, optimizer, work_dir: str, **task_kwargs):
            self.model, self.optimizer, self.work_dir = model, optimizer, work_dir
            self.init_task_state(work_dir, **task_kwargs)

        def train(self, step_fn: Callable, batches: Iterable, cov_forward: Optional[Callable] = None,
                  cov_batches: Optional[Iterable] = None, roi_forward: Optional[Callable] = None,
                  importance_loss: Optional[Callable] = None):
            """The reference's ``train()`` (runner:425-594) in order.  ``step_fn(model, batch) -> loss`` is one training
            forward; ``importance_loss(model, batch) -> loss`` the forward of the EWC importance pass (default: ``step_fn``).
            ``step_fn`` may call ``runner.save_checkpoint(model, 'best_....pth')``; if no file of this task carries
            ``ckpt_keywords`` when the loop ends, the final weights are saved under that keyword."""
            ori = NS.unwrap(self.model)
            # the batches are walked up to three times (training loop, importance pass, covariance pass / RoI dump): a one-shot iterable
            # would leave the later passes empty and the hand-off files silently all-zero
            if not hasattr(batches, "__len__"):
                batches = list(batches)
            if cov_batches is not None and not hasattr(cov_batches, "__len__"):
                cov_batches = list(cov_batches)
            if len(cov_batches if cov_batches is not None else batches) == 0:
                raise ValueError("BRNullSpaceRunner.train: no batches for the importance / covariance passes")
            NS.guard_conv_weights(ori)           # stock MIOpen 1x1 backward-data over-read (see the guard); again after loads / teacher copy
            self.set_pseudo_label_thresholds(self.model)                                              # runner:439-441
            if self.task_id != 1:
                self.load_previous_checkpoint(self.model)                                             # runner:295-299
            if (self.task_id != 1 and not self.is_trained and "joint" not in self.work_dir and hasattr(ori, "roi_head")):
                self.attach_teacher(self.model)
            self.wire_param_names(self.optimizer, self.model)
            if self.task_id != 1 and not self.is_trained:
                self.update_optim_transforms(self.optimizer, self.model)
                if osp.exists(osp.join(self.previous_dir, "ewc_reg_terms_ewc.pth")):
                    self.load_importance(self.model)                                                  # runner:558
                    self.wrap_loss_with_ewc(self.model)                                               # runner:559-565
            if not self.is_trained:
                ori.train()
                for batch in batches:
                    loss = step_fn(self.model, batch)
                    self.optimizer.zero_grad()
                    loss.backward()
                    self.optimizer.step()
                if not any(self.ckpt_keywords in f for f in os.listdir(self.work_dir)):
                    self.save_checkpoint(self.model, f"{self.ckpt_keywords}_final.pth")
            if hasattr(self.optimizer, "close"):
                self.optimizer.close()       # the task's steps are over: release the plans' GPU resources here, not in a finaliser
            cb = cov_batches if cov_batches is not None else batches
            self.calculate_save_importance(self.model, cb, importance_loss or step_fn, self.optimizer.zero_grad)   # runner:591
            self.reload_task_checkpoint(self.model)                                                   # runner:710-716
            cov = self.cal_fea_in(self.model, cov_batches if cov_batches is not None else batches, cov_forward)
            rois = None
            if roi_forward is not None:
                rois = self.cal_rois(self.model, cov_batches if cov_batches is not None else batches, roi_forward)
            return cov, rois
