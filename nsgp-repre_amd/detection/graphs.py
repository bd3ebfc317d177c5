"""hipGraph capture of the static-shape part of the detector step.

At one image per GPU the bf16 training step is host-launch-bound (~2 700 kernel launches, 26 ms of kernel time
in a 44 ms step): backbone + FPN + the RPN head's convolutions always see the same shapes, so their forward and
backward are captured once and replayed -- ``torch.cuda.make_graphed_callables`` for the student (one autograd
node whose backward is a second graph), a plain ``torch.cuda.CUDAGraph`` for the frozen teacher.  Everything
data-dependent (anchor targets, proposals + NMS, RoI sampling, RoIAlign, the heads, the losses) and the NSGP step
stay eager.  Replay runs no Python, so module hooks do not fire inside a graph: the covariance pass
(``cal_fea_in``, eval mode, hooks registered) always takes the eager path.
"""
import torch
import torch.nn as nn


class ConvTrunk(nn.Module):
    """image -> (5 FPN levels, 5 RPN objectness maps, 5 RPN delta maps).  Holds the detector's OWN sub-modules but
    is never attached to the detector, so ``named_parameters()`` of the model is unchanged."""

    def __init__(self, backbone, neck, rpn_head):
        super().__init__()
        self.backbone, self.neck, self.rpn_head = backbone, neck, rpn_head

    def forward(self, x):
        feats = self.neck(self.backbone(x))
        cls, reg = self.rpn_head(feats)
        return tuple(feats) + tuple(cls) + tuple(reg)


def _split(outs):
    n = len(outs) // 3
    return tuple(outs[:n]), (list(outs[n:2 * n]), list(outs[2 * n:]))


class GraphedTrunk:
    """``run(x) -> (feats, (cls, reg))`` through a captured graph when ``x`` has the captured shape."""

    def __init__(self, model, sample: torch.Tensor, train: bool, amp_dtype=None):
        self.shape, self.train = tuple(sample.shape), train
        self.amp_dtype = amp_dtype
        trunk = ConvTrunk(model.backbone, model.neck, model.rpn_head)
        self._trunk = trunk
        if train:
            trunk.train()
            with self._autocast():
                self.fn = torch.cuda.make_graphed_callables(trunk, (sample.detach().clone(),), num_warmup_iters=3)
        else:
            trunk.eval()
            self.static_in = sample.detach().clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad(), self._autocast():
                for _ in range(3):
                    trunk(self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), self._autocast(), torch.cuda.graph(self.graph):
                self.static_out = trunk(self.static_in)

    def _autocast(self):
        return torch.autocast("cuda", dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None, cache_enabled=False)

    def close(self):
        """Drop the captured graphs NOW.  ``make_graphed_callables`` leaves a reference cycle (module -> patched forward
        -> module), so without this the graphs and their private memory pool are torn down whenever the garbage collector
        happens to run -- possibly on another thread in the middle of unrelated GPU work."""
        trunk = getattr(self, "_trunk", None)
        if trunk is not None and "forward" in trunk.__dict__:
            del trunk.__dict__["forward"]                     # restores the class's forward, breaks the cycle
        for name in ("fn", "graph", "static_in", "static_out", "_trunk"):
            if hasattr(self, name):
                delattr(self, name)
        torch.cuda.synchronize()

    def matches(self, x) -> bool:
        return tuple(x.shape) == self.shape and x.is_cuda

    def run(self, x):
        if self.train:
            with self._autocast():
                return _split(self.fn(x))
        self.static_in.copy_(x)
        self.graph.replay()
        return _split(self.static_out)        # valid until the next replay (consumed within the step)
