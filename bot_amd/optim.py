"""Optimizers of the reference's harness whose update runs as ONE launch over all parameters.

`RMSprop`: torch.optim.RMSprop's update (src/no-sampling/run.py:331-333 constructs it with the defaults: alpha 0.99, eps 1e-8, no
momentum, not centered) through bot_rmsprop_step_f32 — the foreach form is six multi-tensor launches per step.  Same constructor
arguments, same `state_dict` layout ("step", "square_avg"), same arithmetic in the same order; options the kernel does not
implement (momentum, centered, maximize) are refused, not emulated.  Adam / AdamW (configs 1, 3-5) have a one-launch form in torch
itself (`fused=True`), which bot_amd.workloads selects."""
from __future__ import annotations

import torch

from . import _C


class RMSprop(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False, capturable=False):
        if momentum != 0 or centered:
            raise NotImplementedError("bot_amd.optim.RMSprop implements the reference's configuration (momentum 0, not centered)")
        if not (0.0 <= alpha and 0.0 <= eps and 0.0 <= weight_decay):
            raise ValueError("invalid RMSprop hyper-parameter")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=0, centered=False, capturable=capturable))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, sq = [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.is_contiguous():
                    raise NotImplementedError("bot_amd.optim.RMSprop: contiguous parameters only")
                ps.append(p)
                gs.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                sq.append(st["square_avg"])
            lr = group["lr"]
            lr_dev = lr if isinstance(lr, torch.Tensor) else None      # a device scalar: read by the kernel (captured steps)
            if lr_dev is not None and ps and (lr_dev.device != ps[0].device or lr_dev.dtype != torch.float32):
                if lr_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                    raise ValueError("bot_amd.optim.RMSprop: a tensor lr must be float32 on the parameters' device to be captured")
                lr, lr_dev = float(lr_dev), None                       # a CPU / non-fp32 tensor lr: its value, not its address
            _C.rmsprop_step(ps, gs, sq, 0.0 if lr_dev is not None else lr, group["alpha"], group["eps"], group["weight_decay"], lr_dev=lr_dev)
        return loss
