"""Fused Adam over the model's flat parameter buffer (one launch for all ~15 M parameters).

Replaces torch.optim.Adam(model.parameters(), lr, eps=1e-4) of src/utils/experiment_init.py:186-187 and its
`.step()` at training_multitask.py:103.  Subclasses torch.optim.Optimizer so ReduceLROnPlateau / CosineAnnealingLR
(experiment_init.py:275-278) and `.param_groups[0]['lr']`, `.zero_grad(set_to_none=True)`, `.state_dict()` work.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.grad_scale = 1.0            # set to 1/world_size by the data-parallel trainer

    def _ensure_state(self) -> None:
        m = self.model
        m.ensure_flat()
        if self.exp_avg is None or self.exp_avg.device != m.flat_p.device or self.exp_avg.numel() != m.flat_numel:
            self.exp_avg = torch.zeros_like(m.flat_p)
            self.exp_avg_sq = torch.zeros_like(m.flat_p)

    @torch.no_grad()
    def step(self, closure=None, grads_in_flat: bool = False):
        loss = closure() if closure is not None else None
        self._ensure_state()
        m = self.model
        if not grads_in_flat:
            # drop-in path: autograd may have stored p.grad outside the flat buffer -> gather
            params = dict(m.named_parameters())
            for name in m._order:
                p, slot = params[name], m._grad_view(name)
                if p.grad is None:
                    slot.zero_()
                elif p.grad.data_ptr() != slot.data_ptr():
                    slot.copy_(p.grad)
        g = self.param_groups[0]
        self.step_count += 1
        a = L.AdamArgs()
        a.n, a.p, a.g = m.flat_numel, m.flat_p.data_ptr(), m.flat_g.data_ptr()
        a.m, a.v = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.lr, (a.beta1, a.beta2), a.eps = float(g["lr"]), g["betas"], float(g["eps"])
        a.grad_scale, a.step, a.zero_grad = float(self.grad_scale), self.step_count, 0
        L.check(L.load().mtbc_adam_step(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adam")
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"step": self.step_count,
                       "exp_avg": None if self.exp_avg is None else self.exp_avg.detach().cpu(),
                       "exp_avg_sq": None if self.exp_avg_sq is None else self.exp_avg_sq.detach().cpu()}
        return sd

    def load_state_dict(self, sd):
        fused = sd.pop("fused", None) if isinstance(sd, dict) else None
        super().load_state_dict(sd)
        if fused is not None:
            self.step_count = int(fused["step"])
            if fused["exp_avg"] is not None:
                self._ensure_state()
                self.exp_avg.copy_(fused["exp_avg"])
                self.exp_avg_sq.copy_(fused["exp_avg_sq"])
