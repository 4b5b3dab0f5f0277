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
        # the full hyper-parameter set of torch.optim.Adam, so that `state_dict()['param_groups']` loads into one
        # (the reference saves / restores optimizer.state_dict(), training_multitask.py:246)
        super().__init__(list(model.parameters()),
                         dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                              capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
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
        self.step_count += 1
        a = self._args()
        L.check(L.load().mtbc_adam_step(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adam")
        return loss

    def _args(self) -> "L.AdamArgs":
        m, g = self.model, self.param_groups[0]
        a = L.AdamArgs()
        a.n, a.p, a.g = m.flat_numel, m.flat_p.data_ptr(), m.flat_g.data_ptr()
        a.m, a.v = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.lr, (a.beta1, a.beta2), a.eps = float(g["lr"]), g["betas"], float(g["eps"])
        a.grad_scale, a.step, a.zero_grad = float(self.grad_scale), self.step_count, 0
        return a

    # ---- the step as two halves, for a training step replayed as a hipGraph (trainer.FusedTrainStep(graph=True)): the three scalars that change from
    #      step to step (grad_scale, lr / (1 - b1^t), 1 / sqrt(1 - b2^t)) travel through 12 bytes of device memory instead of the launch arguments
    @torch.no_grad()
    def advance_dynamic(self) -> None:
        """Count the step and put its scalars where `launch_dynamic`'s kernel reads them -- three fills in stream order (values in the launch arguments of
        torch's fill kernel: no host buffer that a later step could overwrite while a copy is in flight).  NOT captured."""
        self._ensure_state()
        if getattr(self, "_dyn", None) is None or self._dyn.device != self.model.flat_p.device:
            self._dyn = torch.zeros(4, device=self.model.flat_p.device)
        self.step_count += 1
        out = (C.c_float * 3)()
        L.check(L.load().mtbc_adam_dynamic(C.byref(self._args()), C.byref(out)), "adam scalars")
        for i in range(3):
            self._dyn[i:i + 1].fill_(float(out[i]))           # a float32 value passed as a double: exact

    @torch.no_grad()
    def launch_dynamic(self) -> None:
        """The Adam launch itself, reading the scalars `advance_dynamic` left: the same kernel, the same bits as `step`.  Capturable."""
        a = self._args()
        a.step = max(1, a.step)
        a.dynamic = self._dyn.data_ptr()
        L.check(L.load().mtbc_adam_step(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adam")

    def graph_key(self):
        """What a captured launch_dynamic holds by address."""
        return (self.model.flat_p.data_ptr(), self.model.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self._dyn.data_ptr(),
                self.param_groups[0]["betas"], float(self.param_groups[0]["eps"]))

    # ---- checkpoint interchange (training_multitask.py:243-249 saves `optimizer.state_dict()`): the layout is the one
    #      torch.optim.Adam writes -- per-parameter {'step', 'exp_avg', 'exp_avg_sq'} keyed by parameter index -- so a
    #      reference checkpoint resumes here and a checkpoint written here resumes under torch.optim.Adam.
    def state_dict(self):
        sd = super().state_dict()
        if self.exp_avg is not None and self.step_count > 0:
            m = self.model
            state = {}
            for i, name in enumerate(n for n, _ in m.named_parameters()):
                s = m.slots[name]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[s.offset:s.offset + s.numel].view(s.shape).detach().clone(),
                            "exp_avg_sq": self.exp_avg_sq[s.offset:s.offset + s.numel].view(s.shape).detach().clone()}
            sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        legacy = sd.pop("fused", None)                 # round-1 private format
        state = sd.get("state", {}) or {}
        super().load_state_dict({"state": {}, "param_groups": sd["param_groups"]})
        m = self.model
        if state:
            self._ensure_state()
            steps = set()
            for i, name in enumerate(n for n, _ in m.named_parameters()):
                st = state.get(i, state.get(str(i)))
                if st is None:
                    continue
                s = m.slots[name]
                self.exp_avg[s.offset:s.offset + s.numel].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[s.offset:s.offset + s.numel].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not an Adam state this optimizer can hold")
            self.step_count = steps.pop() if steps else 0
        elif legacy is not None:
            self.step_count = int(legacy["step"])
            if legacy["exp_avg"] is not None:
                self._ensure_state()
                self.exp_avg.copy_(legacy["exp_avg"])
                self.exp_avg_sq.copy_(legacy["exp_avg_sq"])
