"""Fused Adam over the flat parameter buffer (one launch per model per step).

Same hyper-parameters and update rule as torch.optim.Adam as used by the reference
(train_nn_patch.py:146-152: betas (0.9, 0.999), eps 1e-8, L2 weight decay folded into the
gradient; train_nn_area.py:149-154: weight_decay 0).  state_dict() has torch.optim.Adam's layout
(per-parameter 'step', 'exp_avg', 'exp_avg_sq'), so the reference's optim_*_latest checkpoints
(train_nn_patch.py:153-156,446-454) load and save unchanged.

capturable=True (torch.optim.Adam's flag of the same name) keeps the step count on the device, so that a whole
training step can be recorded into a hipGraph (qea/graph.py) and replayed."""
import torch

from . import ops
from .params import flat_state_of
from ._lib import QeaError


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = {}
        self.capturable = bool(capturable)
        self._coef = {}

    def _flat_group(self, gi, group):
        """(FlatState, exp_avg flat, exp_avg_sq flat) when the group is exactly one flat model."""
        ps = group["params"]
        fs = flat_state_of(ps[0]) if ps else None
        if fs is None or len(fs.params) != len(ps) or any(a is not b for (_, a), b in zip(fs.params, ps)):
            return None
        ent = self._flat.get(gi)
        if ent is None or ent[0] is not fs:
            m = torch.zeros_like(fs.data)
            v = torch.zeros_like(fs.data)
            for (n, p) in fs.params:                        # adopt existing per-parameter state (resume)
                st = self.state.get(p)
                o = fs.offsets[n]
                mv = m[o:o + p.numel()].as_strided(p.shape, p.stride())
                vv = v[o:o + p.numel()].as_strided(p.shape, p.stride())
                if st:
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                step = st["step"] if st else torch.tensor(0.0)
                self.state[p] = {"step": step if torch.is_tensor(step) else torch.tensor(float(step)), "exp_avg": mv, "exp_avg_sq": vv}
            ent = (fs, m, v)
            self._flat[gi] = ent
        return ent

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = {}                                      # re-adopt the loaded moments on the next step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        # derived weight forms (qea.ops.weight_cached) of the models this optimiser owns are stale from here on
        ops.bump_weight_epoch([p for g in self.param_groups for p in g["params"]])
        for gi, group in enumerate(self.param_groups):
            ps = group["params"]
            if not ps or all(p.grad is None for p in ps):
                continue
            b1, b2 = group["betas"]
            ent = self._flat_group(gi, group) if all(p.grad is not None for p in ps) else None
            if ent is not None:
                fs, m, v = ent
                fs.attach_grads()
                st0 = self.state[ps[0]]
                if self.capturable:
                    step_t = st0["step"]
                    if not (step_t.is_cuda and step_t.dtype == torch.float32 and all(self.state[p]["step"] is step_t for p in ps)):
                        step_t = torch.tensor([float(step_t)], dtype=torch.float32, device=fs.data.device)
                        for p in ps:                         # ONE device counter shared by the parameters of the flat model
                            self.state[p]["step"] = step_t
                    coef = self._coef.get(gi)
                    if coef is None:
                        coef = self._coef[gi] = torch.zeros(2, dtype=torch.float32, device=fs.data.device)
                    ops.adam_step_capturable(fs.data, fs.grad, m, v, fs.total, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                             step_t, coef)
                    continue
                step = int(st0["step"].item()) + 1 if st0["step"].is_cuda else int(st0["step"]) + 1
                ops.adam_step(fs.data, fs.grad, m, v, fs.total, group["lr"], b1, b2, group["eps"], group["weight_decay"], step)
                for p in ps:
                    self.state[p]["step"] = torch.tensor(float(step))
                continue
            # generic path (parameters that are not one flat model): one launch per tensor
            for p in ps:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise QeaError("FusedAdam: CUDA parameters only")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = int(st["step"]) + 1
                st["step"] = torch.tensor(float(step))
                if p.numel() % 4 or p.data_ptr() % 16 or p.grad.data_ptr() % 16 or p.grad.stride() != p.stride():
                    raise QeaError("FusedAdam generic path needs 16-byte aligned, identically laid out tensors with numel % 4 == 0")
                ops.adam_step(p, p.grad, st["exp_avg"], st["exp_avg_sq"], p.numel(), group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], step)
        return loss
