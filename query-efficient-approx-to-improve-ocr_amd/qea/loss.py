"""CTC loss module with torch.nn.CTCLoss's call signature, computed by the HIP kernels.

Drop-in for the reference's `CTCLoss()` / `CTCLoss(reduction="none")` (train_nn_patch.py:143,
train_nn_area.py:146-147): log_probs [T,N,C] on the GPU, targets / input_lengths /
target_lengths as the reference builds them — 1-D int32 CPU tensors (train_nn_patch.py:170-174).
blank = 0, zero_infinity = False: an infeasible target gives loss = inf and NaN gradient rows,
which CRNN's NaN scrub then zeroes.

Graph-capturable form: hand over targets / lengths as CUDA int32 tensors and set `max_target_length` on the module
(nothing on the host then depends on their values, and nothing is copied from pageable memory)."""
import torch

from .autograd import CTCFn, _require_cuda
from ._lib import QeaError


class CTCLoss(torch.nn.Module):
    def __init__(self, blank=0, reduction="mean", zero_infinity=False):
        super().__init__()
        if reduction not in ("mean", "none"):
            raise QeaError("CTCLoss: reduction must be 'mean' or 'none'")
        if zero_infinity:
            raise QeaError("CTCLoss: zero_infinity=True is not on the reference path")
        self.blank, self.reduction = blank, reduction
        self.max_target_length = None          # required when the targets arrive as CUDA tensors

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        _require_cuda(log_probs, "CTCLoss")
        dev = log_probs.device
        if torch.is_tensor(targets) and targets.is_cuda:
            return self._forward_device(log_probs, targets, input_lengths, target_lengths)
        tl_cpu = torch.as_tensor(target_lengths).to("cpu", torch.int64)
        N = log_probs.shape[1]
        if tl_cpu.numel() != N:
            raise QeaError(f"CTCLoss: {tl_cpu.numel()} target lengths for batch {N}")
        max_len = int(tl_cpu.max().item()) if N else 0
        S_max = 2 * max(max_len, 1) + 1
        if S_max > 256:
            raise QeaError(f"CTCLoss: target length {max_len} > 127 not supported")
        if targets.dim() != 1:
            raise QeaError("CTCLoss: concatenated 1-D targets expected (as the reference builds them)")
        offs = torch.zeros(N, dtype=torch.int64)
        if N > 1:
            offs[1:] = torch.cumsum(tl_cpu, 0)[:-1]
        tg = targets.to(dev, torch.int32, non_blocking=True)
        il = torch.as_tensor(input_lengths).to(dev, torch.int32, non_blocking=True)
        tl = tl_cpu.to(dev, torch.int32, non_blocking=True)
        if tg.numel() == 0:
            tg = torch.zeros(1, dtype=torch.int32, device=dev)
        return CTCFn.apply(log_probs, tg, offs.to(dev, non_blocking=True), il, tl, S_max, 1 if self.reduction == "mean" else 0, self.blank)

    def _forward_device(self, log_probs, targets, input_lengths, target_lengths):
        if self.max_target_length is None:
            raise QeaError("CTCLoss: set .max_target_length when targets are CUDA tensors (their values are not read on the host)")
        if not (torch.is_tensor(input_lengths) and input_lengths.is_cuda and torch.is_tensor(target_lengths) and target_lengths.is_cuda):
            raise QeaError("CTCLoss: with CUDA targets, input_lengths and target_lengths must be CUDA tensors too")
        N = log_probs.shape[1]
        if target_lengths.numel() != N or targets.dim() != 1:
            raise QeaError("CTCLoss: one target length per batch member and concatenated 1-D targets expected")
        S_max = 2 * max(int(self.max_target_length), 1) + 1
        if S_max > 256:
            raise QeaError(f"CTCLoss: target length {self.max_target_length} > 127 not supported")
        tl = target_lengths.to(torch.int32)
        offs = torch.cumsum(tl, 0, dtype=torch.int64) - tl
        return CTCFn.apply(log_probs, targets.to(torch.int32), offs, input_lengths.to(torch.int32), tl, S_max,
                           1 if self.reduction == "mean" else 0, self.blank)
