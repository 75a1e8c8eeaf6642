"""torch.autograd bridges: one node per network / loss, so `loss.backward()` in the trainers
drives the hand-written HIP backward schedules.  Parameter gradients are accumulated in place
into the flat gradient buffer (qea/params.py); the Functions return gradients only for tensor
inputs (the image fed to the CRNN, the log-probs fed to CTC)."""
import torch

from . import ops
from ._lib import QeaError


def _require_cuda(t, who):
    if not t.is_cuda:
        raise QeaError(f"{who}: the HIP path needs CUDA (ROCm) tensors; there is no CPU implementation in this package")
    if t.dtype != torch.float32:
        raise QeaError(f"{who}: fp32 tensors only, got {t.dtype}")


class UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, engine, training, groups=1):
        out, saved = engine.forward(x.contiguous(), training, need_grad=bool(ctx.needs_input_grad[1]), groups=groups)
        ctx.engine, ctx.saved = engine, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise QeaError("UNet backward called without saved activations")
        ctx.engine.backward(ctx.saved, dout)
        ctx.saved = None
        return None, None, None, None, None


class CRNNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, engine, bn_training, nan_scrub, param_grads, groups=1, grad_group=-1):
        need = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        out, saved = engine.forward(x.contiguous(), bn_training, need_grad=need, groups=groups)
        if saved is not None:
            saved["grad_group"] = int(grad_group)
        ctx.engine, ctx.saved, ctx.nan_scrub, ctx.param_grads = engine, saved, nan_scrub, param_grads
        ctx.need_dx = bool(ctx.needs_input_grad[0])
        return out

    @staticmethod
    def backward(ctx, dlp):
        if ctx.saved is None:
            raise QeaError("CRNN backward called without saved activations")
        dx = ctx.engine.backward(ctx.saved, dlp, ctx.nan_scrub, ctx.need_dx, ctx.param_grads)
        ctx.saved = None
        return dx, None, None, None, None, None, None, None


class CTCFn(torch.autograd.Function):
    """reduction: 1 = mean (torch.nn.CTCLoss default), 0 = none (per-sample vector)."""

    @staticmethod
    def forward(ctx, lp, targets, offsets, in_len, tg_len, S_max, reduction, blank):
        T, N, C = lp.shape
        if lp.stride(2) != 1:
            lp = lp.contiguous()
        dev = lp.device
        nll = torch.empty(N, device=dev)
        loss = torch.empty(1, device=dev)
        need = ctx.needs_input_grad[0]
        grad = torch.empty(T, N, C, device=dev) if need else None
        ops.ctc_loss(lp, lp.stride(0), lp.stride(1), targets, offsets, in_len, tg_len, T, N, C, blank, S_max, reduction, 1.0, nll, loss,
                     grad, N * C, C)
        ctx.grad, ctx.reduction = grad, reduction
        return loss[0] if reduction == 1 else nll

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        ctx.grad = None
        if ctx.reduction == 1:
            g = g * gout
        else:
            g = g * gout.view(1, -1, 1)
        return g, None, None, None, None, None, None, None
