"""Flat HBM storage for a module's parameters, gradients and BatchNorm buffers.

All fp32 parameters of a model live back to back in ONE device buffer (each at a 64-float
aligned offset) and so do their gradients: the optimiser is a single fused Adam launch and the
data-parallel gradient exchange a single RCCL all-reduce.  Parameters keep
their checkpoint shapes: 4-D filters stay [O,I,kh,kw] tensors but in torch.channels_last
strides, i.e. physically [O][kh][kw][I] — the K-contiguous layout the implicit-GEMM kernels
read — so state_dict()/torch.save()/load_state_dict() of the reference's checkpoints work
unchanged.
"""
import weakref

import torch

ALIGN = 64  # floats

_REGISTRY = {}  # id(parameter) -> weakref to its FlatState (no attributes are hung on tensors:
                # torch pickles a Parameter's __dict__, and checkpoints are whole-module pickles)


def flat_state_of(param, check=True):
    """check=False skips the walk over all parameters (FlatState.intact): for callers that run behind the engines' ensure_flat(), which
    has just made that check (qea.ops.filter_absmax is called once per filter and optimiser step: 46 walks over 46 parameters were
    0.7 ms of host time per step, visible in the launch-bound single-document pass)."""
    ref = _REGISTRY.get(id(param))
    fs = ref() if ref is not None else None
    if fs is not None and fs.by_id.get(id(param)) is param and (not check or fs.intact()):
        return fs
    return None


def _dense_strides(t):
    """Strides to keep when re-homing `t` (channels_last for 4-D filters, contiguous otherwise)."""
    if t.dim() == 4:
        return torch.empty(t.shape, device="meta").to(memory_format=torch.channels_last).stride()
    return torch.empty(t.shape, device="meta").stride()


class FlatState:
    def __init__(self, module):
        self.params = [(n, p) for n, p in module.named_parameters()]
        dev = self.params[0][1].device
        self.device = dev
        self.offsets = {}
        off = 0
        for n, p in self.params:
            self.offsets[n] = off
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views = []
        for n, p in self.params:
            o = self.offsets[n]
            st = _dense_strides(p)
            view = self.data[o:o + p.numel()].as_strided(p.shape, st)
            view.copy_(p.data)
            p.data = view
            gview = self.grad[o:o + p.numel()].as_strided(p.shape, st)
            if p.grad is not None:
                gview.copy_(p.grad)
            p.grad = gview
            self.grad_views.append(gview)
        self.by_id = {id(p): p for _, p in self.params}
        self.epoch = 0                                       # bumped by writers that touch only this model's parameters (qea.ops.bump_weight_epoch)
        for _, p in self.params:
            _REGISTRY[id(p)] = weakref.ref(self)
        # BatchNorm buffers: running stats in one fp32 buffer, step counters in one int64 buffer
        fbufs = [(n, b) for n, b in module.named_buffers() if b.dtype == torch.float32]
        ibufs = [(n, b) for n, b in module.named_buffers() if b.dtype == torch.int64]
        self.fbuf = torch.zeros(sum((b.numel() + ALIGN - 1) // ALIGN * ALIGN for _, b in fbufs) or 1, dtype=torch.float32, device=dev)
        off = 0
        for n, b in fbufs:
            v = self.fbuf[off:off + b.numel()].view(b.shape)
            v.copy_(b)
            _set_buffer(module, n, v)
            off += (b.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.ibuf = torch.zeros(max(len(ibufs), 1), dtype=torch.int64, device=dev)
        for i, (n, b) in enumerate(ibufs):
            v = self.ibuf[i:i + 1].view(b.shape)
            v.copy_(b)
            _set_buffer(module, n, v)
        self._ptrs = [p.data_ptr() for _, p in self.params]
        self.vsum = sum(p._version for _, p in self.params)

    def intact(self):
        """False if something (e.g. module.to(), load of a pickled checkpoint) re-homed a parameter.
        The same walk records `vsum`, the sum of the parameters' version counters: the parameters are views re-homed with
        `p.data = view`, so a torch-side write to ONE of them (load_state_dict, p.copy_, a torch.optim step) moves that parameter's
        _version and not the flat buffer's — anything derived from the WHOLE buffer (qea.ops.filter_absmax) is keyed on vsum."""
        ps = self.params
        ok = True
        vs = 0
        for (_, p), q in zip(ps, self._ptrs):
            ok = ok and p.data_ptr() == q
            vs += p._version
        self.vsum = vs
        return ok

    def attach_grads(self):
        """Make every p.grad the view into the flat gradient buffer again.  A gradient that was
        set to None (torch's zero_grad default) means zero: its slice is cleared."""
        for (_, p), gv in zip(self.params, self.grad_views):
            g = p.grad
            if g is None:
                gv.zero_()
                p.grad = gv
            elif g.data_ptr() != gv.data_ptr():
                gv.copy_(g)
                p.grad = gv

    def zero_grad(self):
        self.grad.zero_()
        for (_, p), gv in zip(self.params, self.grad_views):
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv


def _set_buffer(module, dotted, value):
    mod = module
    parts = dotted.split(".")
    for s in parts[:-1]:
        mod = getattr(mod, s)
    mod._buffers[parts[-1]] = value


def ensure_flat(module):
    """Return the module's FlatState, (re)building it when parameters were moved or replaced."""
    fs = getattr(module, "_qea_flat_state", None)
    if fs is None or not fs.intact() or fs.params[0][1].device != fs.device:
        fs = FlatState(module)
        object.__setattr__(module, "_qea_flat_state", fs)
        from . import ops
        ops.bump_weight_epoch()                              # the parameters were re-homed through p.data
    return fs
