"""CRNN proxy of the black-box OCR — MI355X-native drop-in for the reference's models/model_crnn.py.

Same constructor (`CRNN(vocab_size, multi_gpu=True)`), same sub-module tree (lstm, linear, convo
with conv1..conv7 / batchnorm1,2) and state_dict keys as reference models/model_crnn.py:5-45, and
the same `backward_hook` / `register_backward_hook` protocol the trainers use to scrub NaN
gradients of infeasible CTC targets (model_crnn.py:30-32, train_nn_patch.py:94).  The torch.nn
sub-modules only hold parameters; forward() runs the HIP schedule of qea/crnn_engine.py.
"""
import torch
import torch.nn as nn

from qea._lib import QeaError
from qea.autograd import CRNNFn, _require_cuda
from qea.crnn_engine import CRNNEngine
from qea.params import ensure_flat


class Convolutional(nn.Module):
    """VGG-style backbone container (reference model_crnn.py:34-45); parameters only."""

    def __init__(self):
        super().__init__()
        plan = ((1, 1, 64, 3, 1), (2, 64, 128, 3, 1), (3, 128, 256, 3, 1), (4, 256, 256, 3, 1), (5, 256, 512, 3, 1))
        for i, cin, cout, k, p in plan:
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=p))
        self.batchnorm1 = nn.BatchNorm2d(512)
        self.conv6 = nn.Conv2d(512, 512, kernel_size=3, stride=1, padding=1)
        self.batchnorm2 = nn.BatchNorm2d(512)
        self.conv7 = nn.Conv2d(512, 512, kernel_size=2, stride=1, padding=0)

    def forward(self, x):
        raise QeaError("Convolutional is a parameter container; call CRNN.forward (HIP path)")


class _NullHandle:
    def remove(self):
        pass


class CRNN(nn.Module):
    def __init__(self, vocab_size, multi_gpu=True):
        super().__init__()
        self.lstm = nn.LSTM(512, 256, 2, bidirectional=True)
        self.linear = nn.Linear(512, vocab_size)
        if multi_gpu:
            # kept only so that state_dict keys match what the reference would write with this flag
            # (`convo.module.*`); data parallelism here is one process per GPU (qea/dist.py)
            self.convo = nn.DataParallel(Convolutional())
        else:
            self.convo = Convolutional()

    # ---- reference protocol for the NaN scrub ----
    def backward_hook(self, module, grad_input, grad_output):
        for g in grad_input:
            if g is not None:
                g[g != g] = 0

    def register_backward_hook(self, hook):
        if getattr(hook, "__func__", None) is CRNN.backward_hook:
            self.__dict__["_qea_nan_scrub"] = True           # done inside the fused log_softmax backward kernel
            return _NullHandle()
        raise QeaError("CRNN (HIP path) supports only its own backward_hook (NaN scrub) as a backward hook")

    # ---- HIP path ----
    def _engine(self):
        eng = self.__dict__.get("_qea_engine")
        if eng is None:
            eng = CRNNEngine(self, self.linear.out_features)
            self.__dict__["_qea_engine"] = eng
        return eng

    def _bn_mode(self):
        modes = {m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)}
        if len(modes) != 1:
            raise QeaError("CRNN: batchnorm1/batchnorm2 must be in the same mode")
        return modes.pop()

    def forward(self, x, replica_groups=1, backward_group=None, group_sizes=None):
        """x [B,1,32,W] -> log-probs [T,B,vocab].  replica_groups = R (new, additive): x holds R jitter replicas
        of the same strips stacked replica-major; batch-stat BatchNorm runs per replica group, so one call equals R
        sequential calls of the reference on the R replicas (see CRNNEngine.forward).
        backward_group = g (new, additive): only replica group g will be back-propagated (the area flow keeps the last
        replica's loss only, train_nn_area.py:269-271).  The other groups' log-probs are returned DETACHED, so no gradient
        can reach them, and the backward pass runs on that group's B/R samples instead of on all B.
        group_sizes = [n_0, n_1, ...] (new, additive; round 4): RAGGED BatchNorm groups — the batch is the concatenation of groups of n_i
        samples (e.g. the jittered strips of several documents, document-major, replica-minor); batch-stat BatchNorm runs per group and
        the running statistics are updated once per group in order = the reference's sequential calls (train_nn_patch.py:288-303)."""
        _require_cuda(x, "CRNN")
        eng = self._engine()
        ensure_flat(self)
        anchor = self.__dict__.get("_qea_anchor")
        if anchor is None or anchor.device != x.device:
            anchor = torch.zeros((), device=x.device, requires_grad=True)
            self.__dict__["_qea_anchor"] = anchor
        wants = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        skip = self.__dict__.get("_qea_skip_param_grads", False)
        R = int(replica_groups)
        if group_sizes is not None:
            if R != 1 or backward_group is not None:
                raise ValueError("group_sizes excludes replica_groups / backward_group")
            return CRNNFn.apply(x, anchor if wants else None, eng, self._bn_mode(), bool(self.__dict__.get("_qea_nan_scrub", False)), not skip,
                                tuple(int(v) for v in group_sizes), -1)
        g = -1 if backward_group is None or R <= 1 else int(backward_group)
        if g >= R:
            raise ValueError(f"backward_group {g} out of range for {R} replica groups")
        out = CRNNFn.apply(x, anchor if wants else None, eng, self._bn_mode(), bool(self.__dict__.get("_qea_nan_scrub", False)), not skip, R, g)
        if g >= 0 and out.requires_grad:
            k = x.shape[0] // R
            out = torch.cat([out[:, :g * k].detach(), out[:, g * k:(g + 1) * k], out[:, (g + 1) * k:].detach()], dim=1)
        return out

    def map_to_sequence(self, map):
        batch, channel, height, width = map.size()
        return map.permute(3, 0, 1, 2).reshape(width, batch, channel * height)

    def zero_grad(self, set_to_none=True):
        fs = self.__dict__.get("_qea_flat_state")
        if fs is not None and fs.intact():
            fs.zero_grad()
        else:
            super().zero_grad(set_to_none)

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_qea")}

    def __setstate__(self, state):
        """A whole-module pickle written by the REFERENCE after `register_backward_hook(crnn.backward_hook)`
        (train_nn_patch.py:93-94,440-454) carries a live legacy hook in `_backward_hooks`.  nn.Module would attach it to the
        HIP path's autograd node, whose grad_inputs include None entries; the scrub it stands for runs inside the fused
        log_softmax backward kernel here, so the hook is taken out and the flag set instead."""
        super().__setstate__(state)
        hooks = self.__dict__.get("_backward_hooks")
        if hooks:
            for key, h in list(hooks.items()):
                if getattr(h, "__name__", "") == "backward_hook":
                    del hooks[key]
                    self.__dict__["_qea_nan_scrub"] = True
            if not hooks:
                self.__dict__["_is_full_backward_hook"] = None
