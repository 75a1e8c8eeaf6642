"""UNet document cleaner — MI355X-native drop-in for the reference's models/model_unet.py.

Same constructor, same sub-module tree and therefore the same state_dict keys / pickle layout
as reference models/model_unet.py:7-46 (encoder{1-4}, pool{1-4}, bottleneck, upconv{4-1},
decoder{4-1}, conv; block members `<name>conv{1,2}`, `<name>norm{1,2}`, `<name>relu{1,2}`), so
whole-module checkpoints written by either side load on the other.  The torch.nn
sub-modules are used as parameter containers only: forward() runs the HIP kernel schedule of
qea/unet_engine.py and there is no CPU implementation here.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from qea._lib import QeaError
from qea.autograd import UNetFn, _require_cuda
from qea.params import ensure_flat
from qea.unet_engine import UNetEngine


def _double_conv(cin, cout, tag):
    layers = OrderedDict()
    for i, ci in ((1, cin), (2, cout)):
        layers[f"{tag}conv{i}"] = nn.Conv2d(ci, cout, kernel_size=3, padding=1, bias=False)
        layers[f"{tag}norm{i}"] = nn.BatchNorm2d(cout)
        layers[f"{tag}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(layers)


class UNet(nn.Module):
    def __init__(self, in_channels=1, out_channels=1, init_features=32):
        super().__init__()
        if in_channels != 1 or out_channels != 1:
            raise QeaError("the HIP UNet covers the reference's grey-scale configuration (in_channels = out_channels = 1)")
        f = init_features
        widths = [f, 2 * f, 4 * f, 8 * f]
        cin = in_channels
        for lvl, c in enumerate(widths, start=1):
            setattr(self, f"encoder{lvl}", _double_conv(cin, c, f"enc{lvl}"))
            setattr(self, f"pool{lvl}", nn.MaxPool2d(kernel_size=2, stride=2))
            cin = c
        self.bottleneck = _double_conv(cin, 16 * f, "bottleneck")
        cin = 16 * f
        for lvl in (4, 3, 2, 1):
            c = widths[lvl - 1]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose2d(cin, c, kernel_size=2, stride=2))
            setattr(self, f"decoder{lvl}", _double_conv(2 * c, c, f"dec{lvl}"))
            cin = c
        self.conv = nn.Conv2d(f, out_channels, kernel_size=1)

    # ---- HIP path ----
    def _engine(self):
        eng = self.__dict__.get("_qea_engine")
        if eng is None:
            f = self.conv.weight.shape[1]
            eng = UNetEngine(self, features=f)
            self.__dict__["_qea_engine"] = eng
            self.__dict__["_qea_anchor"] = None
        return eng

    def _bn_mode(self):
        modes = {m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)}
        if len(modes) != 1:
            raise QeaError("UNet: mixed BatchNorm train/eval modes are not supported")
        return modes.pop()

    def forward(self, x, bn_groups=1):
        """x [B,1,H,W] -> [B,1,H,W].  bn_groups = N (new, additive; train mode only): x holds N equal groups of images stacked
        along the batch (N documents of the patch flow); batch-statistic BatchNorm runs per group and the running statistics
        are updated once per group in order, so one call equals N sequential calls of the reference on the N groups."""
        _require_cuda(x, "UNet")
        eng = self._engine()
        ensure_flat(self)
        anchor = self.__dict__.get("_qea_anchor")
        if anchor is None or anchor.device != x.device:
            anchor = torch.zeros((), device=x.device, requires_grad=True)
            self.__dict__["_qea_anchor"] = anchor
        wants = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return UNetFn.apply(x, anchor if wants else None, eng, self._bn_mode(), int(bn_groups))

    def zero_grad(self, set_to_none=True):
        fs = self.__dict__.get("_qea_flat_state")
        if fs is not None and fs.intact():
            fs.zero_grad()                                   # one memset; gradients stay views of the flat buffer
        else:
            super().zero_grad(set_to_none)

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_qea")}
