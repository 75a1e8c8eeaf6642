"""TEST INFRASTRUCTURE ONLY — nn.Module shells around the functional CPU oracle, so that host logic
written against `model(x)`, `.parameters()`, `.train()/.eval()`, `.state_dict()` (the trainers) can be
exercised on a machine without a GPU by INJECTING these classes.  The product never imports them."""
import torch.nn as nn

from . import model_oracle as mo


class _Tree(nn.Module):
    """Registers a flat {dotted name: tensor} state as nested parameters/buffers with those names."""

    def _populate(self, state):
        for name, t in state.items():
            node = self
            parts = name.split(".")
            for s in parts[:-1]:
                if not hasattr(node, s):
                    node.add_module(s, _Node())
                node = getattr(node, s)
            if mo.is_buffer(name):
                node.register_buffer(parts[-1], t.clone())
            else:
                node.register_parameter(parts[-1], nn.Parameter(t.clone()))

    def _state(self):
        return dict(self.named_parameters()), dict(self.named_buffers())


class _Node(nn.Module):
    """container; `training` of nodes holding running stats is what set_bn_eval toggles"""


class _BNNode(nn.modules.batchnorm._BatchNorm):
    pass


class OracleUNet(_Tree):
    def __init__(self, in_channels=1, out_channels=1, init_features=32, seed=0):
        super().__init__()
        self._populate(mo.seeded_state(mo.unet_state_shapes(in_channels, out_channels, init_features), seed))

    def forward(self, x, bn_groups=1):
        """bn_groups = N: the SPECIFICATION of the product's fused form — N sequential passes over the N groups (one document
        per call, train_nn_patch.py:318-321), outputs concatenated."""
        import torch
        P, Bf = self._state()
        if bn_groups <= 1 or not self.training:
            return mo.unet_forward(P, Bf, x, training=self.training)
        k = x.shape[0] // bn_groups
        return torch.cat([mo.unet_forward(P, Bf, x[g * k:(g + 1) * k], training=True) for g in range(bn_groups)])


class OracleCRNN(_Tree):
    def __init__(self, vocab_size, multi_gpu=False, seed=1):
        super().__init__()
        self._populate(mo.seeded_state(mo.crnn_state_shapes(vocab_size), seed))
        self._scrub = False
        self._bn_train = True

    def backward_hook(self, module, grad_input, grad_output):
        for g in grad_input:
            g[g != g] = 0

    def register_backward_hook(self, hook):
        self._scrub = True

        class _H:
            def remove(self_inner):
                pass
        return _H()

    def train(self, mode=True):
        self._bn_train = mode
        return super().train(mode)

    def apply(self, fn):
        # utils.set_bn_eval looks for _BatchNorm instances: present two proxies and read their mode back
        proxies = [_BNNode(1), _BNNode(1)]
        for p in proxies:
            p.train(self._bn_train)
            fn(p)
        self._bn_train = all(p.training for p in proxies)
        return self

    def forward(self, x):
        P, Bf = self._state()
        return mo.crnn_forward(P, Bf, x, bn_training=self._bn_train and self.training, nan_scrub=self._scrub)
