"""Shared helpers for the parity tests (synthetic inputs identical to tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

CHAR_SET = ['`', ' ', '!', '"', '#', '$', '%', '&', "'", '(', ')', '*', '+', ',', '-', '.'] + list("0123456789") + \
    [':', ';', '<', '=', '>', '?', '@'] + [chr(c) for c in range(ord('A'), ord('Z') + 1)] + ['[', ']', '^'] + \
    [chr(c) for c in range(ord('a'), ord('z') + 1)] + ['{', '|', '~', '€', '}', '\\', '/']
assert len(CHAR_SET) == 95
C2I = {c: i for i, c in enumerate(CHAR_SET)}
I2C = {i: c for i, c in enumerate(CHAR_SET)}


def golden(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return np.load(path, allow_pickle=False)


def synth_images(b, seed, h=32, w=128):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(b, 1, h, w, generator=g) < 0.12).float()
    ink = torch.rand(b, 1, h, w, generator=g) * 0.7 + 0.3
    return (1 - m * ink + 0.02 * torch.randn(b, 1, h, w, generator=g)).clamp(0, 1)


def synth_labels(b, seed, lo=1, hi=12):
    rng = np.random.RandomState(seed)
    return ["".join(CHAR_SET[i] for i in rng.randint(1, 95, rng.randint(lo, hi + 1))) for _ in range(b)]


def encode(labels):
    y = torch.tensor([C2I[c] for c in "".join(labels)], dtype=torch.int)
    return y, torch.tensor([len(l) for l in labels], dtype=torch.int)


def check_grad_summary(fix, prefix, named_grads, rtol=2e-4, atol_frac=2e-5, atol=1e-7):
    """Compare gradients with the |sum/|abs/|l2/|head summaries stored by make_golden.grad_summary.
    The sum of n terms is compared with a tolerance relative to the abs-sum (its condition);
    `atol` is a per-element noise floor (e.g. a conv bias in front of a batch-stat BN has an
    exactly-zero gradient that both sides compute as 1e-10 rounding noise)."""
    bad = []
    for name, g in named_grads:
        g = g.detach().double().flatten().cpu()
        ref_abs = float(fix[f"{prefix}{name}|abs"])
        ref_l2 = float(fix[f"{prefix}{name}|l2"])
        tol = rtol * ref_abs + atol * g.numel()
        if abs(g.abs().sum().item() - ref_abs) > tol:
            bad.append((name, "abs", g.abs().sum().item(), ref_abs))
        if abs(g.sum().item() - float(fix[f"{prefix}{name}|sum"])) > tol:
            bad.append((name, "sum", g.sum().item(), float(fix[f"{prefix}{name}|sum"])))
        if abs(g.norm().item() - ref_l2) > rtol * ref_l2 + atol * g.numel() ** 0.5:
            bad.append((name, "l2", g.norm().item(), ref_l2))
        head = torch.from_numpy(fix[f"{prefix}{name}|head"])
        n = head.numel()
        scale = ref_l2 / max(1.0, g.numel() ** 0.5)
        if (g[:n] - head).abs().max().item() > rtol * head.abs().max().item() + atol_frac * scale + atol:
            bad.append((name, "head", g[:n].tolist()[:4], head.tolist()[:4]))
    assert not bad, bad[:6]


def check_tensor_summary(fix, prefix, named, rtol=1e-5):
    bad = []
    for name, t in named:
        v = t.detach().double().flatten().cpu()
        ref_abs = float(fix[f"{prefix}{name}|abs"])
        if abs(v.abs().sum().item() - ref_abs) > rtol * ref_abs + 1e-9:
            bad.append((name, "abs", v.abs().sum().item(), ref_abs))
        if abs(v.sum().item() - float(fix[f"{prefix}{name}|sum"])) > rtol * ref_abs + 1e-9:
            bad.append((name, "sum"))
        head = torch.from_numpy(fix[f"{prefix}{name}|head"])
        if (v[:head.numel()] - head).abs().max().item() > rtol * max(head.abs().max().item(), 1e-3):
            bad.append((name, "head"))
    assert not bad, bad[:6]
