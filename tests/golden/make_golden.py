"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build
container (torch 2.10.0+rocm7.0 CPU path, AVX512 host).  The reference never travels to the
GPU box; only the small .npz/.json files written here do.

    python tests/golden/make_golden.py            # needs /root/reference

Importable reference modules (models.*, selection_utils, tracking_utils, properties) are imported
as they are.  transform_helper.py and utils.py do not import here (torchvision / Levenshtein /
unidecode / wandb / optuna are not installed), so the individual definitions on the hot path
(AddGaussianNoice, pred_to_string, padder, get_text_stack) are compiled from the reference file
with `ast` and executed unchanged with only `torch` in scope — nothing is stubbed.
"""
import ast
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("QEA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import properties  # noqa: E402  (reference)
import selection_utils  # noqa: E402  (reference)
from models.model_crnn import CRNN  # noqa: E402  (reference)
from models.model_unet import UNet  # noqa: E402  (reference)

from oracle import model_oracle as mo  # noqa: E402  (only for the name-keyed weight fill)


def ref_defs(relpath, names):
    """Compile the named top-level defs/classes of a reference file and return them."""
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), relpath, "exec"), ns)
    return [ns[n] for n in names]


def load_seeded(module, shapes, seed):
    st = mo.seeded_state(shapes, seed)
    missing = set(module.state_dict().keys()) ^ set(st.keys())
    assert not missing, missing
    module.load_state_dict(st)
    return module


def sample_index(n):
    """Fixed pseudo-random subset of a flat tensor (same rule in tests/helpers.py)."""
    if n <= 512:
        return torch.arange(n)
    return torch.randperm(n, generator=torch.Generator().manual_seed(1234 + n))[:512]


PERTURB = 4e-6   # relative input perturbation of the size of an fp32 forward's accumulated rounding
PERTURB_SEEDS = tuple(range(99, 107))   # |cond is the LARGEST movement over these draws: a ReLU / max-pool input that sits
                                        # within rounding distance of its threshold flips under some draws and not others, and
                                        # an independent fp32 implementation (another summation order) may land on either side


def perturbed(x, seed=99):
    """x * (1 + PERTURB * N(0,1)): probes how far the exact gradient moves under a change of the size
    of fp32 rounding in the activations — including ReLU / max-pool decisions that flip."""
    g = torch.Generator().manual_seed(seed)
    return x * (1 + PERTURB * torch.randn(x.shape, generator=g, dtype=torch.float64)).to(x.dtype)


def grad_summary(named_params, prefix, out, named_params64=None, named_params64p=None):
    """fp32 reference run: |sum |abs |l2 |head (pins the oracle, same ATen kernels).
    fp64 reference run (optional): |s64 = fp64 values at sample_index, |l264 = fp64 l2 norm,
    |dev = l2-relative deviation of the fp32 run from the fp64 run on that sample — the accuracy
    the reference's own fp32 arithmetic attains on this input, which bounds what an independent
    fp32 implementation can be asked to match."""
    p64 = dict(named_params64) if named_params64 is not None else {}
    p64ps = [dict(d) for d in named_params64p] if named_params64p is not None else []   # one dict per perturbation draw
    for name, p in named_params:
        g = p.grad.detach().double().flatten()
        out[f"{prefix}{name}|sum"] = g.sum().item()
        out[f"{prefix}{name}|abs"] = g.abs().sum().item()
        out[f"{prefix}{name}|l2"] = g.norm().item()
        out[f"{prefix}{name}|head"] = g[:16].numpy().copy()
        if name in p64:
            g64 = p64[name].grad.detach().double().flatten()
            idx = sample_index(g64.numel())
            out[f"{prefix}{name}|s64"] = g64[idx].numpy().copy()
            out[f"{prefix}{name}|l264"] = g64.norm().item()
            out[f"{prefix}{name}|dev"] = ((g[idx] - g64[idx]).norm() / g64[idx].norm().clamp_min(1e-300)).item()
            if p64ps:
                # |cond: largest movement of the exact (fp64) gradient under PERTURB-sized input changes
                out[f"{prefix}{name}|cond"] = max(
                    ((d[name].grad.detach().double().flatten()[idx] - g64[idx]).norm() / g64[idx].norm().clamp_min(1e-300)).item()
                    for d in p64ps)


def tensor_summary(named, prefix, out):
    for name, t in named:
        v = t.detach().double().flatten()
        out[f"{prefix}{name}|sum"] = v.sum().item()
        out[f"{prefix}{name}|abs"] = v.abs().sum().item()
        out[f"{prefix}{name}|head"] = v[:16].numpy().copy()


def dev_of(t32, t64):
    return ((t32.detach().double() - t64.detach().double()).norm() / t64.detach().double().norm().clamp_min(1e-300)).item()


def to64(module):
    return module.double()


def synth_images(b, seed, h=32, w=128):
    """POS-style patches: white background, dark strokes, light noise."""
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(b, 1, h, w, generator=g) < 0.12).float()
    ink = torch.rand(b, 1, h, w, generator=g) * 0.7 + 0.3
    return (1 - m * ink + 0.02 * torch.randn(b, 1, h, w, generator=g)).clamp(0, 1)


def synth_labels(b, seed, lo=1, hi=12):
    rng = np.random.RandomState(seed)
    return ["".join(properties.char_set[i] for i in rng.randint(1, 95, rng.randint(lo, hi + 1))) for _ in range(b)]


def encode(labels):
    c2i = {c: i for i, c in enumerate(properties.char_set)}
    y = torch.tensor([c2i[c] for c in "".join(labels)], dtype=torch.int)
    return y, torch.tensor([len(l) for l in labels], dtype=torch.int)


# --------------------------------------------------------------------------------------------
def make_unet():
    out = {}
    x = synth_images(2, 11)
    out["x"] = x.numpy()
    # eval-mode forward (Phase A use, train_nn_patch.py:227,239)
    net = load_seeded(UNet(), mo.unet_state_shapes(), 1).eval()
    with torch.no_grad():
        out["y_eval"] = net(x).numpy()
    # train-mode forward + backward of scalar*MSE(y, 1) + <y, r>  (Phase B use, :312-329), fp32 and fp64
    r = torch.randn(2, 1, 32, 128, generator=torch.Generator().manual_seed(5))
    out["r"] = r.numpy()
    runs = {}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)) + tuple((("64p", sd), torch.float64) for sd in PERTURB_SEEDS):
        net = load_seeded(UNet(), mo.unet_state_shapes(), 1).to(dt).train()
        y = net(perturbed(x.to(dt), tag[1]) if isinstance(tag, tuple) else x.to(dt))
        loss = torch.nn.MSELoss()(y, torch.ones_like(y)) + (y * r.to(dt)).sum() / y.numel()
        loss.backward()
        runs[tag] = (net, y, loss)
    net, y, loss = runs["32"]
    out["y_train"] = y.detach().numpy()
    out["y_train64"] = runs["64"][1].detach().numpy()
    out["loss"] = loss.item()
    out["loss64"] = runs["64"][2].item()
    grad_summary(net.named_parameters(), "g|", out, runs["64"][0].named_parameters(),
                 [runs[("64p", sd)][0].named_parameters() for sd in PERTURB_SEEDS])
    tensor_summary(((k, v) for k, v in net.state_dict().items() if mo.is_buffer(k)), "buf|", out)
    np.savez_compressed(os.path.join(HERE, "unet_b2.npz"), **out)
    print("unet_b2", loss.item(), "max fp32-vs-fp64 grad dev", max(v for k, v in out.items() if k.endswith("|dev")))


def make_crnn():
    out = {}
    x = synth_images(3, 21)
    labels = synth_labels(3, 3)
    labels[1] = "aa" * 9          # 18 chars with 17 repeats -> needs 35 > 31 frames: infeasible (F3)
    out["labels"] = np.array(labels)
    y, ysz = encode(labels)
    for mode in ("bn_train", "bn_eval"):
        runs = {}
        for tag, dt in (("32", torch.float32), ("64", torch.float64)) + tuple((("64p", sd), torch.float64) for sd in PERTURB_SEEDS):
            net = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 2).to(dt)
            net.register_backward_hook(net.backward_hook)       # train_nn_patch.py:94
            net.train()
            if mode == "bn_eval":                                 # utils.py:113-115 via train_nn_patch.py:314
                for m in net.modules():
                    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                        m.eval()
            xi = (perturbed(x.detach().to(dt), tag[1]) if isinstance(tag, tuple) else x.detach().to(dt)).requires_grad_()
            lp = net(xi)
            T = lp.shape[0]
            insz = torch.tensor([T] * 3, dtype=torch.int)
            per = torch.nn.CTCLoss(reduction="none")(lp, y, insz, ysz)
            loss = torch.nn.CTCLoss()(lp, y, insz, ysz)
            loss.backward()
            runs[tag] = (net, lp, per, loss, xi)
        net, lp, per, loss, xi = runs["32"]
        out[f"{mode}|lp"] = lp.detach().numpy()
        out[f"{mode}|lp64"] = runs["64"][1].detach().numpy()
        out[f"{mode}|nll"] = per.detach().numpy()
        out[f"{mode}|loss"] = loss.item()
        out[f"{mode}|dx"] = xi.grad.numpy().copy()
        out[f"{mode}|dx64"] = runs["64"][4].grad.numpy().copy()
        grad_summary(net.named_parameters(), f"{mode}|g|", out, runs["64"][0].named_parameters(),
                     [runs[("64p", sd)][0].named_parameters() for sd in PERTURB_SEEDS])
        out[f"{mode}|dxcond"] = max(dev_of(runs[("64p", sd)][4].grad, runs["64"][4].grad) for sd in PERTURB_SEEDS)
        tensor_summary(((k, v) for k, v in net.state_dict().items() if mo.is_buffer(k)), f"{mode}|buf|", out)
        print("crnn", mode, loss.item(), per.detach().numpy(), "max dev", max(v for k, v in out.items() if k.startswith(mode) and k.endswith("|dev")))
    out["x"] = x.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "crnn_b3.npz"), **out)


def make_ctc():
    out = {}
    g = torch.Generator().manual_seed(7)
    T, N, C = 31, 8, 95
    lp = (torch.randn(T, N, C, generator=g) * 2).log_softmax(2).requires_grad_()
    rng = np.random.RandomState(9)
    tl = [5, 1, 15, 12, 0, 31, 16, 3]
    tg = [list(rng.randint(1, C, l)) for l in tl]
    tg[2] = [4, 4, 9, 9, 9, 2, 3, 3, 1, 5, 6, 7, 8, 8, 2]   # 5 repeats: 15 + 5 = 20 <= 31 feasible
    tg[3] = [7] * 12                                          # 12 + 11 = 23 feasible, all repeats
    tg[5] = list(rng.randint(1, C, 31))                       # len == T: feasible only if no repeats
    for i in range(1, 31):
        if tg[5][i] == tg[5][i - 1]:
            tg[5][i] = tg[5][i] % (C - 1) + 1
    tg[6] = [3] * 16                                          # 16 + 15 = 31 -> exactly feasible
    tg[7] = [0 + 1, 1 + 1, 1 + 1]
    flat = torch.tensor([c for t in tg for c in t], dtype=torch.int)
    tls = torch.tensor(tl, dtype=torch.int)
    ins = torch.full((N,), T, dtype=torch.int)
    per = torch.nn.CTCLoss(reduction="none")(lp, flat, ins, tls)
    loss = torch.nn.CTCLoss()(lp, flat, ins, tls)
    loss.backward()
    out.update(lp=lp.detach().numpy(), targets=flat.numpy(), target_lengths=tls.numpy(), nll=per.detach().numpy(),
               loss_mean=loss.item(), grad_mean=lp.grad.numpy().copy())
    # a second batch with infeasible members (loss = inf, NaN grads before the scrub)
    lp2 = (torch.randn(T, 4, C, generator=g)).log_softmax(2).requires_grad_()
    tl2 = [40, 2, 17, 20]
    tg2 = [list(rng.randint(1, C, 40)), [5, 5], [6] * 17, list(rng.randint(1, C, 20))]
    flat2 = torch.tensor([c for t in tg2 for c in t], dtype=torch.int)
    tls2 = torch.tensor(tl2, dtype=torch.int)
    ins2 = torch.full((4,), T, dtype=torch.int)
    per2 = torch.nn.CTCLoss(reduction="none")(lp2, flat2, ins2, tls2)
    loss2 = torch.nn.CTCLoss()(lp2, flat2, ins2, tls2)
    loss2.backward()
    out.update(inf_lp=lp2.detach().numpy(), inf_targets=flat2.numpy(), inf_target_lengths=tls2.numpy(),
               inf_nll=per2.detach().numpy(), inf_loss_mean=loss2.item(), inf_grad_mean=lp2.grad.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "ctc_cases.npz"), **out)
    print("ctc", per.detach().numpy(), per2.detach().numpy())


def make_topk():
    cases = []
    Sampler = selection_utils.datasampler_factory("topKCER")
    rng = np.random.RandomState(3)
    real = json.load(open(os.path.join(REF, "cer_data_utils", "pos_dataset_cers.json")))
    real_items = list(real.items())

    def run(cers, names, k, note):
        s = Sampler(dict(cers))
        imgs = torch.arange(len(names)).float().view(-1, 1)
        sel_imgs, sel_labels, idx = s.query(imgs, list(names), k, list(names))
        cases.append({"note": note, "names": list(names), "cers": cers, "k": k, "idx": idx.tolist(),
                      "sel": sel_imgs.view(-1).long().tolist()})

    # tie-free, sizes around the n >= 17 threshold of SURVEY F4 and the bench batch
    for n in (5, 16, 17, 64, 512):
        names = [f"s{i}" for i in range(n)]
        vals = rng.permutation(n).astype(np.float64) / n
        run({nm: float(v) for nm, v in zip(names, vals)}, names, max(1, math.ceil(n * 0.05)), f"tie-free n={n}")
    # ties, n <= 16 (torch's small-sort path is stable there)
    names = [f"t{i}" for i in range(12)]
    run({nm: float(v) for nm, v in zip(names, [0, 1, 0.5, 1, 0, 0.5, 0.25, 1, 0, 0, 0.5, 0.25])}, names, 5, "ties n=12")
    # ties at n > 16: value multiset is the contract; index order is build-specific (recorded)
    names = [f"u{i}" for i in range(40)]
    vals = rng.choice([0.0, 0.5, 1.0, 1 / 3], 40)
    run({nm: float(v) for nm, v in zip(names, vals)}, names, 10, "ties n=40 (index order build-specific: torch 2.10.0+rocm7.0 AVX512)")
    # real POS CER slices (cer_data_utils/pos_dataset_cers.json), with a name missing from the dict
    for start, n in ((0, 20), (1000, 24), (5000, 64)):
        sl = real_items[start:start + n]
        names = [k for k, _ in sl]
        run({k: float(v) for k, v in sl}, names, max(1, math.ceil(n * 0.5)), f"real slice {start}:{start + n}")
    sl = real_items[200:216]
    names = [k for k, _ in sl]
    d = {k: float(v) for k, v in sl}
    del d[names[3]]
    s = Sampler(d)
    imgs = torch.arange(16).float().view(-1, 1)
    _, _, idx = s.query(imgs, names, 4, names)
    cases.append({"note": "name missing from dict (indices address the compacted list)", "names": names, "cers": d,
                  "k": 4, "idx": idx.tolist(), "sel": None})
    json.dump({"torch": torch.__version__, "cases": cases}, open(os.path.join(HERE, "topk_cases.json"), "w"))
    print("topk", len(cases))


def make_helpers():
    out = {}
    (Noise,) = ref_defs("transform_helper.py", ["AddGaussianNoice"])
    pred_to_string, padder, get_text_stack = ref_defs("utils.py", ["pred_to_string", "padder", "get_text_stack"])
    img = synth_images(1, 31)[0]
    for stochastic in (False, True):
        torch.manual_seed(123)
        o, n = Noise(std=5, is_stochastic=stochastic, return_noise=True)(img.clone())
        out[f"jit{int(stochastic)}|out"] = o.numpy()
        out[f"jit{int(stochastic)}|noise"] = n.numpy()
    out["jit|img"] = img.numpy()
    torch.manual_seed(123)
    o, n = Noise(std=3, return_noise=True)(img.clone(), noise_coef=0.5)
    out["jitc|out"], out["jitc|noise"] = o.numpy(), n.numpy()
    # greedy decode
    g = torch.Generator().manual_seed(17)
    scores = torch.randn(31, 6, 95, generator=g)
    scores[:, :, 0] += 2.5      # plenty of blanks
    scores[3:6, 0, :] = scores[3:4, 0, :]   # repeats
    i2c = {i: c for i, c in enumerate(properties.char_set)}
    out["dec|scores"] = scores.numpy()
    out["dec|strings"] = np.array(pred_to_string(scores, [""] * 6, i2c))
    # crop + pad
    page = torch.rand(1, 60, 200, generator=g)
    boxes = [dict(label="ab", x_min=3, y_min=4, x_max=100, y_max=30), dict(label="c", x_min=150, y_min=0, x_max=199, y_max=31),
             dict(label="d", x_min=10, y_min=20, x_max=137, y_max=51), dict(label="e", x_min=0, y_min=0, x_max=1, y_max=1)]
    stack, labels = get_text_stack(page, boxes, (32, 128))
    out["crop|page"] = page.numpy()
    out["crop|boxes"] = np.array([[b["x_min"], b["y_min"], b["x_max"], b["y_max"]] for b in boxes])
    out["crop|stack"] = stack.numpy()
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    print("helpers", out["dec|strings"])


def make_step():
    """One full train_nn_area-style step (Phase A then Phase B, train_nn_area.py:212-287) on the
    reference modules: TopKCER pick, 2 jitter replicas with explicit seeded noise, a fixed label
    oracle in place of the OCR engine, CTC, Adam(CRNN); then UNet(train)->CRNN(BN eval)->
    CTC+MSE->Adam(UNet)."""
    out = {}
    B, inner, prop, std = 4, 2, 0.5, 5
    x = synth_images(B, 41)
    labels = synth_labels(B, 43, 2, 8)
    names = [f"img{i}.png" for i in range(B)]
    cers = {n: c for n, c in zip(names, [0.25, 0.8, 0.0, 0.5])}
    prep = load_seeded(UNet(), mo.unet_state_shapes(), 3)
    crnn = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 4)
    crnn.register_backward_hook(crnn.backward_hook)
    opt_c = torch.optim.Adam(crnn.parameters(), lr=1e-4, weight_decay=0)
    opt_p = torch.optim.Adam(prep.parameters(), lr=5e-5, weight_decay=0)
    ctc = torch.nn.CTCLoss()
    sampler = selection_utils.datasampler_factory("topKCER")(dict(cers))
    # ---- Phase A
    crnn.train(); prep.eval(); prep.zero_grad(); crnn.zero_grad()
    preds_all = prep(x)
    k = max(1, math.ceil(B * (1 - prop)))
    preds, labels_sel, idx = sampler.query(preds_all, labels, k, names)
    preds = preds.detach()
    out["A|idx"] = idx.numpy()
    g = torch.Generator().manual_seed(45)
    losses = []
    for i in range(inner):
        noise = torch.randn(preds.shape, generator=g) * (std / 100.0)
        noisy = (preds - noise).clamp(0, 1)
        out[f"A|noise{i}"] = noise.numpy()
        ocr_labels = [l[::-1] for l in labels_sel]     # the "OCR" of this fixture: reversed GT
        lp = crnn(noisy)
        y, ysz = encode(ocr_labels)
        loss = ctc(lp, y, torch.tensor([lp.shape[0]] * len(ocr_labels), dtype=torch.int), ysz)
        losses.append(loss.item())
    loss.backward()                                      # area trainer: last replica only (F6)
    out["A|losses"] = np.array(losses)
    # fp64 twin of the last replica (same noisy input), for the conditioning-aware HIP check
    crnn64 = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 4).double()
    crnn64.register_backward_hook(crnn64.backward_hook)
    crnn64.train()
    with torch.no_grad():
        for i in range(inner - 1):                       # earlier replicas only move the BN running stats
            noisy_i = (preds - torch.from_numpy(out[f"A|noise{i}"])).clamp(0, 1)
            crnn64(noisy_i.double())
    lp64 = crnn64(noisy.double())
    ctc(lp64, y, torch.tensor([lp64.shape[0]] * len(ocr_labels), dtype=torch.int), ysz).backward()
    crnn64ps = []
    for sd in PERTURB_SEEDS:
        crnn64p = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 4).double()
        crnn64p.register_backward_hook(crnn64p.backward_hook)
        crnn64p.train()
        ctc(crnn64p(perturbed(noisy.double(), sd)), y, torch.tensor([lp64.shape[0]] * len(ocr_labels), dtype=torch.int), ysz).backward()
        crnn64ps.append(crnn64p)
    grad_summary(crnn.named_parameters(), "A|g|", out, crnn64.named_parameters(), [m.named_parameters() for m in crnn64ps])
    opt_c.step()
    tensor_summary(crnn.state_dict().items(), "A|crnn|", out)
    # ---- Phase B starts from a freshly seeded CRNN (seed 6), NOT from the post-Phase-A weights:
    # Adam's first step is lr*sign(g) wherever |g| is rounding noise, so the post-step weights of
    # the reference itself differ by up to 1.6e-4 between a 1-thread and an 8-thread run, and that
    # moves individual Phase-B UNet gradients by 1-2 % (measured here) — a chained A->B fixture
    # cannot be pinned to 1e-4.  The A-then-B ordering is pinned by the post-step
    # summaries above; the Phase-B arithmetic by the block below.
    crnn = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 6)
    crnn.register_backward_hook(crnn.backward_hook)
    # ---- Phase B
    prep.train(); crnn.train()
    for m in crnn.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x)
    lp = crnn(img)
    y, ysz = encode(labels)
    lossB = ctc(lp, y, torch.tensor([lp.shape[0]] * B, dtype=torch.int), ysz) + torch.nn.MSELoss()(img, torch.ones_like(img)) * 1.0
    lossB.backward()
    prep64 = load_seeded(UNet(), mo.unet_state_shapes(), 3).double().train()
    crnn64 = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 6).double()
    crnn64.register_backward_hook(crnn64.backward_hook)
    crnn64.train()
    for m in crnn64.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    img64 = prep64(x.double())
    lp64 = crnn64(img64)
    loss64 = ctc(lp64, y, torch.tensor([lp64.shape[0]] * B, dtype=torch.int), ysz) + torch.nn.MSELoss()(img64, torch.ones_like(img64)) * 1.0
    loss64.backward()
    prep64ps, crnn64ps = [], []
    for sd in PERTURB_SEEDS:
        prep64p = load_seeded(UNet(), mo.unet_state_shapes(), 3).double().train()
        crnn64p = load_seeded(CRNN(95, False), mo.crnn_state_shapes(), 6).double()
        crnn64p.register_backward_hook(crnn64p.backward_hook)
        crnn64p.train()
        for m in crnn64p.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        img64p = prep64p(perturbed(x.double(), sd))
        (ctc(crnn64p(img64p), y, torch.tensor([lp64.shape[0]] * B, dtype=torch.int), ysz) + torch.nn.MSELoss()(img64p, torch.ones_like(img64p))).backward()
        prep64ps.append(prep64p)
        crnn64ps.append(crnn64p)
    out["B|loss"] = lossB.item()
    out["B|loss64"] = loss64.item()
    out["B|img"] = img.detach().numpy()
    out["B|img64"] = img64.detach().numpy()
    out["B|lp"] = lp.detach().numpy()
    out["B|lp64"] = lp64.detach().numpy()
    grad_summary(prep.named_parameters(), "B|g|prep|", out, prep64.named_parameters(), [m.named_parameters() for m in prep64ps])
    grad_summary(crnn.named_parameters(), "B|g|crnn|", out, crnn64.named_parameters(), [m.named_parameters() for m in crnn64ps])
    opt_p.step()
    tensor_summary(prep.state_dict().items(), "B|prep|", out)
    print("step max dev A", max(v for k, v in out.items() if k.startswith("A|g|") and k.endswith("|dev")),
          "B", max(v for k, v in out.items() if k.startswith("B|g|") and k.endswith("|dev")))
    out["x"] = x.numpy()
    out["labels"] = np.array(labels)
    out["names"] = np.array(names)
    out["cers"] = np.array([cers[n] for n in names])
    np.savez_compressed(os.path.join(HERE, "step_area_b4.npz"), **out)
    print("step", losses, lossB.item(), idx)


# ============================================================================================
# Round 2: KNIFE-EDGE-FREE fixtures.
#
# The fixtures above (He-normal weights, random running statistics, B = 2-4 at 32x128) cannot carry north_star's
# "grads within 1e-4": the gradient of a ReLU / max-pool network is a DISCONTINUOUS function of its forward
# activations, and at 32x128 every pass evaluates ~1 M decisions per image.  Measured here on the reference itself
# (default-init distributions, uniform images, B = 4): in 11 of 12 (weight seed, image seed) pairs the reference's own
# fp32 run differs from its fp64 run by 2.5e-3 ... 1.1e-2 on some gradient tensor (one flipped decision moves the weight
# gradient below it by ~1/sqrt(#pixels)); the one clean pair moves by 6.5e-3 under a 1e-7 relative input perturbation;
# and the smallest |pre-activation| of a layer is 5-100x SMALLER than that layer's fp32-vs-fp64 error in every candidate.
# A decision-flip-free comparison between two independent fp32 implementations therefore only exists where the decision
# count is small: the fixtures below use the same two networks at B = 2 on 32x32 and 32x64 images (T = 7 / 15), and a
# candidate is accepted only if NINE fp32 evaluations of the reference (default, reversed batch order = another BatchNorm
# summation order, one thread, and six draws of the input moved by <= 1 fp32 ulp) ALL agree with the
# fp64 run to COND_MAX on EVERY gradient tensor of both phases.  The HIP path is then gated on them at a plain
# ||g - g64|| / ||g64|| <= 1e-4 on full tensors (tests/test_conditioned_gpu.py).
# ============================================================================================
COND_MAX = 2.5e-5
ZERO_GRAD = ("convo.conv5.bias", "convo.conv6.bias")   # exactly zero in Phase A (bias in front of a batch-statistics BatchNorm)


def _bn_eval(net):
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    return net


def load_default(module, shapes, seed):
    st = mo.default_init_state(shapes, seed)
    assert not (set(module.state_dict().keys()) ^ set(st.keys()))
    module.load_state_dict(st)
    return module


def _ref_models(ws, dt):
    prep = load_default(UNet(), mo.unet_state_shapes(), ws).to(dt)
    crnn = load_default(CRNN(95, False), mo.crnn_state_shapes(), ws + 1).to(dt)
    crnn.register_backward_hook(crnn.backward_hook)
    return prep, crnn


class RefRunner:
    """The reference modules for one weight seed, re-armed (state reloaded) before every evaluation."""

    def __init__(self, ws):
        self.ws = ws
        self.m, self.st = {}, {}
        for dt in (torch.float32, torch.float64):
            prep, crnn = _ref_models(ws, dt)
            self.m[dt] = (prep, crnn)
            self.st[dt] = ({k: v.clone() for k, v in prep.state_dict().items()}, {k: v.clone() for k, v in crnn.state_dict().items()})

    def arm(self, dt):
        prep, crnn = self.m[dt]
        prep.load_state_dict(self.st[dt][0]); crnn.load_state_dict(self.st[dt][1])
        prep.zero_grad(); crnn.zero_grad()
        prep.train(); crnn.train()
        return prep, crnn

    def phase_b(self, dt, x, labels, variant="default"):
        """train_nn_area.py:277-287: UNet(train) -> CRNN(train, BN eval) -> CTC + MSE -> backward."""
        prep, crnn = self.arm(dt)
        _bn_eval(crnn)
        xi, perm = _variant_input(x.to(dt), variant)
        lab = [labels[i] for i in perm]
        img = prep(xi)
        lp = crnn(img)
        y, ysz = encode(lab)
        ins = torch.tensor([lp.shape[0]] * x.shape[0], dtype=torch.int)
        loss = torch.nn.CTCLoss()(lp, y, ins, ysz) + torch.nn.MSELoss()(img, torch.ones_like(img))
        loss.backward()
        inv = torch.argsort(torch.tensor(perm))
        return dict(prep=prep, crnn=crnn, img=img.detach()[inv], lp=lp.detach()[:, inv], loss=loss.item())

    def phase_a(self, dt, x, labels, variant="default"):
        """train_nn_area.py:262-271: CRNN(train-mode BN) -> CTC -> backward; gradient wrt the input kept."""
        _, crnn = self.arm(dt)
        xi, perm = _variant_input(x.to(dt), variant)
        xi = xi.clone().requires_grad_()
        lab = [labels[i] for i in perm]
        lp = crnn(xi)
        y, ysz = encode(lab)
        ins = torch.tensor([lp.shape[0]] * x.shape[0], dtype=torch.int)
        loss = torch.nn.CTCLoss()(lp, y, ins, ysz)
        loss.backward()
        inv = torch.argsort(torch.tensor(perm))
        return dict(crnn=crnn, dx=xi.grad.detach()[inv], lp=lp.detach()[:, inv], loss=loss.item())


FP32_VARIANTS = ("default", "reversed", "one_thread", "ulp5", "ulp6", "ulp7", "ulp8", "ulp9", "ulp10")


def _variant_input(x, variant):
    """Independent fp32 evaluations of the same function: another summation order of the batch statistics (reversed
    batch), another reduction partition (one thread), and six draws of the inputs moved by <= 1 fp32 ulp (every
    downstream rounding then falls differently; this is the variant that rejects most candidates)."""
    perm = list(range(x.shape[0]))
    if variant == "reversed":
        perm = perm[::-1]
        x = x[perm].contiguous()
    elif variant.startswith("ulp"):
        g = torch.Generator().manual_seed(int(variant[3:]))
        x = (x.double() * (1 + 5.96e-8 * (torch.rand(x.shape, generator=g, dtype=torch.float64) * 2 - 1))).to(x.dtype)
    return x, perm


def _grads(named):
    return {n: p.grad.detach().double().clone() for n, p in named}


def _worst_rel(ga, g64, skip=()):
    return max(((ga[n] - g64[n]).norm() / g64[n].norm().clamp_min(1e-300)).item() for n in g64 if n not in skip)


def _candidate_ok(R, x, labels, labels_a):
    """-> (accepted, per-variant worst deviations).  Cheapest rejections first."""
    figs = {}
    b64 = R.phase_b(torch.float64, x, labels)
    gB = {**{"prep." + n: g for n, g in _grads(b64["prep"].named_parameters()).items()},
          **{"crnn." + n: g for n, g in _grads(b64["crnn"].named_parameters()).items()}}
    a64 = R.phase_a(torch.float64, x, labels_a)
    gA = {**_grads(a64["crnn"].named_parameters()), "dx": a64["dx"].double()}
    for v in FP32_VARIANTS:
        if v == "one_thread":
            torch.set_num_threads(1)
        try:
            b = R.phase_b(torch.float32, x, labels, v)
            g = {**{"prep." + n: q for n, q in _grads(b["prep"].named_parameters()).items()},
                 **{"crnn." + n: q for n, q in _grads(b["crnn"].named_parameters()).items()}}
            wb = _worst_rel(g, gB)
            if wb > COND_MAX:
                return False, {**figs, v: ("B", wb)}
            a = R.phase_a(torch.float32, x, labels_a, v)
            wa = _worst_rel({**_grads(a["crnn"].named_parameters()), "dx": a["dx"].double()}, gA, skip=ZERO_GRAD)
            if wa > COND_MAX:
                return False, {**figs, v: ("A", wa)}
            figs[v] = max(wb, wa)
        finally:
            torch.set_num_threads(8)
    return True, figs


N_S64 = 128     # fp64 samples kept per gradient tensor (the full tensors come from the pinned oracle at test time)


def sample_index_small(n):
    if n <= N_S64:
        return torch.arange(n)
    return torch.randperm(n, generator=torch.Generator().manual_seed(4321 + n))[:N_S64]


def full64(named_params64, prefix, out):
    """Per tensor of the fp64 reference run: |s64 (values at sample_index_small), |l264, |sum64."""
    for name, p in named_params64:
        g64 = p.grad.detach().double().flatten()
        out[f"{prefix}{name}|s64"] = g64[sample_index_small(g64.numel())].numpy().copy()
        out[f"{prefix}{name}|l264"] = g64.norm().item()
        out[f"{prefix}{name}|sum64"] = g64.sum().item()


COND_SHAPES = ((2, 32, 50, 1000, 6), (4, 64, 54, 4000, 6), (4, 128, 56, 3000, 8))   # B, W, weight seed, first image seed, candidates kept


def make_conditioned():
    """tests/golden/cond_b2w32.npz, cond_b4w64.npz, cond_b4w128.npz: for each shape the first K image seeds (fixed weight
    seed, fixed seed order) that pass _candidate_ok, stored as candidates c0..c{K-1}; the rejected seeds and their figures
    are stored with them.  Several candidates per shape, because passing nine evaluations of ATen's fp32 kernels does not
    make a candidate flip-free for ANOTHER fp32 implementation (its roundings fall elsewhere): the GPU tests state their
    requirement over the candidate set (tests/test_conditioned_gpu.py)."""
    for B, W, ws, xs0, keep in COND_SHAPES:
        T = W // 4 - 1
        R = RefRunner(ws)
        log, out, k = [], {}, 0
        for xs in range(xs0, xs0 + 2000):
            x = torch.rand(B, 1, 32, W, generator=torch.Generator().manual_seed(xs))
            labels = synth_labels(B, xs, 1, max(1, T // 2))
            labels_a = synth_labels(B, xs + 100, 1, max(1, T // 2))
            ok, figs = _candidate_ok(R, x, labels, labels_a)
            log.append(f"{xs}:{'ok' if ok else 'rejected'}:{figs}")
            if not ok:
                continue
            c = f"c{k}|"
            print(f"cond B={B} W={W}: candidate {k} = image seed {xs} ({len(log) - k - 1} rejections so far); fp32 variants vs fp64: max {max(figs.values()):.2e}", flush=True)
            out.update({c + "x": x.numpy(), c + "labels": np.array(labels), c + "labels_a": np.array(labels_a), c + "xs": xs,
                        c + "variant_worst": np.array([figs[v] for v in FP32_VARIANTS])})
            b64 = R.phase_b(torch.float64, x, labels)
            full64(b64["prep"].named_parameters(), c + "B|g|prep|", out)
            full64(b64["crnn"].named_parameters(), c + "B|g|crnn|", out)
            out.update({c + "B|loss64": b64["loss"], c + "B|img64": b64["img"].numpy(), c + "B|lp64": b64["lp"].numpy(),
                        c + "B|loss32": R.phase_b(torch.float32, x, labels)["loss"]})
            for kk, v in b64["prep"].state_dict().items():
                if mo.is_buffer(kk) and v.is_floating_point():
                    out[f"{c}B|buf|{kk}"] = v.numpy().copy()
            a64 = R.phase_a(torch.float64, x, labels_a)
            full64(a64["crnn"].named_parameters(), c + "A|g|", out)
            out.update({c + "A|loss64": a64["loss"], c + "A|lp64": a64["lp"].numpy(), c + "A|dx64": a64["dx"].numpy(),
                        c + "A|loss32": R.phase_a(torch.float32, x, labels_a)["loss"]})
            for kk, v in a64["crnn"].state_dict().items():
                if mo.is_buffer(kk) and v.is_floating_point():
                    out[f"{c}A|buf|{kk}"] = v.numpy().copy()
            k += 1
            if k == keep:
                break
        else:
            raise SystemExit(f"B={B} W={W}: fewer than {keep} knife-edge-free candidates in 2000 seeds")
        out.update({"ws": ws, "W": W, "B": B, "n_candidates": keep, "search_log": np.array(log)})
        np.savez_compressed(os.path.join(HERE, f"cond_b{B}w{W}.npz"), **out)


def _hook_sites(prep, crnn, rec):
    """forward hooks that record, under the oracle's site names (oracle/model_oracle.py::Trace), the output of every conv /
    BatchNorm / transposed conv of the reference modules; -> handles"""
    hs = []

    def keep(name):
        def hook(mod, inp, out):
            rec[name] = out.detach().clone()          # clone: the UNet's ReLUs are in-place
        return hook
    if prep is not None:
        for name, m in prep.named_modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.ConvTranspose2d)) and name != "conv":
                hs.append(m.register_forward_hook(keep(name)))
    for name, m in crnn.named_modules():
        if name.startswith("convo.") and isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d)):
            hs.append(m.register_forward_hook(keep(name)))
    return hs


def _complete_rec(rec):
    """add the max-pool input sites (the ReLU outputs in front of the pools) to a hooked record"""
    for l in (1, 2, 3, 4):
        k = f"encoder{l}.enc{l}norm2"
        if k in rec:
            rec[f"pool{l}"] = torch.relu(rec[k])
    for site, src in (("convo.pool1", "convo.conv1"), ("convo.pool2", "convo.conv2"), ("convo.pool4", "convo.conv4"),
                      ("convo.pool6", "convo.batchnorm2")):
        rec[site] = torch.relu(rec[src])
    return rec


def make_ladder():
    """tests/golden/ladder.json — what the REFERENCE's own fp32 arithmetic does on the cond_b*.npz candidates, layer by layer,
    against its fp64 run: l2-relative error of every conv / BatchNorm / transposed-conv output, the ReLU / max-pool decisions
    that come out differently (count, and the worst fp64 margin in fp32 rounding units of the layer, tests/decisions.py),
    and the worst full-tensor gradient error.  Two fp32 evaluations: ATen's default CPU path (oneDNN convolutions) and the
    same with torch.backends.mkldnn disabled (ATen's own im2col + GEMM: another summation order).  The GPU tests print the
    HIP path's figures next to these (tests/test_conditioned_gpu.py::test_forward_ladder_and_flip_attribution)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import decisions as D

    def entry(R, x, labels, labels_a, tag):
        runs = {}
        for variant, dt, mk in (("fp64", torch.float64, True), ("default", torch.float32, True), ("nomkldnn", torch.float32, False)):
            with torch.backends.mkldnn.flags(enabled=mk):
                recB, recA = {}, {}
                hs = _hook_sites(*R.m[dt], recB)
                b = R.phase_b(dt, x, labels)
                gB = {**{"prep." + n: g for n, g in _grads(b["prep"].named_parameters()).items()},
                      **{"crnn." + n: g for n, g in _grads(b["crnn"].named_parameters()).items()}}
                for h in hs:
                    h.remove()
                hs = _hook_sites(None, R.m[dt][1], recA)
                a = R.phase_a(dt, x, labels_a)
                gA = {**_grads(a["crnn"].named_parameters()), "dx": a["dx"].double()}
                for h in hs:
                    h.remove()
            runs[variant] = dict(recB=_complete_rec(recB), recA=_complete_rec(recA), gB=gB, gA=gA, img=b["img"])
        r64 = runs["fp64"]
        res = {}
        for variant in ("default", "nomkldnn"):
            r = runs[variant]
            ent = {"worst_grad_B": _worst_rel(r["gB"], r64["gB"]), "worst_grad_A": _worst_rel(r["gA"], r64["gA"], skip=ZERO_GRAD),
                   "img_maxabs": (r["img"].double() - r64["img"]).abs().max().item()}
            for ph, key in (("B", "recB"), ("A", "recA")):
                sites = [s_ for s_ in r[key] if s_ not in mo.POOL_SITES]
                ent["err_" + ph] = {s_: D.rel_l2(r[key][s_], r64[key][s_]) for s_ in sites}
                fl = D.flip_report(mo.own_decisions(r[key]), r64[key])
                ent["flips_" + ph] = {s_: [v[0], v[1], round(v[2], 3)] for s_, v in fl.items()}
            res[variant] = ent
            print(f"ladder {tag} {variant}: grad worst B {ent['worst_grad_B']:.2e} A {ent['worst_grad_A']:.2e}; img {ent['img_maxabs']:.1e}; "
                  f"flips B {sum(v[0] for v in ent['flips_B'].values())} A {sum(v[0] for v in ent['flips_A'].values())}; "
                  f"worst units {max(v[2] for v in list(ent['flips_B'].values()) + list(ent['flips_A'].values())):.1f}; "
                  f"max layer err B {max(ent['err_B'].values()):.2e}", flush=True)
        return res

    out = {}
    for B, W, ws, xs0, keep in COND_SHAPES:
        case = f"cond_b{B}w{W}.npz"
        fx = np.load(os.path.join(HERE, case), allow_pickle=False)
        R = RefRunner(ws)
        out[case] = {}
        for k in range(keep):
            c = f"c{k}|"
            out[case][f"c{k}"] = entry(R, torch.from_numpy(fx[c + "x"]), [str(s) for s in fx[c + "labels"]],
                                       [str(s) for s in fx[c + "labels_a"]], f"{case} c{k}")
        if B == 4:
            # the same shape WITHOUT the knife-edge-free selection: the first 8 image seeds of the search, accepted or not.  The
            # selected candidates are by construction inputs on which nine evaluations of ATen's fp32 path take no decision the
            # other way, so flip counts on them are biased towards zero for ATen (and for anything sharing its arithmetic)
            T = W // 4 - 1
            seeds = list(range(xs0, xs0 + 8))
            un = {"B": B, "W": W, "ws": ws, "seeds": seeds}
            for i, xs in enumerate(seeds):
                x = torch.rand(B, 1, 32, W, generator=torch.Generator().manual_seed(xs))
                un[f"c{i}"] = entry(R, x, synth_labels(B, xs, 1, max(1, T // 2)), synth_labels(B, xs + 100, 1, max(1, T // 2)),
                                    f"unselected b{B}w{W} seed {xs}")
            out[f"unselected_b{B}w{W}"] = un
    json.dump(out, open(os.path.join(HERE, "ladder.json"), "w"), indent=0)


def make_crop_oversize():
    """tests/golden/crop_oversize.npz — get_text_stack / padder (utils.py:118-141) on boxes LARGER than 32x128 by odd and
    even amounts: ConstantPad2d with negative pads crops, and Python floor division decides which side loses the extra pixel."""
    _, padder, get_text_stack = ref_defs("utils.py", ["pred_to_string", "padder", "get_text_stack"])
    g = torch.Generator().manual_seed(23)
    page = torch.rand(1, 80, 300, generator=g)
    boxes = [dict(label="a", x_min=5, y_min=3, x_max=136, y_max=38),      # 131 x 35: odd oversize both ways
             dict(label="b", x_min=100, y_min=40, x_max=230, y_max=74),   # 130 x 34: even oversize
             dict(label="c", x_min=0, y_min=0, x_max=129, y_max=20),      # 129 wide (odd), 20 high (padded)
             dict(label="d", x_min=200, y_min=10, x_max=260, y_max=43)]   # 60 wide (padded), 33 high (odd oversize)
    stack, _ = get_text_stack(page, boxes, (32, 128))
    np.savez_compressed(os.path.join(HERE, "crop_oversize.npz"), page=page.numpy(), stack=stack.numpy(),
                        boxes=np.array([[b["x_min"], b["y_min"], b["x_max"], b["y_max"]] for b in boxes]))
    print("crop_oversize", stack.shape)


def make_area_step():
    """tests/golden/area_step_b8.npz — ONE minibatch of train_nn_area.TrainNNPrep.train (train_nn_area.py:212-304) on the
    reference modules from default-distribution weights, chained A -> B exactly as the trainer does it: TopKCER (prop 0.5),
    2 jitter replicas with recorded noise, the stub label source (GT reversed), CTC, backward of the last replica,
    Adam(CRNN); UNet(train) -> CRNN(BN eval) -> CTC + MSE -> Adam(UNet); decode + CER update.  Driven on the HIP side
    through TrainNNPrep itself (tests/test_trainers_gpu.py::test_area_trainer_one_minibatch_vs_reference)."""
    out = {}
    B, inner, prop, std, ws = 8, 2, 0.5, 5, 20
    x = torch.rand(B, 1, 32, 128, generator=torch.Generator().manual_seed(31))
    labels = synth_labels(B, 33, 2, 8)
    names = [f"img{i}.png" for i in range(B)]
    cers = {n: c for n, c in zip(names, [0.25, 0.8, 0.0, 0.5, 0.9, 0.1, 0.3, 0.7])}
    prep, crnn = _ref_models(ws, torch.float32)
    opt_c = torch.optim.Adam(crnn.parameters(), lr=1e-4, weight_decay=0)
    opt_p = torch.optim.Adam(prep.parameters(), lr=5e-5, weight_decay=0)
    pre = {("prep|" + k): v.clone() for k, v in prep.state_dict().items()}
    pre.update({("crnn|" + k): v.clone() for k, v in crnn.state_dict().items()})
    ctc = torch.nn.CTCLoss()
    sampler = selection_utils.datasampler_factory("topKCER")(dict(cers))
    (pred_to_string,) = ref_defs("utils.py", ["pred_to_string"])
    # ---- Phase A (train_nn_area.py:214-275)
    crnn.train(); prep.eval(); prep.zero_grad(); crnn.zero_grad()
    preds_all = prep(x)
    k = max(1, math.ceil(B * (1 - prop)))
    preds, labels_sel, idx = sampler.query(preds_all, labels, k, names)
    preds = preds.detach()
    out["A|idx"] = idx.numpy()
    g = torch.Generator().manual_seed(35)
    losses = []
    for i in range(inner):
        noise = torch.randn(preds.shape, generator=g) * (std / 100.0)
        noisy = (preds - noise).clamp(0, 1)
        out[f"A|noise{i}"] = noise.numpy()
        ocr_labels = [l[::-1] for l in labels_sel]
        lp = crnn(noisy)
        y, ysz = encode(ocr_labels)
        loss = ctc(lp, y, torch.tensor([lp.shape[0]] * len(ocr_labels), dtype=torch.int), ysz)
        losses.append(loss.item())
    loss.backward()
    opt_c.step()
    out["A|losses"] = np.array(losses)
    # ---- Phase B (:277-287)
    prep.train(); crnn.train(); _bn_eval(crnn)
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x)
    lp = crnn(img)
    y, ysz = encode(labels)
    lossB = ctc(lp, y, torch.tensor([lp.shape[0]] * B, dtype=torch.int), ysz) + torch.nn.MSELoss()(img, torch.ones_like(img)) * 1.0
    lossB.backward()
    opt_p.step()
    out["B|loss"] = lossB.item()
    # ---- CER update (:290-304)
    i2c = {i: c for i, c in enumerate(properties.char_set)}
    out["B|decoded"] = np.array(pred_to_string(lp.detach(), labels, i2c))
    # post-step state: update (post - pre) of every parameter at sample_index + its l2 norm; buffers in full
    for tag, net in (("prep|", prep), ("crnn|", crnn)):
        for kname, v in net.state_dict().items():
            if mo.is_buffer(kname):
                if v.is_floating_point():
                    out["post|" + tag + kname] = v.numpy().copy()
                continue
            d = (v.detach().double() - pre[tag + kname].double()).flatten()
            out["upd|" + tag + kname + "|s"] = d[sample_index(d.numel())].numpy().copy()
            out["upd|" + tag + kname + "|l2"] = d.norm().item()
    out.update({"x": x.numpy(), "labels": np.array(labels), "names": np.array(names), "cers": np.array([cers[n] for n in names]),
                "ws": ws, "inner": inner, "prop": prop, "std": std})
    np.savez_compressed(os.path.join(HERE, "area_step_b8.npz"), **out)
    print("area_step", losses, lossB.item(), idx.tolist(), out["B|decoded"])


def make_tracking():
    """tests/golden/tracking_b6.npz — the label-history branch (train_nn_patch.py:280-287) run with the REFERENCE's own
    tracking_utils.py (imported unchanged) and DecayingWeightGenerator (label_tracking/tracking_methods.py:105-115,
    compiled from the reference file: the module needs the absent python-Levenshtein): a 3-epoch OCR-label history with
    ragged depth per strip, window 3, decay 0.7 -> generate_ctc_target_batches -> weighted_ctc_loss on the reference CRNN
    (train-mode BN), loss + every CRNN gradient in fp64.  Four image candidates c0..c3 (first seeds whose fp32 run agrees with
    the fp64 run to COND_MAX on every tensor), same history for all."""
    import abc
    import types
    import tracking_utils as rtu                                 # reference, root module
    src = open(os.path.join(REF, "label_tracking", "tracking_methods.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("LossWeightGenerator", "DecayingWeightGenerator")]
    ns = {"torch": torch, "ABCMeta": abc.ABCMeta, "abstractmethod": abc.abstractmethod}
    exec(compile(ast.Module(body=body, type_ignores=[]), "tracking_methods.py", "exec"), ns)
    out = {}
    B, ws, window, decay, keep = 6, 40, 3, 0.7, 4
    names = [f"s{i}" for i in range(B)]
    hist = [synth_labels(B, 60 + e, 1, 9) for e in range(3)]     # epoch 0, 1, 2 OCR labels
    depth = [3, 1, 2, 3, 0, 2]                                   # how many past epochs each strip was queried in (0: new now)
    c2i = {c: i for i, c in enumerate(properties.char_set)}
    current = synth_labels(B, 70, 1, 9)                          # this iteration's OCR labels

    def run(dt, x):
        _, crnn = _ref_models(ws, dt)
        crnn.train()
        self = types.SimpleNamespace(device=torch.device("cpu"), crnn_model=crnn, char_to_index=c2i, window_size=window,
                                     weightgen_method="decaying", primary_loss_fn=torch.nn.CTCLoss(),
                                     primary_loss_fn_sample_wise=torch.nn.CTCLoss(reduction="none"),
                                     tracked_labels={n: [hist[e][i] for e in range(3)][:depth[i]] for i, n in enumerate(names)})
        wg = ns["DecayingWeightGenerator"](types.SimpleNamespace(decay_factor=decay, window_size=window), self.device)
        loss_weights = wg.gen_weights(self.tracked_labels, names)
        rtu.add_labels_to_history(self, names, current)
        batches = rtu.generate_ctc_target_batches(self, names)
        scores, pred_size = rtu.call_crnn(self, x.to(dt))
        loss = rtu.weighted_ctc_loss(self, scores, pred_size, batches, loss_weights)
        loss.backward()
        return crnn, loss.item(), scores.detach(), batches, loss_weights, self.tracked_labels

    k, xs = 0, 5100
    while k < keep:
        x = torch.rand(B, 1, 32, 128, generator=torch.Generator().manual_seed(xs))
        crnn32, l32, _, batches, lw, tracked = run(torch.float32, x)
        crnn64, l64, sc64, _, _, _ = run(torch.float64, x)
        dev = _worst_rel(_grads(crnn32.named_parameters()), _grads(crnn64.named_parameters()), skip=ZERO_GRAD)
        print("tracking candidate seed", xs, "fp32 vs fp64", dev)
        xs += 1
        if dev > COND_MAX:
            continue
        c = f"c{k}|"
        full64(crnn64.named_parameters(), c + "g|", out)
        out.update({c + "x": x.numpy(), c + "loss32": l32, c + "loss64": l64, c + "lp64": sc64.numpy(), c + "dev32": dev, c + "xs": xs - 1})
        k += 1
    out.update({"weights": lw.numpy(), "window": window, "decay": decay, "ws": ws, "names": np.array(names), "current": np.array(current),
                "history_json": np.array(json.dumps({n: [hist[e][i] for e in range(3)][:depth[i]] for i, n in enumerate(names)})),
                "tracked_after_json": np.array(json.dumps(tracked)), "n_batches": len(batches), "n_candidates": keep})
    for i, (t, ts, idx) in enumerate(batches):
        out[f"batch{i}|target"], out[f"batch{i}|size"], out[f"batch{i}|idx"] = t.numpy(), ts.numpy(), np.array(idx)
    np.savez_compressed(os.path.join(HERE, "tracking_b6.npz"), **out)
    print("tracking", l32, l64, [len(b[2]) for b in batches], lw.tolist())


def _unit_cost_distance(a, b):
    """python-Levenshtein 0.12.0's `distance` (requirements.txt:70; not installed): classic unit-cost edit distance."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def make_tracking_variants():
    """tests/golden/tracking_f3.npz — SURVEY §8 f3: the NON-decaying weight generators and the sample-wise weighted CTC.
    LevenshteinWeightGenerator / AttentionWeightGenerator (label_tracking/tracking_methods.py:26-101) are compiled from the
    reference file with `ast` (the module imports the absent python-Levenshtein: its `distance` is restated as the classic
    unit-cost edit distance and handed in under the name the class body uses), HistoryAttention is the reference's
    models/model_attention.py imported unchanged, and tracking_utils.py is the reference's root module: gen_weights ->
    add_labels_to_history -> generate_ctc_target_batches -> call_crnn -> weighted_ctc_loss with
    CTCLoss(reduction="none") (train_nn_area.py:147) on the reference CRNN (train-mode BN).  Per method: the weight table,
    the target batches, and for two image candidates the fp64 loss, log-probs and CRNN gradients; the attention model's
    state_dict travels with the fixture (it is never optimised: label_tracking/tracking_methods.py:35)."""
    import abc
    import types
    import tracking_utils as rtu
    from models.model_attention import HistoryAttention
    src = open(os.path.join(REF, "label_tracking", "tracking_methods.py")).read()
    tree = ast.parse(src)
    want = ("LossWeightGenerator", "AttentionWeightGenerator", "LevenshteinWeightGenerator")
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in want]
    ns = {"torch": torch, "ABCMeta": abc.ABCMeta, "abstractmethod": abc.abstractmethod, "tracking_utils": rtu, "properties": properties,
          "HistoryAttention": HistoryAttention, "Levenshtein": types.SimpleNamespace(distance=_unit_cost_distance)}
    exec(compile(ast.Module(body=body, type_ignores=[]), "tracking_methods.py", "exec"), ns)
    out = {}
    B, ws, window, keep = 6, 44, 3, 2
    names = [f"s{i}" for i in range(B)]
    # histories with agreement (repeated / near-repeated labels) so that the Levenshtein weights are not all zero
    base = synth_labels(B, 160, 2, 9)
    near = [w[:-1] + "x" for w in base]
    hist = [base, near, base]
    depth = [3, 1, 2, 3, 0, 2]
    c2i = {c: i for i, c in enumerate(properties.char_set)}
    current = [base[0], near[1], base[2], synth_labels(1, 171, 1, 9)[0], base[4], near[5]]
    history = {n: [hist[e][i] for e in range(3)][:depth[i]] for i, n in enumerate(names)}
    targs = types.SimpleNamespace(window_size=window, query_dim=8, emb_dim=16, attn_activation="sigmoid")
    torch.manual_seed(1234)
    gens = {"levenshtein": ns["LevenshteinWeightGenerator"](targs, torch.device("cpu")),
            "self_attention": ns["AttentionWeightGenerator"](targs, torch.device("cpu"), c2i)}
    att = gens["self_attention"].attention_model
    with torch.no_grad():                                        # the reference leaves the positional encodings at zero and never
        att.positional_encodings.normal_(0, 0.3)                 # trains them; non-zero values exercise the addition
    for k_, v in att.state_dict().items():
        out["att|" + k_] = v.numpy().copy()
    for act in ("softmax", "relu"):                              # the other two activations: weight tables only
        att.activation = act
        out[f"weights|self_attention_{act}"] = gens["self_attention"].gen_weights({n: list(v) for n, v in history.items()}, names).detach().numpy()
    att.activation = "sigmoid"

    def run(method, dt, x):
        _, crnn = _ref_models(ws, dt)
        crnn.train()
        self = types.SimpleNamespace(device=torch.device("cpu"), crnn_model=crnn, char_to_index=c2i, window_size=window,
                                     weightgen_method=method, primary_loss_fn=torch.nn.CTCLoss(),
                                     primary_loss_fn_sample_wise=torch.nn.CTCLoss(reduction="none"),
                                     tracked_labels={n: list(v) for n, v in history.items()})
        lw = gens[method].gen_weights(self.tracked_labels, names).detach()
        rtu.add_labels_to_history(self, names, current)
        batches = rtu.generate_ctc_target_batches(self, names)
        scores, pred_size = rtu.call_crnn(self, x.to(dt))
        loss = rtu.weighted_ctc_loss(self, scores, pred_size, batches, lw.to(dt))
        loss.backward()
        return crnn, loss.item(), scores.detach(), batches, lw, self.tracked_labels

    for method in ("levenshtein", "self_attention"):
        k, xs = 0, 5300
        while k < keep:
            x = torch.rand(B, 1, 32, 128, generator=torch.Generator().manual_seed(xs))
            crnn32, l32, _, batches, lw, tracked = run(method, torch.float32, x)
            crnn64, l64, sc64, _, _, _ = run(method, torch.float64, x)
            dev = _worst_rel(_grads(crnn32.named_parameters()), _grads(crnn64.named_parameters()), skip=ZERO_GRAD)
            print("tracking", method, "candidate seed", xs, "fp32 vs fp64", dev, "loss", l64)
            xs += 1
            if dev > COND_MAX:
                continue
            c = f"{method}|c{k}|"
            full64(crnn64.named_parameters(), c + "g|", out)
            out.update({c + "x": x.numpy(), c + "loss32": l32, c + "loss64": l64, c + "lp64": sc64.numpy(), c + "xs": xs - 1})
            k += 1
        out["weights|" + method] = lw.numpy()
        out[f"{method}|n_batches"] = len(batches)
        for i, (t, ts, idx) in enumerate(batches):
            out[f"{method}|batch{i}|target"], out[f"{method}|batch{i}|size"], out[f"{method}|batch{i}|idx"] = t.numpy(), ts.numpy(), np.array(idx)
        print("tracking", method, "weights", lw.tolist())
    out.update({"window": window, "ws": ws, "names": np.array(names), "current": np.array(current), "n_candidates": keep,
                "query_dim": targs.query_dim, "emb_dim": targs.emb_dim, "history_json": np.array(json.dumps(history)),
                "tracked_after_json": np.array(json.dumps(tracked))})
    np.savez_compressed(os.path.join(HERE, "tracking_f3.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["unet", "crnn", "ctc", "topk", "helpers", "step", "conditioned", "area_step", "tracking", "crop_oversize", "ladder", "tracking_variants"]
    for w in which:
        globals()["make_" + w]()
