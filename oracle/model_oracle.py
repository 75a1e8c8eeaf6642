"""TEST INFRASTRUCTURE ONLY — CPU oracle for the UNet cleaner and the CRNN proxy.

A functional torch-CPU restatement (F.conv2d / F.batch_norm / torch._VF.lstm ... over a flat
{state_dict key: tensor} mapping) of the two networks of the hot path:

  * UNet      — reference models/model_unet.py:7-76 (forward) and :78-109 (_block)
  * CRNN      — reference models/model_crnn.py:5-32; backbone :34-56

The arithmetic itself lives in ATen (torch 1.8.0 pinned by the reference's
requirements.txt:94; torch 2.10 CPU here), not under /root/reference.  The restatement is
pinned against outputs of the reference modules themselves run in the build container:
tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (query-efficient-approx-to-improve-ocr_amd/) never does.
"""
import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default, models/model_unet.py:92,105; model_crnn.py:42,44
BN_MOMENTUM = 0.1

UNET_LEVELS = (("encoder1", "enc1"), ("encoder2", "enc2"), ("encoder3", "enc3"), ("encoder4", "enc4"),
               ("bottleneck", "bottleneck"), ("decoder4", "dec4"), ("decoder3", "dec3"),
               ("decoder2", "dec2"), ("decoder1", "dec1"))


# ----------------------------------------------------------------------------- state layout
def unet_state_shapes(in_ch=1, out_ch=1, f=32):
    """state_dict key -> shape, same keys/order as the reference UNet."""
    sd = OrderedDict()

    def block(mod, name, cin, cout):
        for i, ci in ((1, cin), (2, cout)):
            sd[f"{mod}.{name}conv{i}.weight"] = (cout, ci, 3, 3)
            sd[f"{mod}.{name}norm{i}.weight"] = (cout,)
            sd[f"{mod}.{name}norm{i}.bias"] = (cout,)
            sd[f"{mod}.{name}norm{i}.running_mean"] = (cout,)
            sd[f"{mod}.{name}norm{i}.running_var"] = (cout,)
            sd[f"{mod}.{name}norm{i}.num_batches_tracked"] = ()

    block("encoder1", "enc1", in_ch, f)
    block("encoder2", "enc2", f, 2 * f)
    block("encoder3", "enc3", 2 * f, 4 * f)
    block("encoder4", "enc4", 4 * f, 8 * f)
    block("bottleneck", "bottleneck", 8 * f, 16 * f)
    for lvl, c in ((4, 8 * f), (3, 4 * f), (2, 2 * f), (1, f)):
        sd[f"upconv{lvl}.weight"] = (2 * c, c, 2, 2)
        sd[f"upconv{lvl}.bias"] = (c,)
        block(f"decoder{lvl}", f"dec{lvl}", 2 * c, c)
    sd["conv.weight"] = (out_ch, f, 1, 1)
    sd["conv.bias"] = (out_ch,)
    return sd


def crnn_state_shapes(vocab=95):
    sd = OrderedDict()
    for layer, cin in ((0, 512), (1, 512)):
        for suf in ("", "_reverse"):
            sd[f"lstm.weight_ih_l{layer}{suf}"] = (1024, cin)
            sd[f"lstm.weight_hh_l{layer}{suf}"] = (1024, 256)
            sd[f"lstm.bias_ih_l{layer}{suf}"] = (1024,)
            sd[f"lstm.bias_hh_l{layer}{suf}"] = (1024,)
    sd["linear.weight"] = (vocab, 512)
    sd["linear.bias"] = (vocab,)
    chans = ((1, 1, 64, 3), (2, 64, 128, 3), (3, 128, 256, 3), (4, 256, 256, 3), (5, 256, 512, 3))
    for i, ci, co, k in chans:
        sd[f"convo.conv{i}.weight"] = (co, ci, k, k)
        sd[f"convo.conv{i}.bias"] = (co,)
    sd["convo.batchnorm1.weight"] = (512,)
    sd["convo.batchnorm1.bias"] = (512,)
    sd["convo.batchnorm1.running_mean"] = (512,)
    sd["convo.batchnorm1.running_var"] = (512,)
    sd["convo.batchnorm1.num_batches_tracked"] = ()
    sd["convo.conv6.weight"] = (512, 512, 3, 3)
    sd["convo.conv6.bias"] = (512,)
    sd["convo.batchnorm2.weight"] = (512,)
    sd["convo.batchnorm2.bias"] = (512,)
    sd["convo.batchnorm2.running_mean"] = (512,)
    sd["convo.batchnorm2.running_var"] = (512,)
    sd["convo.batchnorm2.num_batches_tracked"] = ()
    sd["convo.conv7.weight"] = (512, 512, 2, 2)
    sd["convo.conv7.bias"] = (512,)
    return sd


def seeded_state(shapes, seed):
    """Deterministic, order-independent fill keyed by parameter NAME, so the reference module,
    this oracle and the HIP modules can all be given bit-identical weights without shipping
    them: value stream = torch CPU Generator seeded with crc32(name) ^ seed."""
    out = OrderedDict()
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.long)
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1 and ("norm" in name):
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if leaf == "weight" else 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            if name.startswith("upconv"):          # ConvTranspose2d weight [Cin, Cout, 2, 2]
                fan_in = shape[0]
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = shape[1]
            gain = 1.0 if name.startswith(("lstm", "linear")) else math.sqrt(2.0)
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        out[name] = t
    return out


def default_init_state(shapes, seed):
    """Name-keyed fill with the DISTRIBUTIONS of torch.nn's default initialisation (what `UNet()` /
    `CRNN(95, False)` get in the reference when no checkpoint is given, train_nn_patch.py:90-99):
    conv / linear weights and biases U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (kaiming_uniform_(a=sqrt 5)),
    ConvTranspose2d fan_in = C_out*kh*kw, LSTM tensors U(-1/sqrt(hidden), +), BatchNorm weight 1 / bias 0 /
    running_mean 0 / running_var 1.  The value stream is keyed by the parameter name like seeded_state(),
    so the reference modules, this oracle and the HIP modules can be given bit-identical weights
    without shipping 66 MB of them.  Networks initialised this way are WELL conditioned (fp32-vs-fp64
    gradient deviation of the reference ~1e-6, tests/golden/make_golden.py::make_conditioned), unlike
    seeded_state()'s He-normal weights with random running statistics."""
    out = OrderedDict()
    bias_fan = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2246822519 + 977)) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        stem = name.rsplit(".", 1)[0]
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.long)
        elif leaf == "running_var":
            t = torch.ones(shape)
        elif leaf == "running_mean":
            t = torch.zeros(shape)
        elif "norm" in name and len(shape) == 1:
            t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif name.startswith("lstm."):
            bound = 1.0 / math.sqrt(256.0)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            if len(shape) >= 2:
                fan_in = (shape[1] * shape[2] * shape[3]) if len(shape) == 4 else shape[1]
                bias_fan[stem] = fan_in
            else:
                fan_in = bias_fan[stem]                      # the bias follows its weight in the state order
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        out[name] = t
    return out


def is_buffer(name):
    return name.rsplit(".", 1)[-1] in ("running_mean", "running_var", "num_batches_tracked")


def split_state(state, requires_grad=True):
    """-> (params with requires_grad, buffers) as two OrderedDicts of fresh tensors."""
    params, bufs = OrderedDict(), OrderedDict()
    for k, v in state.items():
        if is_buffer(k):
            bufs[k] = v.clone()
        else:
            params[k] = v.clone().requires_grad_(requires_grad)
    return params, bufs


# ----------------------------------------------------------------------------- forward passes
class Trace:
    """Side channel of one forward pass (test infrastructure of the decision-conditioned gradient gate).

    rec   : dict filled with the detached tensor at every named site — conv / BN / transposed-conv outputs (the BN output
            is the PRE-activation of the ReLU behind it), max-pool inputs, LSTM layer outputs, logits.
    force : {site: decision} imposed on the non-differentiable points of the networks instead of taking them from this
            pass's own values: a ReLU site gets a boolean mask (True = passes), a max-pool site the int64 window winners in
            F.max_pool2d(return_indices=True) form (flat index into the H*W plane of the pool's input).  With the decisions
            of ANOTHER implementation's forward imposed, this pass evaluates the function that implementation
            differentiates, so its fp64 backward is the exact gradient that implementation's backward approximates
            (the networks are smooth everywhere else: conv, BN, sigmoid/tanh gates, log_softmax, CTC)."""

    def __init__(self, force=None, record=True):
        self.force = force
        self.rec = {} if record else None


def _tap(x, site, tr):
    if tr is not None and tr.rec is not None:
        tr.rec[site] = x.detach()
    return x


def _relu(x, site, tr):
    if tr is None or tr.force is None:
        return F.relu(x)
    return x * tr.force[site].to(x.dtype)


def _pool(x, k, site, tr):
    _tap(x, site, tr)
    if tr is None or tr.force is None:
        return F.max_pool2d(x, k)
    idx = tr.force[site]
    return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


POOL_SITES = {"pool1": (2, 2), "pool2": (2, 2), "pool3": (2, 2), "pool4": (2, 2),                       # UNet (model_unet.py:14-20)
              "convo.pool1": (2, 2), "convo.pool2": (2, 2), "convo.pool4": (2, 1), "convo.pool6": (2, 1)}  # CRNN (model_crnn.py:48-54)
CRNN_BARE_RELU = ("convo.conv1", "convo.conv2", "convo.conv3", "convo.conv4")      # conv outputs that feed a ReLU directly


def is_preactivation(site):
    return site in CRNN_BARE_RELU or site.startswith("convo.batchnorm") or "norm" in site.rsplit(".", 1)[-1]


def own_decisions(rec):
    """The decisions a recorded pass took itself, in Trace.force form (ATen's rules: ReLU passes x > 0, a pool window keeps
    its first maximum in scan order)."""
    out = {}
    for site, t in rec.items():
        if is_preactivation(site):
            out[relu_site(site)] = t > 0
        elif site in POOL_SITES:
            out[site] = F.max_pool2d(t, POOL_SITES[site], return_indices=True)[1]
    return out


def relu_site(pre_site):
    """name of the ReLU behind a recorded pre-activation site: encoder1.enc1norm2 -> encoder1.enc1relu2,
    convo.conv3 -> convo.relu3, convo.batchnorm1 -> convo.relu5, convo.batchnorm2 -> convo.relu6"""
    head, leaf = pre_site.rsplit(".", 1)
    if head == "convo":
        return "convo.relu" + {"batchnorm1": "5", "batchnorm2": "6"}.get(leaf, leaf[-1])
    return head + "." + leaf.replace("norm", "relu")


def _bn(x, P, Bf, prefix, training):
    return F.batch_norm(x, Bf[prefix + ".running_mean"], Bf[prefix + ".running_var"],
                        P[prefix + ".weight"], P[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


def _bump(Bf, prefix, training):
    if training:
        Bf[prefix + ".num_batches_tracked"] += 1


def _unet_block(x, P, Bf, mod, name, training, tr=None):
    # conv3x3(pad 1, no bias) -> BN -> ReLU, twice (model_unet.py:78-109)
    for i in (1, 2):
        x = _tap(F.conv2d(x, P[f"{mod}.{name}conv{i}.weight"], None, padding=1), f"{mod}.{name}conv{i}", tr)
        x = _tap(_bn(x, P, Bf, f"{mod}.{name}norm{i}", training), f"{mod}.{name}norm{i}", tr)
        _bump(Bf, f"{mod}.{name}norm{i}", training)
        x = _relu(x, f"{mod}.{name}relu{i}", tr)
    return x


def unet_forward(P, Bf, x, training, trace=None):
    """model_unet.py:49-76.  `training` selects batch statistics (+ running-stat update)."""
    tr = trace
    e1 = _unet_block(x, P, Bf, "encoder1", "enc1", training, tr)
    e2 = _unet_block(_pool(e1, 2, "pool1", tr), P, Bf, "encoder2", "enc2", training, tr)
    e3 = _unet_block(_pool(e2, 2, "pool2", tr), P, Bf, "encoder3", "enc3", training, tr)
    e4 = _unet_block(_pool(e3, 2, "pool3", tr), P, Bf, "encoder4", "enc4", training, tr)
    d = _unet_block(_pool(e4, 2, "pool4", tr), P, Bf, "bottleneck", "bottleneck", training, tr)
    for lvl, skip in ((4, e4), (3, e3), (2, e2), (1, e1)):
        d = _tap(F.conv_transpose2d(d, P[f"upconv{lvl}.weight"], P[f"upconv{lvl}.bias"], stride=2), f"upconv{lvl}", tr)
        d = torch.cat((d, skip), dim=1)
        d = _unet_block(d, P, Bf, f"decoder{lvl}", f"dec{lvl}", training, tr)
    return _tap(torch.sigmoid(F.conv2d(d, P["conv.weight"], P["conv.bias"])), "img", tr)


def crnn_backbone(P, Bf, x, bn_training, trace=None):
    """model_crnn.py:47-56.  Pool sites of the first four layers sit behind a ReLU: their recorded input (and the
    imposed winners) refer to the ReLU's OUTPUT."""
    c, tr = "convo.", trace

    def conv(i, x, **kw):
        return _tap(F.conv2d(x, P[f"{c}conv{i}.weight"], P[f"{c}conv{i}.bias"], **kw), f"{c}conv{i}", tr)
    x = _pool(_relu(conv(1, x, padding=1), c + "relu1", tr), (2, 2), c + "pool1", tr)
    x = _pool(_relu(conv(2, x, padding=1), c + "relu2", tr), (2, 2), c + "pool2", tr)
    x = _relu(conv(3, x, padding=1), c + "relu3", tr)
    x = _pool(_relu(conv(4, x, padding=1), c + "relu4", tr), (2, 1), c + "pool4", tr)
    x = conv(5, x, padding=1)
    x = _relu(_tap(_bn(x, P, Bf, c + "batchnorm1", bn_training), c + "batchnorm1", tr), c + "relu5", tr)
    _bump(Bf, c + "batchnorm1", bn_training)
    x = conv(6, x, padding=1)
    x = _relu(_tap(_bn(x, P, Bf, c + "batchnorm2", bn_training), c + "batchnorm2", tr), c + "relu6", tr)
    _bump(Bf, c + "batchnorm2", bn_training)
    x = _pool(x, (2, 1), c + "pool6", tr)
    return conv(7, x)


def lstm_layer_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.LSTM layer, seq-first, zero initial state; gate order i,f,g,o."""
    T, B, _ = x.shape
    Hh = w_hh.shape[1]
    h = x.new_zeros(B, Hh)
    c = x.new_zeros(B, Hh)
    gx = x @ w_ih.t() + b_ih + b_hh
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gx[t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs)


def bilstm(P, x, trace=None):
    """nn.LSTM(512, 256, 2, bidirectional=True) (model_crnn.py:9,19)."""
    for layer in (0, 1):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_layer_dir(x, P[f"lstm.weight_ih_l{layer}{suf}"], P[f"lstm.weight_hh_l{layer}{suf}"],
                                       P[f"lstm.bias_ih_l{layer}{suf}"], P[f"lstm.bias_hh_l{layer}{suf}"], rev))
        x = torch.cat(outs, dim=2)
        if layer == 0:
            _tap(x, "lstm0", trace)
    return x


def crnn_forward(P, Bf, x, bn_training, nan_scrub=True, trace=None):
    """model_crnn.py:16-28 (+ the NaN-scrubbing backward hook of :30-32 as registered by
    train_nn_patch.py:94: NaNs in the gradient entering log_softmax's backward are zeroed)."""
    f = crnn_backbone(P, Bf, x, bn_training, trace)          # [B,512,1,W']
    b, ch, h, w = f.shape
    seq = f.permute(3, 0, 1, 2).reshape(w, b, ch * h)        # map_to_sequence :23-28 (H == 1)
    y = _tap(bilstm(P, seq, trace), "lstm1", trace)
    logits = _tap(y @ P["linear.weight"].t() + P["linear.bias"], "logits", trace)
    if nan_scrub and logits.requires_grad:
        logits.register_hook(lambda g: torch.nan_to_num(g, nan=0.0, posinf=float("inf"), neginf=float("-inf")))
    return F.log_softmax(logits, 2)
