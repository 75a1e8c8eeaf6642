"""Helpers of the training loop — drop-in for the hot subset of the reference's utils.py
(get_char_maps :22-40, pred_to_string :74-92, compare_labels :95-110, set_bn_eval :113-115,
padder / get_text_stack :118-141, get_ocr_helper :180-188, create_dirs :191-206, save_json /
save_all_jsons :209-231, handle_optuna_trial :233-237, set_random_seeds :240-243).

GPU-side work goes through the HIP library: greedy CTC decode (one wave per sample instead of a
Python loop with an .item() per (b,t)), crop + pad gather with a scatter-add backward.  The optional
third-party pieces of the reference file (wandb, optuna, unidecode, Levenshtein, torchvision) are
imported lazily and only where the corresponding feature is used."""
import json
import os
import random as python_random

import numpy as np
import torch

import properties


def get_char_maps(vocabulary=None):
    if vocabulary is None:
        import string
        vocabulary = ["-"] + list(string.ascii_lowercase) + list(string.ascii_uppercase) + list(string.digits)
    char_to_index = {c: i for i, c in enumerate(vocabulary)}
    index_to_char = {i: c for i, c in enumerate(vocabulary)}
    return char_to_index, index_to_char, len(vocabulary)


# ----------------------------------------------------------------------------- decode / CER
def pred_to_string(scores, labels, index_to_char, show_text=False):
    """Greedy CTC decode of scores [T,B,C]: per-step argmax, collapse repeats, drop index 0."""
    T, B, C = scores.shape
    if scores.is_cuda:
        from qea import ops
        s = scores.detach()
        if s.stride(2) != 1:
            s = s.contiguous()
        tokens = torch.empty(B, T, dtype=torch.int32, device=s.device)
        lengths = torch.empty(B, dtype=torch.int32, device=s.device)
        ops.greedy_decode(s, s.stride(0), s.stride(1), T, B, C, 0, tokens, lengths)
        tk, ln = tokens.cpu().tolist(), lengths.cpu().tolist()
        preds = ["".join(index_to_char[i] for i in tk[b][:ln[b]]) for b in range(B)]
    else:
        idx = scores.detach().argmax(dim=2).t().tolist()
        preds = []
        for row in idx:
            out, prev = "", None
            for k in row:
                if k != 0 and (len(out) == 0 or k != prev):
                    out += index_to_char[k]
                prev = k
            preds.append(out)
    if show_text:
        for l, p in zip(labels, preds):
            print(l, " -> ", p)
    return preds


def levenshtein(a, b):
    """Unit-cost edit distance (what python-Levenshtein's `distance` returns)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def compare_labels(preds, labels):
    if not isinstance(labels, (list, tuple)):
        labels = [labels]
    correct, total_cer = 0, 0
    for p, l in zip(preds, labels):
        correct += int(p == l)
        total_cer += levenshtein(l, p) / max(1, len(l))
    return correct, total_cer


def batch_cers(scores, labels, char_to_index, blank=0):
    """Per-sample CER of the greedy decode of `scores` [T,B,C] against `labels`, entirely on the GPU up to
    the final division: decode kernel -> edit-distance kernel -> one [B] int32 copy to the host.  Equals
    [compare_labels([pred_i], [label_i])[1] for i] of the reference loop (train_nn_patch.py:336-342)."""
    from qea import ops
    T, B, C = scores.shape
    s = scores.detach()
    if s.stride(2) != 1:
        s = s.contiguous()
    dev = s.device
    tokens = torch.empty(B, T, dtype=torch.int32, device=dev)
    plen = torch.empty(B, dtype=torch.int32, device=dev)
    ops.greedy_decode(s, s.stride(0), s.stride(1), T, B, C, blank, tokens, plen)
    glen = torch.tensor([len(l) for l in labels], dtype=torch.int32)
    goff = torch.zeros(B, dtype=torch.int64)
    if B > 1:
        goff[1:] = torch.cumsum(glen.to(torch.int64), 0)[:-1]
    flat = [char_to_index[c] for c in "".join(labels)] or [0]
    gt = torch.tensor(flat, dtype=torch.int32)
    dist = torch.empty(B, dtype=torch.int32, device=dev)
    ops.edit_distance(tokens, T, plen, gt.to(dev), goff.to(dev), glen.to(dev), B, dist)
    d = dist.cpu().tolist()
    return [d[i] / max(1, len(labels[i])) for i in range(B)]


def set_bn_eval(module):
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
        module.eval()


# ----------------------------------------------------------------------------- crop + pad
class _CropPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, boxes_dev, n, oh, ow):
        from qea import ops
        _, H, W = image.shape
        out = torch.empty(n, 1, oh, ow, device=image.device)
        ops.crop_pad_gather(image.contiguous(), H, W, boxes_dev, n, oh, ow, out)
        ctx.save_for_backward(boxes_dev)
        ctx.dims = (H, W, n, oh, ow)
        return out

    @staticmethod
    def backward(ctx, dout):
        from qea import ops
        (boxes_dev,) = ctx.saved_tensors
        H, W, n, oh, ow = ctx.dims
        dimg = torch.zeros(1, H, W, device=dout.device)
        ops.crop_pad_scatter(dout.contiguous(), boxes_dev, n, oh, ow, dimg, H, W)
        return dimg, None, None, None, None


def padder(crop, h, w):
    _, c_h, c_w = crop.shape
    left, top = (w - c_w) // 2, (h - c_h) // 2
    return torch.nn.functional.pad(crop, (left, w - left - c_w, top, h - top - c_h), value=1.0)


def get_text_stack(image, labels, input_size):
    """image [1,H,W]; labels: list of dicts with label,x_min,y_min,x_max,y_max -> ([N,1,h,w], [label])."""
    names = [l["label"] for l in labels]
    if image.is_cuda and image.shape[0] == 1 and labels:
        _, H, W = image.shape
        # python slicing clips to the image; the kernel takes pre-clipped boxes
        boxes = torch.tensor([[max(0, l["x_min"]), max(0, l["y_min"]), min(W, l["x_max"]), min(H, l["y_max"])] for l in labels],
                             dtype=torch.int32)
        return _CropPad.apply(image, boxes.to(image.device), len(labels), input_size[0], input_size[1]), names
    crops = [padder(image[:, l["y_min"]:l["y_max"], l["x_min"]:l["x_max"]], *input_size) for l in labels]
    return torch.stack(crops), names


# ----------------------------------------------------------------------------- OCR / dirs / json / seeds
def get_ununicode(text):
    for a, b in (("_", "-"), ("`", "'"), ("©", "c"), ("°", "'"), ("£", "E"), ("§", "S")):
        text = text.replace(a, b)
    from unidecode import unidecode          # only needed by the real OCR engines
    has_eur = "€" in text
    text = unidecode(text.replace("€", "<eur>"))
    return text.replace("<eur>", "€") if has_eur else text


def get_ocr_helper(ocr, is_eval=False):
    """Same names as the reference (:180-188).  The real engines are optional third-party black boxes;
    'stub' is the deterministic stand-in with the identical get_labels contract."""
    if ocr == "stub":
        from ocr_helper.stub_helper import StubHelper
        return StubHelper(is_eval=is_eval)
    if ocr in ("Tesseract", "EasyOCR", "gvision"):
        mod = {"Tesseract": "tess_helper", "EasyOCR": "eocr_helper", "gvision": "gcloud_helper"}[ocr]
        cls = {"Tesseract": "TessHelper", "EasyOCR": "EocrHelper", "gvision": "GcloudHelper"}[ocr]
        try:
            m = __import__(f"ocr_helper.{mod}", fromlist=[cls])
        except ImportError as e:
            raise ImportError(f"OCR engine {ocr!r} needs its third-party package and an ocr_helper/{mod}.py adapter "
                              f"(black box, out of scope here); use --ocr stub for plumbing runs") from e
        return getattr(m, cls)(is_eval=is_eval)
    return None


def create_dirs(self, args):
    self.crnn_model_path = args.crnn_model
    self.prep_model_path = args.prep_model
    self.data_base_path = args.data_base_path
    self.exp_base_path = args.exp_base_path
    sub = lambda s: os.path.join(self.exp_base_path, s)
    self.ckpt_base_path = sub(properties.prep_crnn_ckpts)
    self.cers_base_path = sub("cers")
    self.tracked_labels_path = sub("tracked_labels")
    self.selectedsamples_path = sub("selected_samples")
    self.img_out_path = sub(properties.img_out)
    for d in (self.exp_base_path, self.ckpt_base_path, self.img_out_path, self.cers_base_path, self.tracked_labels_path,
              self.selectedsamples_path):
        os.makedirs(d, exist_ok=True)


def _wandb():
    try:
        import wandb
        return wandb if wandb.run is not None else None
    except ImportError:
        return None


def save_json(metrics, json_path, wandb_save=True):
    with open(json_path, "w") as f:
        json.dump(metrics, f)
    wb = _wandb()
    if wandb_save and wb is not None:
        wb.save(json_path)


def save_all_jsons(self, epoch):
    save_json(self.tracked_labels, os.path.join(self.tracked_labels_path, f"tracked_labels_{epoch}.json"), wandb_save=False)
    save_json(self.tracked_labels, os.path.join(self.tracked_labels_path, "tracked_labels_current.json"))
    save_json(self.selected_samples, os.path.join(self.selectedsamples_path, "selected_samples_current.json"))
    save_json(self.sampler.all_cers, os.path.join(self.cers_base_path, "all_cers.json"))


def save_img(images, name, dir, nrow=8):
    """Grid PNG of [N,1,H,W] images in [0,1] (reference :43-46 via torchvision.make_grid, 2-px padding)."""
    from PIL import Image
    imgs = images.detach().cpu().float().clamp(0, 1)
    n, _, h, w = imgs.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    pad = 2
    grid = torch.zeros(rows * (h + pad) + pad, cols * (w + pad) + pad)
    for i in range(n):
        r, c = divmod(i, cols)
        grid[pad + r * (h + pad): pad + r * (h + pad) + h, pad + c * (w + pad): pad + c * (w + pad) + w] = imgs[i, 0]
    Image.fromarray((grid.numpy() * 255).round().astype(np.uint8)).save(os.path.join(dir, name + ".png"), "PNG")


def handle_optuna_trial(trial, accuracy, epoch):
    if trial is not None:
        trial.report(accuracy, epoch)
        if trial.should_prune():
            import optuna
            raise optuna.TrialPruned()


def set_random_seeds(random_seed):
    torch.manual_seed(random_seed)
    python_random.seed(random_seed)
    np.random.seed(random_seed)
