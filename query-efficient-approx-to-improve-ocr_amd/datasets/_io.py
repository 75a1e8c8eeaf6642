"""Shared file helpers of the on-disk datasets (PIL + numpy only; no torchvision)."""
import os

import numpy as np
import torch

IMAGE_EXT = ("png", "jpg", "jpeg")


def list_images(root, extensions=IMAGE_EXT, exclude=()):
    """All image files under `root` (recursive), like the reference's utils.get_files (utils.py:162-171)."""
    out = []
    for dirpath, _, files in os.walk(root):
        for f in sorted(files):
            if f not in exclude and f.lower().endswith(tuple(extensions)):
                out.append(os.path.join(dirpath, f))
    return out


def to_tensor(pil_img):
    """8-bit grey PIL image -> float tensor [1,H,W] in [0,1] (what torchvision's ToTensor yields for mode 'L')."""
    return torch.from_numpy(np.asarray(pil_img.convert("L"), dtype=np.float32) / 255.0)[None].contiguous()


def ascii_label(text):
    """The reference normalises labels with unidecode (utils.py:49-71); without that optional package the label is
    used as it is (labels outside properties.char_set then fail at encoding time, exactly as in the reference)."""
    try:
        from utils import get_ununicode
        return get_ununicode(text)
    except ImportError:
        return text
