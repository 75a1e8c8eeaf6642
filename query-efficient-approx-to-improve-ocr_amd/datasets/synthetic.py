"""Synthetic stand-ins for the reference's datasets with the SAME sample tuples, for plumbing runs,
tests and benchmarks (there is no network for the real POS / VGG data):

  * SyntheticTextAreas  ~ datasets/img_dataset.py:15-48   -> (image [1,32,128], label, name[, index])
  * SyntheticPatches    ~ datasets/patch_dataset.py:14-129 -> (image [1,400,512], [ {label,x_min,y_min,x_max,y_max}, ... ], name)

Images are POS-style: white background (1.0), dark strokes, light sensor noise."""
import numpy as np
import torch

import properties


def _strokes(shape, gen, density=0.12):
    m = (torch.rand(shape, generator=gen) < density).float()
    ink = torch.rand(shape, generator=gen) * 0.7 + 0.3
    return (1 - m * ink + 0.02 * torch.randn(shape, generator=gen)).clamp(0, 1)


def _label(rng, lo=1, hi=12):
    return "".join(properties.char_set[i] for i in rng.randint(1, 95, rng.randint(lo, hi + 1)))


class SyntheticTextAreas(torch.utils.data.Dataset):
    def __init__(self, n, seed=0, include_name=True, include_index=False, size=properties.input_size, widths=None):
        """widths: optional list of per-sample widths (multiples of 16) for variable-width lines (datasets/bucketing.py)"""
        self.n, self.seed, self.size = n, seed, size
        self.include_name, self.include_index = include_name, include_index
        self.widths = list(widths) if widths is not None else None
        rng = np.random.RandomState(seed)
        self.labels = [_label(rng) for _ in range(n)]
        self.names = [f"{i}_{self.labels[i]}_synthetic.png" for i in range(n)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        size = (self.size[0], self.widths[i]) if self.widths is not None else tuple(self.size)
        item = (_strokes((1,) + size, g), self.labels[i])
        if self.include_name:
            item += (self.names[i],)
        if self.include_index:
            item += (i,)
        return item


class SyntheticPatches(torch.utils.data.Dataset):
    def __init__(self, n, seed=0, strips=(4, 9), pad_shape=(400, 512), include_name=True):
        self.n, self.seed, self.strips, self.pad_shape, self.include_name = n, seed, strips, pad_shape, include_name

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        rng = np.random.RandomState(self.seed * 7919 + i)
        H, W = self.pad_shape
        img = torch.ones(1, H, W)
        boxes = []
        y = 8
        for _ in range(rng.randint(self.strips[0], self.strips[1] + 1)):
            h, w = rng.randint(14, 31), rng.randint(40, 127)          # h < 32, w < 128 as patch_dataset.py:95 filters
            x = rng.randint(4, W - w - 4)
            if y + h >= H - 4:
                break
            img[:, y:y + h, x:x + w] = _strokes((1, h, w), g)
            boxes.append(dict(label=_label(rng, 1, 10), x_min=int(x), y_min=int(y), x_max=int(x + w), y_max=int(y + h)))
            y += h + rng.randint(4, 12)
        item = (img, boxes)
        if self.include_name:
            item += (f"synthetic/folder{self.seed}/doc_{i:05d}.png",)
        return item

    @staticmethod
    def collate(data):
        images = [d[0] for d in data]
        labels = [d[1] for d in data]
        if len(data[0]) == 3:
            return torch.stack(images), labels, [d[2] for d in data]
        return torch.stack(images), labels
