"""Document images with word boxes on disk — same files and sample tuple as the reference's
datasets/patch_dataset.py:14-129: `<name>.png|jpg|jpeg` + `<name>.json` (a list of
{label, x1..x4, y1..y4} or {label, x_min, y_min, x_max, y_max}); images are centred on a white
400x512 canvas (pad=True), boxes shifted accordingly, boxes wider than 127 / taller than 31 px or with
labels over max_char_len are dropped; a document left without boxes gets one blank placeholder box.
sample = (image [1,400,512], [box dicts], path)."""
import json
import random

import torch
from PIL import Image, ImageOps
from torch.utils.data import Dataset

import properties
from datasets._io import list_images, to_tensor

CANVAS = (400, 512)   # (height, width)


class PatchDataset(Dataset):
    def __init__(self, data_dir, pad=False, include_name=False, num_subset=None, resize_images=False):
        self.pad, self.include_name, self.resize_images = pad, include_name, resize_images
        self.files = list_images(data_dir)
        self.size = CANVAS

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        path = self.files[idx]
        image = Image.open(path).convert("L")
        w, h = image.size
        top, left, sx, sy = 0, 0, 1, 1
        if self.pad:
            H, W = self.size
            if h <= H or w <= W:
                dh, dw = H - h, W - w
                top, left = dh // 2, dw // 2
                image = ImageOps.expand(image, (left, top, dw - left, dh - top), fill=255)
            elif self.resize_images:
                image = image.resize((W, H), Image.BILINEAR)
                sy, sx = H / h, W / w
            else:
                print("Height screwed", idx)
        boxes = self.coord_loader(path, top, left, sx, sy)
        return (to_tensor(image), boxes, path) if self.include_name else (to_tensor(image), boxes)

    def coord_loader(self, img_path, top_padding=0, left_padding=0, resize_w=1, resize_h=1):
        with open(img_path.rsplit(".", 1)[0] + ".json", "r") as f:
            areas = json.load(f)
        quad = bool(areas) and "x1" in areas[0]
        out = []
        for i, a in enumerate(areas):
            if quad:
                xs = [a[f"x{k}"] + left_padding for k in (1, 2, 3, 4)]
                ys = [a[f"y{k}"] + top_padding for k in (1, 2, 3, 4)]
                x_min, x_max = int(min(xs) * resize_w), int(max(xs) * resize_w)
                y_min, y_max = int(min(ys) * resize_h), int(max(ys) * resize_h)
            else:
                x_min, x_max = a["x_min"] + left_padding, a["x_max"] + left_padding
                y_min, y_max = a["y_min"] + top_padding, a["y_max"] + top_padding
                xs, ys = [x_min, x_max, x_max, x_min], [y_min, y_min, y_max, y_max]
            if len(a["label"]) <= properties.max_char_len and x_max - x_min < 128 and y_max - y_min < 32:
                box = {"label": a["label"], "x_min": x_min, "y_min": y_min, "x_max": x_max, "y_max": y_max, "index": i}
                box.update({f"x{k + 1}": xs[k] for k in range(4)})
                box.update({f"y{k + 1}": ys[k] for k in range(4)})
                out.append(box)
        if not out:
            out.append({"label": properties.empty_char, "x_min": 0, "y_min": 0, "x_max": 127, "y_max": 31, "index": 0})
        return out

    def shuffle(self):
        random.shuffle(self.files)

    @staticmethod
    def collate(data):
        cols = list(zip(*data))
        res = [torch.stack(cols[0]), list(cols[1])]
        if len(cols) == 3:
            res.append(list(cols[2]))
        return res
