"""Text-strip images on disk — same sample layout as the reference's datasets/img_dataset.py:15-48:
files `<idx>_<label>_<anything>.png|jpg`, sample = [image, label, file_name][, index]; labels longer
than max_char_len are dropped at listing time, `transform` receives the grey PIL image."""
import os

from PIL import Image
from torch.utils.data import Dataset

import properties
from datasets._io import ascii_label, list_images, to_tensor

_BROKEN_NAMES = ("22_✔_786.png", "162_✓_467.png", "26_✓_receipt_00627.png", "61_✓_145.png", "19__V_receipt_00188.png")


def _label_of(path):
    parts = os.path.basename(path).split("_")
    return parts[1] if len(parts) > 1 else ""


class ImgDataset(Dataset):
    def __init__(self, data_dir, transform=None, include_name=False, include_index=False):
        self.transform, self.include_name, self.include_index = transform, include_name, include_index
        self.files = [f for f in list_images(data_dir, ("png", "jpg"), exclude=_BROKEN_NAMES)
                      if len(_label_of(f)) <= properties.max_char_len]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        path = self.files[idx]
        image = Image.open(path).convert("L")
        image = self.transform(image) if self.transform is not None else to_tensor(image)
        label = ascii_label(_label_of(path))
        if len(label) > properties.max_char_len:
            label = properties.empty_char
        sample = [image, label]
        if self.include_name:
            sample.append(os.path.basename(path))
        if self.include_index:
            sample.append(idx)
        return sample
