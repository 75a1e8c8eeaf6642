"""Deterministic stand-in for the black-box OCR (same contract as reference TessHelper,
ocr_helper/tess_helper.py:10-44): labels are a pure function of the image content, use only
char_set symbols, '' -> empty_char, never longer than max_char_len, count_calls is maintained."""
import zlib

import properties


class StubHelper:
    def __init__(self, empty_char=properties.empty_char, is_eval=False, label_source=None):
        self.empty_char, self.is_eval = empty_char, is_eval
        self.count_calls = 0
        self.label_source = label_source       # optional callable(imgs) -> list[str] (e.g. ground truth with edits)

    def _label(self, img):
        # 8 bins of column ink mass -> up to 8 symbols; depends smoothly on the image so that jitter
        # replicas of one strip mostly (not always) agree, like a real recogniser
        ink = (1.0 - img.float()).clamp_(0, 1).sum(dim=(0, 1))                      # [W]
        bins = ink.reshape(8, -1).sum(1)
        chars = []
        for i, b in enumerate(bins.tolist()):
            q = int(b * 4)
            if q > 0:
                chars.append(properties.char_set[1 + (zlib.crc32(bytes([i, q % 251])) % 94)])
        return "".join(chars)

    def get_labels(self, imgs):
        imgs = imgs.detach().cpu()
        labels = self.label_source(imgs) if self.label_source else [self._label(imgs[i]) for i in range(imgs.shape[0])]
        labels = [(l if l != "" and len(l) <= properties.max_char_len else self.empty_char) for l in labels]
        self.count_calls += len(labels)
        return labels

    def get_string(self, img):
        return self._label(img.detach().cpu()).split()
