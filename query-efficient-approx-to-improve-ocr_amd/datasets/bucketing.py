"""Width-bucketed batching for variable-width text lines (BASELINE configs[4] as reinterpreted in SURVEY.md F8:
the reference itself pads or shrinks every strip to 32x128; UNet and CRNN are width-agnostic for W % 16 == 0 with
T = W/4 - 1, so wider lines can be trained without shrinking them).

Every sample is white-padded on the right to the smallest bucket that holds it; a batch only ever contains samples of
ONE bucket, so each batch is a dense [B,1,32,Wb] tensor and all its CTC inputs have the same length T(Wb)."""
import torch

BUCKETS = (128, 256, 384, 512)


def bucket_of(width, buckets=BUCKETS):
    for b in buckets:
        if width <= b:
            return b
    return buckets[-1]


def pad_to_bucket(img, buckets=BUCKETS):
    """img [1,32,w] -> [1,32,Wb], white (1.0) on the right; wider than the last bucket: cropped."""
    w = img.shape[-1]
    wb = bucket_of(w, buckets)
    if w >= wb:
        return img[..., :wb].contiguous()
    out = torch.ones(img.shape[:-1] + (wb,), dtype=img.dtype)
    out[..., :w] = img
    return out


class BucketBatchSampler(torch.utils.data.Sampler):
    """Yields index lists whose samples all fall into one width bucket (shuffled within and across buckets)."""

    def __init__(self, widths, batch_size, buckets=BUCKETS, drop_last=True, generator=None, seed=0):
        """generator=None: the sampler draws from its OWN torch.Generator, seeded from (seed, epoch).  Under data parallelism every
        rank must build the same batch list before dealing it out (qea.dist.deal_batches); the global CPU generator cannot give
        that, because the ranks consume it by data-dependent amounts in between (CPU jitter of k_r strips, skipped replicas)."""
        self.batch_size, self.drop_last, self.generator = batch_size, drop_last, generator
        self.seed, self.epoch = int(seed), 0
        self.groups = {}
        for i, w in enumerate(widths):
            self.groups.setdefault(bucket_of(int(w), buckets), []).append(i)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _batches(self):
        out = []
        gen = self.generator
        if gen is None:
            gen = torch.Generator().manual_seed((self.seed * 1000003 + self.epoch) & 0x7FFFFFFF)
        for idx in self.groups.values():
            perm = torch.randperm(len(idx), generator=gen).tolist()
            for s in range(0, len(idx), self.batch_size):
                chunk = [idx[j] for j in perm[s:s + self.batch_size]]
                if len(chunk) == self.batch_size or not self.drop_last:
                    out.append(chunk)
        order = torch.randperm(len(out), generator=gen).tolist()
        return [out[i] for i in order]

    def __iter__(self):
        from qea import dist as qdist
        # under torch.distributed every rank draws the same global batch list (own generator, same seed and epoch) and takes every
        # world-th batch; the epoch advances per pass so that successive epochs see different batches
        batches = qdist.deal_batches(self._batches())
        self.epoch += 1
        return iter(batches)

    def __len__(self):
        from qea import dist as qdist
        n = sum(len(idx) // self.batch_size if self.drop_last else -(-len(idx) // self.batch_size) for idx in self.groups.values())
        return n // qdist.world()


def bucket_collate(batch):
    """[(img [1,32,w], label, name[, index])] of ONE bucket -> (images [B,1,32,Wb], labels, names[, indices])."""
    cols = list(zip(*batch))
    imgs = torch.stack([pad_to_bucket(im) for im in cols[0]])
    out = [imgs, list(cols[1]), list(cols[2])]
    if len(cols) > 3:
        out.append(torch.tensor(cols[3]))
    return out
