"""Minibatch-subset pickers — drop-in for the reference's selection_utils.py: same factory keys
(:220-229), same `query(images, labels, num_samples, names) -> (images_sel, labels_sel, idx)` and
`update_cer(batch_cers, names)` protocol, same `.cers` / `.all_cers` attributes.

TopKCERSampler ranks on the GPU (qea_topk_desc_stable: descending CER, ties in ascending index
order) when the images live there; the reference's `torch.argsort(descending=True)` leaves the order
of tied CERs to the sort implementation, this one pins it to the stable order."""
import random

import numpy as np
import torch


def calc_entropy(probs, num_classes=95):
    p = probs + 0.000001
    return -(probs * torch.log(p)).sum(dim=1) / torch.log(torch.tensor(float(num_classes)))


def update_entropies(self, crnn_scores, names):
    probs = torch.exp(crnn_scores.detach()).cpu()
    ents = [calc_entropy(probs[:, i, :]).mean().item() for i in range(probs.shape[1])]
    self.sampler.update_entropies(ents, names)


def _known(names, table):
    return [table[n] for n in names if n in table]


def _desc_stable_topk(values, k, device):
    """indices (CPU LongTensor) of the k largest values, descending, stable."""
    n = len(values)
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    k = min(k, n)
    keys = torch.tensor(values, dtype=torch.float32)
    if device is not None and device.type == "cuda":
        from qea import ops
        out = torch.empty(k, dtype=torch.int64, device=device)
        ops.topk_desc_stable(keys.to(device), n, k, out)
        return out.cpu()
    # host path for CPU tensors: same total order (value desc, index asc)
    return torch.from_numpy(np.argsort(-keys.numpy(), kind="stable")[:k].astype(np.int64))


def _spread_pick(values, num_samples):
    """The reference's range sampling (:41-57, :122-135): draw `num_samples` points uniformly over
    [min, max] of the estimates and take, without replacement, the sample nearest to each."""
    est = torch.tensor(values)
    if est.shape[0] == 0:
        return torch.tensor([], dtype=torch.long)
    pts = (est.max() - est.min()) * torch.rand(num_samples) + est.min()
    left = est.clone()
    idx = torch.zeros(num_samples, dtype=torch.long)
    for i, p in enumerate(pts):
        j = torch.argmin(torch.abs(p - left))
        idx[i] = j
        left[j] = 100
    return idx


class DataSampler:
    def __init__(self, cers=None):
        self.cers = cers if cers is not None else dict()
        self.all_cers = dict()

    def query(self, images, labels, num_samples, names=None):
        raise NotImplementedError

    def update_cer(self, batch_cers, names):
        for name, cer in zip(names, batch_cers):
            if name not in self.cers:
                print(f"Sample not present - {name}")
            self.cers[name] = cer
            self.all_cers.setdefault(name, []).append(cer)

    @staticmethod
    def _take(images, labels, idx):
        return images[idx.to(images.device)], [labels[i] for i in idx.tolist()], idx


class RandomSampler(DataSampler):
    def query(self, images, labels, num_samples, names=None):
        return self._take(images, labels, torch.randperm(images.shape[0])[:num_samples])


class CerRangeSampler(DataSampler):
    def __init__(self, cers, discount_factor=1):
        super().__init__(cers)
        self.discount_factor = discount_factor

    def query(self, images, labels, num_samples, names):
        return self._take(images, labels, _spread_pick(_known(names, self.cers), num_samples))


class TopKCERSampler(DataSampler):
    def __init__(self, cers, discount_factor=1):
        super().__init__(cers)
        self.discount_factor = discount_factor

    def query(self, images, labels, num_samples, names):
        # names missing from the dict are skipped BEFORE ranking (reference :146-148), so the returned
        # indices address the compacted list exactly as the reference's do
        idx = _desc_stable_topk(_known(names, self.cers), num_samples, images.device if torch.is_tensor(images) else None)
        return self._take(images, labels, idx)

    content_free = True     # the pick depends on names / CERs only: a trainer may pick first and clean only the picked images

    def query_global(self, images, labels, num_samples_global, names, with_counts=False):
        """Data-parallel form: the reference ranks the WHOLE minibatch (train_nn_area.py:220-225); here the minibatch is sharded
        over the ranks, so the shards' CERs are all-gathered (qea.dist.global_topk: stable descending order, rank-major index
        as the tie-break = the single-process order of the concatenated minibatch) and each rank keeps the winners that live
        in its shard — possibly none.  Returns (images_sel, labels_sel, idx, k_global[, winners per rank])."""
        from qea import dist as qdist
        res = qdist.global_topk(_known(names, self.cers), num_samples_global, with_counts=with_counts)
        imgs, labs, idx = self._take(images, labels, res[0])
        return (imgs, labs, idx) + tuple(res[1:])


class UniformEntropySampler(DataSampler):
    def __init__(self, entropies, cers):
        super().__init__(cers)
        self.entropies = entropies

    def query(self, images, labels, num_samples, names):
        return self._take(images, labels, _spread_pick(_known(names, self.entropies), num_samples))

    def update_entropies(self, ents, names):
        for n, e in zip(names, ents):
            if n not in self.entropies:
                print(f"Sample not present - {n}")
            self.entropies[n] = e


class _GlobalSampler(DataSampler):
    def __init__(self, cers, num_samples):
        super().__init__(cers)
        self.num_samples = num_samples
        self.selected_samplenames = dict()

    def query(self, images, labels, num_samples=-1, names=None):
        idx = torch.tensor([i for i, n in enumerate(names) if n in self.selected_samplenames], dtype=torch.long)
        return self._take(images, labels, idx)


class UniformSamplerGlobal(_GlobalSampler):
    def select_samples(self):
        self.selected_samplenames.clear()
        keys = list(self.cers.keys())
        order = np.argsort(np.array(list(self.cers.values())))
        for split in np.array_split(order, self.num_samples):
            self.selected_samplenames[keys[np.random.choice(split)]] = True


class RandomSamplerGlobal(_GlobalSampler):
    def select_samples(self):
        self.selected_samplenames.clear()
        for name in random.sample(list(self.cers.keys()), self.num_samples):
            self.selected_samplenames[name] = True


_METHODS = {
    "random": RandomSampler,
    "topKCER": TopKCERSampler,
    "uniformCERglobal": UniformSamplerGlobal,
    "randomglobal": RandomSamplerGlobal,
    "rangeCER": CerRangeSampler,
    "uniformEntropy": UniformEntropySampler,
}


def datasampler_factory(sampling_method):
    return _METHODS[sampling_method]
