"""Image transforms of the training loop — drop-in for the reference's transform_helper.py.

AddGaussianNoice keeps the reference call (`noiser(image, noise_coef)` on one image,
transform_helper.py:26-45: sigma = randint(0..std)/100 if stochastic else std/100, + 1e-13;
out = clamp(image - coef*N(0, sigma), 0, 1)) and adds `batch()`, the MI355X form used by the
trainers: all `replicas x images` jittered by ONE Philox kernel launch with per-image sigma,
replicas fused into the batch dimension, no host round trip.  PadWhite is data preparation
(PIL), as in the reference (:6-23)."""
import torch


class PadWhite(object):
    def __init__(self, size):
        assert isinstance(size, (int, tuple))
        self.height, self.width = (size, size) if isinstance(size, int) else size

    def __call__(self, img):
        from PIL import ImageOps
        if img.size[0] > self.width or img.size[1] > self.height:
            img.thumbnail((self.width, self.height))
        dw, dh = self.width - img.size[0], self.height - img.size[1]
        left, top = dw // 2, dh // 2
        return ImageOps.expand(img, (left, top, dw - left, dh - top), fill=255)


class AddGaussianNoice(object):
    def __init__(self, std=5, mean=0, is_stochastic=False, return_noise=False):
        self.std, self.mean = std, mean
        self.is_stochastic, self.return_noise = is_stochastic, return_noise
        self._calls = 0

    def _sigma(self):
        s = torch.randint(low=0, high=self.std + 1, size=(1,)).item() / 100.0 if self.is_stochastic else self.std / 100.0
        return s + 0.0000000000001

    def __call__(self, image, noise_coef=1):
        noise = torch.normal(float(self.mean), self._sigma(), image.shape).to(image.device)
        out = (image - noise_coef * noise).clamp_(0, 1)
        return (out, noise) if self.return_noise else out

    def batch(self, images, replicas=1, noise_coef=1.0, seed=None):
        """images [K,1,H,W] on the GPU -> ([replicas*K,1,H,W] jittered, noise or None); sigma is drawn per
        image and replica on the host exactly as __call__ does, the normal deviates on the device."""
        from qea import ops
        K = images.shape[0]
        hw = images[0].numel()
        sig = torch.tensor([self._sigma() for _ in range(replicas * K)], dtype=torch.float32).to(images.device, non_blocking=True)
        out = torch.empty((replicas * K,) + tuple(images.shape[1:]), device=images.device)
        noise = torch.empty_like(out) if self.return_noise else None
        if seed is None:
            seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
        if self.mean != 0:
            raise NotImplementedError("AddGaussianNoice.batch: mean != 0 is not on the reference path")
        ops.jitter(images.contiguous(), sig, out, noise, K, replicas, hw, float(noise_coef), seed, self._calls)
        self._calls += 1
        return out, noise
