"""A tiny deterministic stand-in for the SVD frame autoencoder (diffusers AutoencoderKLTemporalDecoder interface).

TEST INFRASTRUCTURE.  The real VAE weights need a network fetch and are out of reach offline; the encode/decode
BOUNDARY around it (chunking, dtype/device moves, pre-encoded de-normalisation, reshapes - reference
gaussian_diffusion.py:914-947) does not depend on what the autoencoder computes, so the golden generator runs the
reference's ``encode`` / ``decode`` against this object and the tests run the build's against the same object.
"""
import torch


class _Dist:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std


class _Enc:
    def __init__(self, dist):
        self.latent_dist = dist


class _Dec:
    def __init__(self, sample):
        self.sample = sample


class FakeVAE:
    """encode: 8x8 average pooling of the 3 colour channels + their mean as a 4th latent channel, posterior std 0
    (so that the sampled latent equals the mean whatever the generator state); decode: nearest 8x upsampling of
    latent channels 0..2 plus 0.25 x channel 3.  ``calls`` records the chunk sizes seen."""

    def __init__(self):
        self.calls = []

    def encode(self, frames):
        self.calls.append(("encode", int(frames.shape[0])))
        pooled = torch.nn.functional.avg_pool2d(frames.float(), 8)
        mean = torch.cat([pooled, pooled.mean(dim=1, keepdim=True)], dim=1).to(frames.dtype)
        return _Enc(_Dist(mean, torch.zeros_like(mean)))

    def decode(self, latents, num_frames=1):
        assert num_frames == 1
        self.calls.append(("decode", int(latents.shape[0])))
        z = latents.float()
        rgb = z[:, :3] + 0.25 * z[:, 3:4]
        return _Dec(torch.nn.functional.interpolate(rgb, scale_factor=8, mode="nearest").to(latents.dtype))


class FakeImageProcessor:
    """preprocess: [0, 1] -> [-1, 1] (what the diffusers VaeImageProcessor does for tensors with do_normalize)."""

    @staticmethod
    def preprocess(frames):
        return frames * 2 - 1


def stats_dict(channels=4):
    """Synthetic per-channel normalisation statistics in the layout of the reference's encoded_train_norm_stats.pt."""
    c = torch.arange(channels, dtype=torch.float32)
    return {"mean": 0.1 * c - 0.15, "std": 0.5 + 0.25 * c}
