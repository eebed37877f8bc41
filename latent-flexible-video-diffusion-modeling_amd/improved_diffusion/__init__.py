"""MI355X-native implementation of the `improved_diffusion` hot path (video U-Net +
Gaussian diffusion), API-compatible with plai-group/latent-flexible-video-diffusion-modeling.
Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute
on the path runs in hand-written gfx950 kernels behind the C ABI of include/lfvdm_hip.h."""
