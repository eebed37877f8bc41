"""Timestep respacing (reference respace.py): ``space_timesteps``, ``SpacedDiffusion`` and the
model wrapper that remaps/rescales timesteps.  The remap table lives on the device once (the
reference rebuilds ``th.tensor(timestep_map)`` on every model call, respace.py:120)."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Timesteps to keep when dividing the chain into equally sized sections with the given step
    counts ("10,15,20"), or the fixed DDIM stride ("ddimN") (reference respace.py:7-60)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, kept = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """Diffusion over a subset of the base process' timesteps (reference respace.py:63-107)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_acp = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, acp in enumerate(base_acp):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def p_sample(self, model, *args, **kwargs):
        return super().p_sample(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        if not hasattr(self, "_map_cache"):
            self._map_cache = {}      # device-resident remap tables, shared by every wrapper of this diffusion
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, self._map_cache)

    def _scale_timesteps(self, t):
        return t  # done by the wrapped model (reference respace.py:105-107)

    def model_timestep_table(self, device):
        ts = th.tensor(self.timestep_map, dtype=th.float32, device=device)
        if self.rescale_timesteps:
            ts = ts * (1000.0 / self.original_num_steps)
        return ts


class _WrappedModel:
    """Maps spaced timestep indices to the base process and rescales them to 0..1000
    (reference respace.py:110-124)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, map_cache=None):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = map_cache if map_cache is not None else {}

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, timesteps, **kwargs):
        key = (str(timesteps.device), timesteps.dtype)
        mt = self._maps.get(key)
        if mt is None:
            mt = th.tensor(self.timestep_map, device=timesteps.device, dtype=timesteps.dtype)
            self._maps[key] = mt
        new_ts = mt[timesteps]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, timesteps=new_ts, **kwargs)
