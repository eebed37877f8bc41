"""Timestep samplers behind ``create_named_schedule_sampler`` (reference resample.py).

``uniform`` is what the training CLI uses by default and what the hot path is measured with.  ``loss-second-moment``
(importance sampling by the RMS of recent losses per timestep) is kept for API completeness; it is written on
vectorised NumPy ring buffers and one ``all_gather_object`` per update rather than the reference's padded tensor
gathers and per-element Python loops.
"""
import numpy as np
import torch as th
import torch.distributed as dist


def create_named_schedule_sampler(name, diffusion):
    makers = {"uniform": UniformSampler, "loss-second-moment": LossSecondMomentResampler}
    if name not in makers:
        raise NotImplementedError(f"unknown schedule sampler: {name}")
    return makers[name](diffusion)


class ScheduleSampler:
    """Draws timesteps t ~ p and returns the importance weights 1 / (T p_t) that keep the loss unbiased."""

    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = np.asarray(self.weights(), dtype=np.float64)
        p = w / w.sum()
        t = np.random.choice(p.size, size=(batch_size,), p=p)
        scale = 1.0 / (p.size * p[t])
        return th.from_numpy(t).long().to(device), th.from_numpy(scale).float().to(device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones(diffusion.num_timesteps)

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    """Samplers that learn from the training losses.  Every rank must end up with the same history, so the local
    (t, loss) pairs are exchanged before the update."""

    def update_with_local_losses(self, local_ts, local_losses):
        pairs = (local_ts.detach().cpu().numpy().astype(np.int64), local_losses.detach().cpu().numpy().astype(np.float64))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            everyone = [None] * dist.get_world_size()
            dist.all_gather_object(everyone, pairs)
        else:
            everyone = [pairs]
        self.update_with_all_losses(np.concatenate([t for t, _ in everyone]), np.concatenate([l for _, l in everyone]))

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError


class LossSecondMomentResampler(LossAwareSampler):
    """p_t ∝ sqrt(mean of the last ``history_per_term`` squared losses at t), mixed with ``uniform_prob`` of uniform;
    uniform until every timestep has a full history."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        n = diffusion.num_timesteps
        self._ring = np.zeros((n, history_per_term))     # oldest entry of row t sits at column _head[t] once full
        self._head = np.zeros(n, dtype=np.int64)
        self._filled = np.zeros(n, dtype=np.int64)

    def _warmed_up(self):
        return bool((self._filled == self.history_per_term).all())

    def weights(self):
        n = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones(n)
        rms = np.sqrt((self._ring ** 2).mean(axis=1))
        return rms / rms.sum() * (1.0 - self.uniform_prob) + self.uniform_prob / n

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(np.asarray(ts).tolist(), np.asarray(losses).tolist()):   # order matters when a t repeats
            self._ring[t, self._head[t]] = loss
            self._head[t] = (self._head[t] + 1) % self.history_per_term
            self._filled[t] = min(self._filled[t] + 1, self.history_per_term)
