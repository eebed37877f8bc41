"""Timestep samplers (reference resample.py).  ``UniformSampler`` is the hot-path default; the
loss-aware sampler is kept (with the NumPy>=1.24 ``np.int`` bug of reference :132 fixed)."""
from abc import ABC, abstractmethod

import numpy as np
import torch as th
import torch.distributed as dist


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler(ABC):
    """Importance sampler over diffusion timesteps (reference :23-58)."""

    @abstractmethod
    def weights(self):
        """Positive (unnormalised) weight per timestep, as a numpy array."""

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = th.from_numpy(idx).long().to(device)
        weights = th.from_numpy(1 / (len(p) * p[idx])).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """All-gather (ts, loss) pairs so every rank keeps the same history (reference :71-105)."""
        world = dist.get_world_size()
        sizes = [th.tensor([0], dtype=th.int32, device=local_ts.device) for _ in range(world)]
        dist.all_gather(sizes, th.tensor([len(local_ts)], dtype=th.int32, device=local_ts.device))
        sizes = [int(x.item()) for x in sizes]
        mx = max(sizes)
        ts_b = [th.zeros(mx).to(local_ts) for _ in sizes]
        ls_b = [th.zeros(mx).to(local_losses) for _ in sizes]
        pad_t = th.zeros(mx).to(local_ts)
        pad_t[:len(local_ts)] = local_ts
        pad_l = th.zeros(mx).to(local_losses)
        pad_l[:len(local_losses)] = local_losses
        dist.all_gather(ts_b, pad_t)
        dist.all_gather(ls_b, pad_l)
        ts = [int(x.item()) for y, n in zip(ts_b, sizes) for x in y[:n]]
        ls = [float(x.item()) for y, n in zip(ls_b, sizes) for x in y[:n]]
        self.update_with_all_losses(ts, ls)

    @abstractmethod
    def update_with_all_losses(self, ts, losses):
        ...


class LossSecondMomentResampler(LossAwareSampler):
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=np.int64)

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            if self._loss_counts[t] == self.history_per_term:
                self._loss_history[t, :-1] = self._loss_history[t, 1:]
                self._loss_history[t, -1] = loss
            else:
                self._loss_history[t, self._loss_counts[t]] = loss
                self._loss_counts[t] += 1

    def _warmed_up(self):
        return (self._loss_counts == self.history_per_term).all()
