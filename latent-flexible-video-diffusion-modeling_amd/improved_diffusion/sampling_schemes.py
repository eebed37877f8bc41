"""Frame-index schedules for generating long videos window by window.

Host-side logic only (no tensors on the hot path): every scheme is an iterator yielding
``(obs_frame_indices, latent_frame_indices)`` for the next window of at most ``max_frames`` frames, exactly as
the reference does (improved_diffusion/sampling_schemes.py:34-398; consumed by scripts/video_sample.py:39-84).
The index sequences are pinned against the reference by tests/golden/schemes.json.

Schemes:  autoreg · long-range · hierarchy-{2..5} · adaptive-autoreg · adaptive-hierarchy-{2,3}.
The adaptive schemes pick the conditioning frames by greedy farthest-point selection in an LPIPS embedding
(reference :154-182); the LPIPS network needs the `lpips` package and its weights, so they accept any
``embed_fn(videos, indices) -> (B, len(indices), D)`` and only fall back to LPIPS when none is given.
"""
import numpy as np
import torch as th


class SamplingSchemeBase:
    """Iterator protocol + bookkeeping shared by all schemes (reference :34-121).

    video_length: frames in the full video; num_obs: frames observed at the start; max_frames: window size K;
    step_size: latent frames generated per window; optimal_schedule_path: optional ``.pt`` dict
    {step -> observed indices} overriding the scheme's own choice of conditioning frames.
    """

    def __init__(self, video_length: int, num_obs: int, max_frames: int, step_size: int, optimal_schedule_path=None):
        msg = f'Inferring using the sampling scheme "{self.typename}"'
        msg += "." if optimal_schedule_path is None else f", and the optimal schedule stored at {optimal_schedule_path}."
        print(msg)
        self._video_length = video_length
        self._max_frames = max_frames
        self._num_obs = num_obs
        self._step_size = step_size
        self._done_frames = set(range(num_obs))
        self._obs_frames = list(range(num_obs))
        self.optimal_schedule = th.load(optimal_schedule_path) if optimal_schedule_path is not None else None
        self._current_step = 0
        self.B = None

    # -- hooks ---------------------------------------------------------------------------------------------
    def next_indices(self):
        raise NotImplementedError

    def get_unconditional_indices(self):
        return list(range(self._max_frames))

    def set_videos(self, videos):
        """Non-adaptive schemes only need the batch size (every video gets the same indices)."""
        self.B = len(videos)

    @property
    def typename(self):
        return type(self).__name__

    # -- iterator ------------------------------------------------------------------------------------------
    def is_done(self):
        return len(self._done_frames) >= self._video_length

    def __iter__(self):
        self.step = 0
        return self

    def _check_and_commit(self, obs, lat):
        assert isinstance(obs, list) and isinstance(lat, list)
        for i in np.array(obs, dtype=np.int64).flatten():
            assert int(i) in self._done_frames, (
                f"Attempting to condition on frame {i} while it is not generated yet.\n"
                f"Generated frames: {self._done_frames}\nObserving: {obs}\nGenerating: {lat}")
        assert np.all(np.array(lat) < self._video_length)
        self._done_frames.update(lat)
        self._current_step += 1

    def __next__(self):
        if self.is_done():
            raise StopIteration
        first_unconditional = self._num_obs == 0 and self._current_step == 0
        if first_unconditional:
            # nothing observed: the first window is all latent; afterwards behave like a conditional model
            obs, lat = [], self.get_unconditional_indices()
        else:
            obs, lat = self.next_indices()
            if self.optimal_schedule is not None:
                if self._current_step in self.optimal_schedule:
                    obs = self.optimal_schedule[self._current_step]
                else:
                    print(f"WARNING: optimal observations for prediction step #{self._current_step} was not found "
                          "in the saved optimal schedule.")
                    obs = []
        self._check_and_commit(obs, lat)
        if first_unconditional:
            self._obs_frames = lat
        if self.B is not None:
            obs, lat = [obs] * self.B, [lat] * self.B
        return obs, lat


class Autoregressive(SamplingSchemeBase):
    """Condition on the most recent K - step frames, generate the next `step` (reference :124-135)."""

    def next_indices(self):
        if not self._done_frames:
            return [], list(range(self._max_frames))
        obs = sorted(self._done_frames)[-(self._max_frames - self._step_size):]
        start = obs[-1] + 1
        return obs, list(range(start, min(start + self._step_size, self._video_length)))


class LongRangeAutoregressive(SamplingSchemeBase):
    """Half of the conditioning budget on the latest frames, the rest on the latest originally observed
    frames (reference :138-152)."""

    def next_indices(self):
        budget = self._max_frames - self._step_size
        chosen = set(sorted(self._done_frames)[-(budget // 2):])
        for i in sorted(self._obs_frames, reverse=True):
            chosen.add(i)
            if len(chosen) == budget:
                break
        start = max(self._done_frames) + 1
        return sorted(chosen), list(range(start, min(start + self._step_size, self._video_length)))


class HierarchyNLevel(SamplingSchemeBase):
    """Coarse-to-fine: level 1 spreads `step` latents over the whole remaining video, each later level fills
    in a grid `sample_every` apart, conditioning on generated frames between / after / before the latents
    (reference :155-229).  ``N`` (number of levels) is set by the subclass factory."""

    @property
    def N(self):
        raise NotImplementedError

    @property
    def typename(self):
        return f"{super().typename}-{self.N}"

    def _start_level_one(self, last):
        self.current_level = 1
        self.last_sampled_idx = last

    def _spread(self):
        return [int(i) for i in np.linspace(0, self._video_length - 1, self._max_frames)]

    def get_unconditional_indices(self):
        self._start_level_one(self._video_length - 1)
        return self._spread()

    @property
    def sample_every(self):
        level1 = (self._video_length - len(self._obs_frames)) / (self._step_size - 1)
        return int(level1 ** ((self.N - self.current_level) / (self.N - 1)))

    def _latent_grid(self):
        """Latent indices of the next window (updates the level when the current one is exhausted)."""
        n_sample = self._step_size
        idx = self.last_sampled_idx + self.sample_every
        if all(i in self._done_frames for i in range(idx, self._video_length)):
            self.current_level += 1
            self.last_sampled_idx = 0
            first_missing = min(i for i in range(self._video_length) if i not in self._done_frames)
            idx = first_missing - 1 + self.sample_every
        if self.current_level == 1:
            return [int(i) for i in np.linspace(max(self._obs_frames) + 1, self._video_length - 0.001, n_sample)]
        lat = []
        while len(lat) < n_sample and idx < self._video_length:
            if idx in self._done_frames:
                idx += 1
            else:
                lat.append(idx)
                idx += self.sample_every
        return lat

    def _retry_with_smaller_step(self):
        if self._step_size == 1:
            raise Exception("Cannot condition before and after even with step size of 1")
        self._step_size -= 1
        try:
            return self.next_indices()
        finally:
            self._step_size += 1

    def next_indices(self):
        if not self._done_frames:
            self._start_level_one(self._video_length - 1)
            return [], self._spread()
        if len(self._done_frames) == len(self._obs_frames):
            self._start_level_one(max(self._obs_frames))
        budget = self._max_frames - self._step_size
        lat = self._latent_grid()
        lo, hi = min(lat), max(lat)
        obs = [i for i in range(lo, hi) if i in self._done_frames]          # frames between the latents
        spare = budget - len(obs)
        if spare < 2:       # keep room to condition both before and after the latents
            return self._retry_with_smaller_step()
        after = [i for i in range(hi + 1, self._video_length) if i in self._done_frames]
        obs.extend(after[:spare // 2])
        n_before = budget - len(obs)
        if self.current_level == 1:
            obs.extend(list(np.linspace(0, max(self._obs_frames) + 0.999, n_before).astype(np.int32)))
        else:
            before = [i for i in range(lo - 1, -1, -1) if i in self._done_frames]
            obs.extend(before[:n_before])
        self.last_sampled_idx = hi
        return obs, lat


# ------------------------------------------------------------------------------------------------ adaptive
def _lpips_embed_fn():
    """Embedding whose squared distances are LPIPS distances (reference :6-31).  Needs `lpips` + weights."""
    try:
        import lpips
    except ImportError as e:   # pragma: no cover - package absent offline
        raise ImportError("adaptive sampling schemes need the `lpips` package (or pass embed_fn=...)") from e

    class Embedder(lpips.LPIPS):
        def forward(self, x):
            outs = self.net.forward(self.scaling_layer(x))
            parts = []
            for k in range(self.L):
                f = lpips.normalize_tensor(outs[k])
                f = (self.lins[k].model[-1].weight ** 0.5) * f
                B, C, H, W = f.shape
                parts.append(f.view(B, C * H * W, 1, 1) / (H * W) ** 0.5)
            return th.cat(parts, dim=1)

    nets = {}

    def embed(videos, indices):
        dev = videos.device
        if dev not in nets:
            nets[dev] = Embedder(net="alex", spatial=False).to(dev)
        return th.stack([nets[dev](videos[:, i]) for i in indices], dim=1)
    return embed


class AdaptiveSamplingSchemeBase(SamplingSchemeBase):
    """Per-video choice of conditioning frames: greedy farthest-point selection in an embedding space
    (reference :232-288)."""

    _embed_fn = None

    def set_embed_fn(self, fn):
        """fn(videos (B,T,C,H,W), indices) -> (B, len(indices), ...); replaces the LPIPS embedding."""
        self._embed_fn = fn

    def set_videos(self, videos):
        self.videos = videos

    def embed(self, indices):
        if self._embed_fn is None:
            self._embed_fn = _lpips_embed_fn()
        return self._embed_fn(self.videos, indices)

    def select_obs_indices(self, possible_next_indices, n, always_selected=(0,)):
        embs = self.embed(possible_next_indices)
        chosen_per_video = []
        for b in range(len(self.videos)):
            e = embs[b].reshape(len(possible_next_indices), -1)
            min_d = np.full(len(possible_next_indices), np.inf)
            picks = [always_selected[0]]
            for i in range(1, n):
                d_new = ((e[picks[-1]][None] - e) ** 2).sum(dim=1).cpu().numpy()
                min_d = np.minimum(min_d, d_new)
                picks.append(always_selected[i] if i < len(always_selected) else int(np.argmax(min_d)))
            chosen_per_video.append([possible_next_indices[j] for j in picks])
        return chosen_per_video

    def __next__(self):
        nvid = len(self.videos)
        if self._num_obs == 0 and self._current_step == 0:
            self.B = None
            obs, lat = SamplingSchemeBase.__next__(self)
            return [obs for _ in range(nvid)], [lat for _ in range(nvid)]
        if self.is_done():
            raise StopIteration
        obs, lat = self.next_indices()
        self._check_and_commit(obs, lat)
        return obs, [lat] * len(obs)


class AdaptiveAutoregressive(AdaptiveSamplingSchemeBase):
    """Next `step` frames, conditioned on adaptively selected earlier frames (reference :291-304)."""

    def next_indices(self):
        if not self._done_frames:
            return [[]] * len(self.videos), list(range(self._max_frames))
        start = max(self._done_frames) + 1
        lat = list(range(start, min(start + self._step_size, self._video_length)))
        candidates = sorted(self._done_frames, reverse=True)
        return self.select_obs_indices(candidates, self._max_frames - self._step_size), lat


class AdaptiveHierarchyNLevel(AdaptiveSamplingSchemeBase, HierarchyNLevel):
    """Hierarchy latents; conditioning = frames between the latents + the two closest before + the closest
    after, the remainder chosen adaptively (reference :307-372)."""

    def next_indices(self):
        if not self._done_frames:
            self._start_level_one(self._video_length - 1)
            return [], self._spread()
        if len(self._done_frames) == len(self._obs_frames):
            self._start_level_one(max(self._obs_frames))
        budget = self._max_frames - self._step_size
        lat = self._latent_grid()
        lo, hi = min(lat), max(lat)
        obs = [i for i in range(lo, hi) if i in self._done_frames]
        if budget - len(obs) < 2:
            return self._retry_with_smaller_step()
        i = lo
        for _ in range(2):                       # the two closest generated frames before the latents
            while i not in self._done_frames:
                i -= 1
            obs.append(i)
            i -= 1
        j = hi
        while j not in self._done_frames and j < self._video_length:
            j += 1
        if j < self._video_length:
            obs.append(j)
        candidates = list(self._done_frames)
        forced = [candidates.index(k) for k in obs]
        print("ALWAYS SELECTED", obs)
        obs = self.select_obs_indices(possible_next_indices=candidates, n=budget, always_selected=forced)
        self.last_sampled_idx = hi
        return obs, lat


def get_hierarchy_n_level(n):
    return type("Hierarchy", (HierarchyNLevel,), {"N": n})


def get_adaptive_hierarchy_n_level(n):
    return type("AdaptiveHierarchy", (AdaptiveHierarchyNLevel,), {"N": n})


sampling_schemes = {
    "autoreg": Autoregressive,
    "long-range": LongRangeAutoregressive,
    "hierarchy-2": get_hierarchy_n_level(2),
    "hierarchy-3": get_hierarchy_n_level(3),
    "hierarchy-4": get_hierarchy_n_level(4),
    "hierarchy-5": get_hierarchy_n_level(5),
    "adaptive-autoreg": AdaptiveAutoregressive,
    "adaptive-hierarchy-2": get_adaptive_hierarchy_n_level(2),
    "adaptive-hierarchy-3": get_adaptive_hierarchy_n_level(3),
}
