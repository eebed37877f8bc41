"""Video datasets + the infinite training loader (API of the reference's improved_diffusion/video_datasets.py).

One video per file under ``<DATA_ROOT>/<dataset dir>``; items are ``(video (T, C, H, W) float32 in [-1, 1] (or
pre-encoded latents), {})``.  Differences from the reference, none visible to the scripts:
  * the data shard is this process's ``torch.distributed`` / torchrun rank (the reference asks MPI,
    video_datasets.py:45-46) - one process per GPU;
  * uint8 frames are converted with torch ops instead of torchvision's ToTensor (not installed here);
  * ``synthetic`` datasets (``synthetic_latent``, ``synthetic_pixel``) generate seeded random videos so that
    training / sampling can be exercised without any files.
"""
import os
import shutil
from dataclasses import dataclass
from pathlib import Path
from typing import Callable

import numpy as np
import torch as th
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset

from .test_util import Protect

video_data_paths_dict = {
    "minerl": "datasets/minerl_navigate-torch",
    "mazes_cwvae": "datasets/gqn_mazes-torch",
    "carla_no_traffic": "datasets/carla/no-traffic",
    "carla_no_traffic_2x": "datasets/carla/no-traffic",
    "carla_no_traffic_2x_encoded": "datasets/carla/no-traffic-encoded",
}

default_T_dict = {
    "minerl": 500,
    "mazes_cwvae": 300,
    "carla_no_traffic": 1000,
    "carla_no_traffic_2x": 1000,
    "carla_no_traffic_2x_encoded": 1000,
    "synthetic_latent": 40,
    "synthetic_pixel": 40,
}

default_image_size_dict = {
    "minerl": 64,
    "mazes_cwvae": 64,
    "carla_no_traffic": 128,
    "carla_no_traffic_2x": 256,
    "carla_no_traffic_2x_encoded": 32,
    "synthetic_latent": 16,
    "synthetic_pixel": 128,
}

data_encoding_stats_dict = {
    "carla_no_traffic_2x_encoded": "datasets/carla/no-traffic-encoded/encoded_train_norm_stats.pt",
}


def _shard():
    """(rank, world) of this process: torch.distributed if initialised, else the torchrun / MPI launcher env."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    for r, w in (("RANK", "WORLD_SIZE"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"), ("PMI_RANK", "PMI_SIZE")):
        if r in os.environ and w in os.environ:
            return int(os.environ[r]), int(os.environ[w])
    return 0, 1


# ------------------------------------------------------------------------------------------------ storage formats
def _frames_uint8_to_signed_unit(frames):
    """(T, H, W, C) uint8 frames (tensor or array) -> (T, C, H, W) float32 in [-1, 1]."""
    v = frames if isinstance(frames, th.Tensor) else th.from_numpy(np.ascontiguousarray(frames))
    return v.permute(0, 3, 1, 2).to(th.float32).div_(255.0).mul_(2.0).sub_(1.0)


def _numbered_npy(folder, split):
    """<folder>/<split>/<idx>.npy, idx = 0 .. n-1 (MineRL, GQN mazes)."""
    n = sum(1 for f in (folder / split).iterdir())
    return [f"{split}/{i}.npy" for i in range(n)]


def _csv_listed(prefix=""):
    """<folder>/video_<split>.csv names one .pt file per line (CARLA); ``prefix`` selects the pre-encoded copies."""
    def lister(folder, split):
        with open(folder / f"video_{split}.csv") as f:
            return [prefix + line.strip().rsplit("/", 1)[-1] for line in f if ".pt" in line]
    return lister


@dataclass(frozen=True)
class _Format:
    """How one dataset family is stored: where, which files make a split, how a file becomes a (T, C, H, W) video."""
    folder: str
    files: Callable            # (local folder, "train" | "test") -> file names relative to the folder
    read: Callable             # path -> raw video
    to_video: Callable         # raw -> float (T, C, H, W) in [-1, 1] (or latents)
    shardable: bool = True
    index_file: str = ""       # a file the lister reads (fetched into the local copy first)


_FORMATS = {
    "minerl": _Format(video_data_paths_dict["minerl"], _numbered_npy, np.load, _frames_uint8_to_signed_unit, shardable=False),
    "mazes_cwvae": _Format(video_data_paths_dict["mazes_cwvae"], _numbered_npy, np.load, _frames_uint8_to_signed_unit,
                           shardable=False),
    "carla_no_traffic": _Format(video_data_paths_dict["carla_no_traffic"], _csv_listed(), th.load, _frames_uint8_to_signed_unit,
                                index_file="video_{split}.csv"),
    "carla_no_traffic_2x": _Format(
        video_data_paths_dict["carla_no_traffic_2x"], _csv_listed(), th.load,
        lambda raw: th.nn.functional.interpolate(_frames_uint8_to_signed_unit(raw), scale_factor=2),   # nearest 2x
        index_file="video_{split}.csv"),
    "carla_no_traffic_2x_encoded": _Format(video_data_paths_dict["carla_no_traffic_2x_encoded"], _csv_listed("encoded_"), th.load,
                                           lambda latents: latents, index_file="video_{split}.csv"),
}


class VideoFiles(Dataset):
    """One video per file.  The working copy of the dataset lives under ``$DATA_ROOT`` (node-local scratch) when that
    is set: a file that is not there yet is copied, under a lock, from the same relative path below the current
    directory the first time it is needed.  Without DATA_ROOT the files are read in place.

    Items are ``(video[T frames], {})``: a random window of T frames while training, the first T frames after
    ``set_test()``."""

    def __init__(self, fmt, split, T, shard=0, num_shards=1):
        if not fmt.shardable and (shard, num_shards) != (0, 1):
            raise AssertionError("Distributed training is not supported by this dataset yet.")
        self.fmt, self.split, self.T = fmt, split, T
        scratch = os.environ.get("DATA_ROOT", "")
        self.origin = Path(fmt.folder)
        self.local = Path(scratch) / fmt.folder if scratch else self.origin
        if fmt.index_file:
            self._fetch(fmt.index_file.format(split=split))
        listing_root = self.local if fmt.index_file else self.origin
        self.fnames = fmt.files(listing_root, split)[shard::num_shards]
        self.is_test = False
        print(f"{fmt.folder} [{split}]: {len(self.fnames)} videos in shard {shard}/{num_shards}")

    def _fetch(self, rel):
        dst = self.local / rel
        if not dst.exists():
            dst.parent.mkdir(parents=True, exist_ok=True)
            with Protect(dst):
                if not dst.exists():
                    shutil.copyfile(self.origin / rel, dst)
        return dst

    def set_test(self):
        self.is_test = True

    def __len__(self):
        return len(self.fnames)

    def __getitem__(self, idx):
        path = self._fetch(self.fnames[idx])
        try:
            video = self.fmt.to_video(self.fmt.read(path))
        except Exception:
            print(f"could not load {path}")
            raise
        return self._window(video), {}

    def _window(self, video):
        if self.T is None:
            return video
        n = len(video)
        if n < self.T:
            raise AssertionError(f"video has {n} frames, {self.T} requested")
        first = 0 if (self.is_test or n == self.T) else int(np.random.randint(n - self.T + 1))
        return video[first:first + self.T]


def _open(dataset_name, T, split, shard, num_shards):
    T = default_T_dict[dataset_name] if T is None else T
    if dataset_name.startswith("synthetic"):
        return SyntheticVideoDataset(dataset_name, T=T, seed=1234 + shard + (0 if split == "train" else 10_000))
    if dataset_name not in _FORMATS:
        raise Exception("no dataset", dataset_name)
    return VideoFiles(_FORMATS[dataset_name], split, T, shard, num_shards)


def load_data(dataset_name, batch_size, T=None, deterministic=False, num_workers=1, return_dataset=False):
    """Endless stream of training batches ``(video (B, T, C, H, W), {})`` from this rank's shard of the train split.
    (``return_dataset=True`` makes the generator yield the Dataset once; ``get_train_dataset`` returns it directly.)"""
    dataset = get_train_dataset(dataset_name, T)
    if return_dataset:
        yield dataset
        return
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=not deterministic, num_workers=num_workers, drop_last=True)
    while True:
        yield from loader


def get_train_dataset(dataset_name, T=None):
    return _open(dataset_name, T, "train", *_shard())


def get_test_dataset(dataset_name, T=None):
    """The whole (unsharded) test split with deterministic first-T windows."""
    if dataset_name == "mazes":
        raise Exception("Deprecated dataset.")
    dataset = _open(dataset_name, T, "test", 0, 1)
    dataset.set_test()
    return dataset


class SyntheticVideoDataset(Dataset):
    """Seeded random videos of the shape of a real dataset (no files): ``synthetic_latent`` = 4x16x16 latents,
    ``synthetic_pixel`` = 3x128x128 frames.  Smooth in time so that conditioning carries information."""

    SHAPES = {"synthetic_latent": (4, 16, 16), "synthetic_pixel": (3, 128, 128)}

    def __init__(self, name="synthetic_latent", T=40, length=1024, seed=1234):
        self.name, self.T, self.length, self.seed = name, T, length, seed
        self.chw = self.SHAPES[name]
        self.is_test = False

    def set_test(self):
        self.is_test = True

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = th.Generator().manual_seed(self.seed * 1_000_003 + idx)
        base = th.randn(1, *self.chw, generator=g)
        drift = th.randn(self.T, *self.chw, generator=g).cumsum(0) * 0.1
        video = (base + drift).clamp(-1, 1) if self.name == "synthetic_pixel" else (base + drift) * 0.7
        return video, {}
