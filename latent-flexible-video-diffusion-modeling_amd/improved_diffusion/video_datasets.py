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
from pathlib import Path

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


def _data_root():
    root = os.environ.get("DATA_ROOT", "")
    return Path(root) if root else Path(".")


def _make_dataset(dataset_name, T, train, shard, num_shards, root):
    if dataset_name.startswith("synthetic"):
        return SyntheticVideoDataset(dataset_name, T=T, seed=1234 + shard + (0 if train else 10_000))
    path = root / video_data_paths_dict[dataset_name] if root is not None else Path(video_data_paths_dict[dataset_name])
    split = "train" if train else "test"
    if dataset_name == "minerl":
        return MineRLDataset(os.path.join(path, split), shard=shard, num_shards=num_shards, T=T)
    if dataset_name == "mazes_cwvae":
        return GQNMazesDataset(os.path.join(path, split), shard=shard, num_shards=num_shards, T=T)
    if dataset_name == "carla_no_traffic":
        return CarlaDataset(train=train, path=path, shard=shard, num_shards=num_shards, T=T)
    if dataset_name == "carla_no_traffic_2x":
        return Carla2xDataset(train=train, path=path, shard=shard, num_shards=num_shards, T=T)
    if dataset_name == "carla_no_traffic_2x_encoded":
        return Carla2xDataset(train=train, path=path, shard=shard, num_shards=num_shards, T=T, encoded=True)
    raise Exception("no dataset", dataset_name)


def load_data(dataset_name, batch_size, T=None, deterministic=False, num_workers=1, return_dataset=False):
    """Infinite generator of training batches ``(video (B, T, C, H, W), {})`` for this rank's shard
    (reference :41-69).  With ``return_dataset=True`` the generator yields the Dataset object once - use
    ``get_train_dataset`` for a plain return value."""
    T = default_T_dict[dataset_name] if T is None else T
    shard, num_shards = _shard()
    if dataset_name not in video_data_paths_dict and not dataset_name.startswith("synthetic"):
        raise Exception("no dataset", dataset_name)
    dataset = _make_dataset(dataset_name, T, True, shard, num_shards, None)
    if return_dataset:
        yield dataset
        return
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=not deterministic, num_workers=num_workers,
                        drop_last=True)
    while True:
        yield from loader


def get_train_dataset(dataset_name, T=None):
    T = default_T_dict[dataset_name] if T is None else T
    shard, num_shards = _shard()
    return _make_dataset(dataset_name, T, True, shard, num_shards, None)


def get_test_dataset(dataset_name, T=None):
    """Unsharded test split rooted at ``$DATA_ROOT`` with deterministic (first-T) subsequences (reference :80-101)."""
    if dataset_name == "mazes":
        raise Exception("Deprecated dataset.")
    T = default_T_dict[dataset_name] if T is None else T
    dataset = _make_dataset(dataset_name, T, False, 0, 1, _data_root())
    dataset.set_test()
    return dataset


class BaseDataset(Dataset):
    """One file per video under ``path``.  When ``DATA_ROOT`` is set, files are copied there on first access
    (node-local scratch) and the original location is the path relative to DATA_ROOT (reference :104-190).
    Subclasses provide ``getitem_path``, ``loaditem`` and ``postprocess_video``."""

    def __init__(self, path, T):
        super().__init__()
        self.T = T
        self.path = Path(path)
        self.is_test = False

    def __len__(self):
        return len(list(self.get_src_path(self.path).iterdir()))

    def __getitem__(self, idx):
        path = self.getitem_path(idx)
        self.cache_file(path)
        try:
            video = self.loaditem(path)
        except Exception:
            print(f"Failed on loading {path}")
            raise
        return self.get_video_subsequence(self.postprocess_video(video), self.T), {}

    def getitem_path(self, idx):
        raise NotImplementedError

    def loaditem(self, path):
        raise NotImplementedError

    def postprocess_video(self, video):
        raise NotImplementedError

    def cache_file(self, path):
        if not path.exists():
            path.parent.mkdir(parents=True, exist_ok=True)
            with Protect(path):
                shutil.copyfile(str(self.get_src_path(path)), str(path))

    @staticmethod
    def get_src_path(path):
        root = os.environ.get("DATA_ROOT", "")
        if not root:
            return path
        root = Path(root)
        assert root in path.parents, f"Expected dataset item path ({path}) to be located under the data root ({root})."
        return Path(*path.parts[len(root.parts):])

    def set_test(self):
        self.is_test = True
        print("setting test mode")

    def get_video_subsequence(self, video, T):
        if T is None:
            return video
        if T < len(video):
            start = 0 if self.is_test else np.random.randint(len(video) - T + 1)
            video = video[start:start + T]
        assert len(video) == T
        return video


def _uint8_frames_to_unit_range(video):
    """(T, H, W, C) uint8 array -> (T, C, H, W) float in [-1, 1]."""
    v = th.as_tensor(np.asarray(video))
    return v.permute(0, 3, 1, 2).float() / 255 * 2 - 1


class CarlaDataset(BaseDataset):
    """``video_{train,test}.csv`` lists the ``.pt`` files (uint8 T,H,W,C); sharded by rank (reference :193-212)."""

    def __init__(self, train, path, shard, num_shards, T):
        super().__init__(path=path, T=T)
        self.split_path = self.path / f"video_{'train' if train else 'test'}.csv"
        self.cache_file(self.split_path)
        with open(self.split_path) as f:
            names = [line.rstrip("\n").split("/")[-1] for line in f if ".pt" in line]
        self.fnames = names[shard::num_shards]
        print(f"Loading {len(self.fnames)} files (Carla dataset).")

    def loaditem(self, path):
        return th.load(path)

    def getitem_path(self, idx):
        return self.path / self.fnames[idx]

    def postprocess_video(self, video):
        return -1 + 2 * (video.permute(0, 3, 1, 2).float() / 255)

    def __len__(self):
        return len(self.fnames)


class Carla2xDataset(CarlaDataset):
    """CARLA upsampled 2x (nearest), or its pre-encoded latents (``encoded_<name>.pt``) (reference :215-228)."""

    def __init__(self, train, path, shard, num_shards, T, encoded=False):
        super().__init__(train, path, shard, num_shards, T)
        self.encoded = encoded
        if encoded:
            self.fnames = ["encoded_" + n for n in self.fnames]
        print(f"Loading {len(self.fnames)} files (Carla dataset).")

    def postprocess_video(self, video):
        if self.encoded:
            return video
        return th.nn.functional.interpolate(super().postprocess_video(video), scale_factor=2)


class _NpyVideoDataset(BaseDataset):
    def __init__(self, path, shard, num_shards, T):
        assert shard == 0 and num_shards == 1, "Distributed training is not supported by this dataset yet."
        super().__init__(path=path, T=T)

    def getitem_path(self, idx):
        return self.path / f"{idx}.npy"

    def loaditem(self, path):
        return np.load(path)

    def postprocess_video(self, video):
        return _uint8_frames_to_unit_range(video)


class GQNMazesDataset(_NpyVideoDataset):
    """``<idx>.npy`` uint8 (T, H, W, C) maze videos (reference :231-247)."""


class MineRLDataset(_NpyVideoDataset):
    """``<idx>.npy`` uint8 (T, H, W, C) MineRL videos (reference :250-264)."""


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
