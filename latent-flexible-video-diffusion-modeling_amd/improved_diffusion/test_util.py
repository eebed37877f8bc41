"""Result-path naming, a lock-file guard and small visualisation helpers used by the sampling script
(API of the reference's improved_diffusion/test_util.py:10-110).  Host-only; PIL / imageio are imported
lazily because they are optional here."""
import os
from pathlib import Path

import numpy as np
import torch as th
from filelock import FileLock


class Protect(FileLock):
    """``with Protect(path):`` serialises writers of ``path`` through ``<path>.lock`` (reference :10-18)."""

    def __init__(self, path, timeout=2, **kwargs):
        path = Path(path)
        super().__init__(path.parent / f"{path.name}.lock", timeout=timeout, **kwargs)


def get_model_results_path(args):
    """``results/<dirs after the first path component containing 'checkpoint'>/<ckpt stem>[_<step>][_ddim][_respaceN]``
    unless ``args.eval_dir`` is given (reference :21-62).  ``*latest`` checkpoints get their training step
    appended (read from the file)."""
    if args.eval_dir is not None:
        return Path(args.eval_dir)
    ckpt = Path(args.checkpoint_path)
    name = ckpt.stem
    if name.endswith("latest"):
        name += f"_{th.load(args.checkpoint_path, map_location='cpu')['step']}"
    if args.use_ddim:
        name += "_ddim"
    if args.timestep_respacing != "":
        name += f"_respace{args.timestep_respacing}"
    anchor = next((i for i, part in enumerate(ckpt.parts) if "checkpoint" in part), None)
    assert anchor is not None
    sub = Path(*ckpt.parts[anchor + 1:])
    return Path("results") / sub.parent / name


def get_eval_run_identifier(args):
    """``[trainset_]<scheme>[_optimal-<kind>]_<K>_<step>_<T>_<n_obs>`` (reference :65-72)."""
    ident = args.sampling_scheme
    if getattr(args, "optimality", None) is not None:
        ident += f"_optimal-{args.optimality}"
    ident += f"_{args.max_frames}_{args.max_latent_frames}_{args.T}_{args.n_obs}"
    if getattr(args, "dataset_partition", None) == "train":
        ident = "trainset_" + ident
    return ident


# ----------------------------------------------------------------------------------------- visualisation
def mark_as_observed(images, color=(255, 0, 0)):
    """Draw a 1-pixel frame, one pixel in from the border, on (..., 3, H, W) images in place (reference :78-83)."""
    for ch, value in enumerate(color):
        plane = images[..., ch, :, :]
        plane[..., :, 1:2] = value
        plane[..., 1:2, :] = value
        plane[..., :, -2:-1] = value
        plane[..., -2:-1, :] = value


def tensor2pil(tensor, drange=(0, 1)):
    """(B x) 3 x H x W tensor with values in drange -> PIL image (list for a batch) (reference :86-99)."""
    from PIL import Image
    assert tensor.ndim in (3, 4)
    if tensor.ndim == 3:
        return tensor2pil(tensor.unsqueeze(0), drange=drange)[0]
    arr = tensor.cpu().numpy().transpose(0, 2, 3, 1)
    arr = ((arr - drange[0]) / (drange[1] - drange[0]) * 255).astype(np.uint8)
    return [Image.fromarray(a) for a in arr]


def tensor2gif(tensor, path, drange=(0, 1), random_str=""):
    import imageio
    frames = [np.asarray(img) for img in tensor2pil(tensor, drange=drange)]
    imageio.mimsave(path, frames)


def tensor2mp4(tensor, path, drange=(0, 1), random_str=""):
    gif = f"/tmp/tmp_{random_str}.gif"
    tensor2gif(tensor, gif, drange=drange, random_str=random_str)
    os.system(f'ffmpeg -y -hide_banner -loglevel error -i {gif} -r 10 -movflags faststart -pix_fmt yuv420p '
              f'-vf "scale=trunc(iw/2)*2:trunc(ih/2)*2" {path}')
