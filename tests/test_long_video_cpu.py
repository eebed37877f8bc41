"""CPU checks of the long-video sampling host logic: sampling-scheme index sequences against
the reference's (tests/golden/schemes.json, written by oracle/make_golden.py), the window plumbing of
sample_video, result-path naming and the dataset module surface."""
import contextlib
import io
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import GOLDEN

with open(os.path.join(GOLDEN, "schemes.json")) as f:
    SCHEME_CASES = json.load(f)


def run_scheme(name, T, n_obs, K, step, B=2):
    from improved_diffusion.sampling_schemes import sampling_schemes
    with contextlib.redirect_stdout(io.StringIO()):
        it = iter(sampling_schemes[name](video_length=T, num_obs=n_obs, max_frames=K, step_size=step))
        it.set_videos([None] * B)
        out = []
        for obs, lat in it:
            assert len(obs) == B and len(lat) == B and obs[0] is obs[1] and lat[0] is lat[1]
            out.append([[int(i) for i in obs[0]], [int(i) for i in lat[0]]])
    return out


@pytest.mark.parametrize("case", SCHEME_CASES, ids=lambda c: f"{c['scheme']}-T{c['video_length']}-obs{c['n_obs']}-K{c['max_frames']}")
def test_scheme_index_sequences_equal_reference(case):
    got = run_scheme(case["scheme"], case["video_length"], case["n_obs"], case["max_frames"], case["step_size"])
    assert got == case["windows"]


def test_scheme_window_counts_and_invariants():
    """97 / 99 windows for T=1000, K=20, step=10 with 36 / 0 observed frames; every window
    has <= K frames, conditions only on finished frames and every frame is generated exactly once."""
    for case in SCHEME_CASES:
        T, n_obs, K = case["video_length"], case["n_obs"], case["max_frames"]
        done = set(range(n_obs))
        for obs, lat in case["windows"]:
            assert len(obs) + len(lat) <= K and len(lat) >= 1
            assert set(obs) <= done and not (set(lat) & done)
            done |= set(lat)
        assert done == set(range(T))
        if (T, K, case["step_size"]) == (1000, 20, 10) and n_obs in (0, 36):
            assert len(case["windows"]) == (97 if n_obs else 99)


def test_scheme_registry_and_typenames():
    from improved_diffusion.sampling_schemes import sampling_schemes
    assert list(sampling_schemes) == ["autoreg", "long-range", "hierarchy-2", "hierarchy-3", "hierarchy-4", "hierarchy-5",
                                      "adaptive-autoreg", "adaptive-hierarchy-2", "adaptive-hierarchy-3"]
    with contextlib.redirect_stdout(io.StringIO()) as out:
        s = sampling_schemes["hierarchy-3"](video_length=50, num_obs=2, max_frames=8, step_size=4)
    assert s.typename == "Hierarchy-3" and "Hierarchy-3" in out.getvalue()


def test_adaptive_scheme_with_injected_embedding():
    """Greedy farthest-point choice of conditioning frames (reference sampling_schemes.py:154-182) with a
    plain pixel embedding standing in for LPIPS (the lpips package / weights are not available offline)."""
    from improved_diffusion.sampling_schemes import sampling_schemes
    torch.manual_seed(0)
    B, T, K, step = 2, 24, 6, 2
    videos = torch.randn(B, T, 3, 4, 4)
    with contextlib.redirect_stdout(io.StringIO()):
        it = iter(sampling_schemes["adaptive-autoreg"](video_length=T, num_obs=4, max_frames=K, step_size=step))
        it.set_embed_fn(lambda vids, idx: torch.stack([vids[:, i].flatten(1) for i in idx], dim=1))
        done = set(range(4))
        nwin = 0
        while True:
            it.set_videos(videos)
            try:
                obs, lat = next(it)
            except StopIteration:
                break
            assert len(obs) == B and len(lat) == B and lat[0] == lat[1]
            for b in range(B):
                assert len(obs[b]) == K - step and len(set(obs[b])) == K - step and set(obs[b]) <= done
                assert obs[b][0] == max(done)            # always starts from the most recent frame
                # second pick = the frame farthest (in embedding) from the first
                cand = sorted(done, reverse=True)
                d = [float(((videos[b, cand[0]] - videos[b, c]) ** 2).sum()) for c in cand]
                assert obs[b][1] == cand[int(np.argmax(d))]
            done |= set(lat[0])
            nwin += 1
    assert done == set(range(T)) and nwin == 10


class _FrameIndexDiffusion:
    """Stub with the p_sample_loop signature: 'generates' frames whose value is their frame index, and
    records what it was asked for."""
    diffusion_space = "pixel"

    def __init__(self):
        self.calls = []

    def p_sample_loop(self, model, shape, clip_denoised=True, model_kwargs=None, latent_mask=None,
                      return_attn_weights=False, return_decoded=True):
        fi, x0, om, lm = (model_kwargs[k] for k in ("frame_indices", "x0", "obs_mask", "latent_mask"))
        assert tuple(shape) == tuple(x0.shape) and fi.dtype == torch.long and latent_mask is lm
        assert torch.equal(om + lm, torch.ones_like(om)) and om.shape == (x0.shape[0], x0.shape[1], 1, 1, 1)
        n_obs = int(om[0].sum())
        assert torch.equal(om[:, :n_obs], torch.ones_like(om[:, :n_obs]))          # observed frames first
        # conditioning frames must already hold their final value
        assert torch.equal(x0[:, :n_obs], fi[:, :n_obs].view(*fi[:, :n_obs].shape, 1, 1, 1).expand_as(x0[:, :n_obs]).float())
        self.calls.append((tuple(shape), return_decoded))
        return fi.view(*fi.shape, 1, 1, 1).expand_as(x0).float().clone(), {}


@pytest.mark.parametrize("scheme,T,n_obs,K,step", [("autoreg", 31, 3, 8, 3), ("long-range", 40, 5, 10, 4),
                                                   ("hierarchy-2", 60, 4, 10, 5), ("hierarchy-2", 50, 0, 8, 4)])
def test_sample_video_window_plumbing(scheme, T, n_obs, K, step):
    from improved_diffusion.video_sampler import sample_video, default_sampling_args
    B = 2
    batch = torch.arange(T, dtype=torch.float32).view(1, T, 1, 1, 1).expand(B, T, 2, 3, 3).contiguous()
    args = default_sampling_args(sampling_scheme=scheme, n_obs=n_obs, max_frames=K, max_latent_frames=step, device="cpu")
    diff = _FrameIndexDiffusion()
    samples, used = sample_video(args, None, diff, batch, verbose=False)
    assert torch.equal(samples, batch)                           # every frame filled with "its" content
    assert [[list(map(int, o[0])), list(map(int, l[0]))] for o, l in used] == run_scheme(scheme, T, n_obs, K, step)
    assert len(diff.calls) == len(used) and all(dec for _, dec in diff.calls)
    # just_get_indices copies ground-truth frames instead of sampling (reference video_sample.py:62-63)
    truth = torch.randn(B, T, 2, 3, 3)
    samples2, used2 = sample_video(args, None, None, truth, just_get_indices=True, verbose=False)
    assert torch.equal(samples2, truth) and len(used2) == len(used)


def test_latent_space_windows_are_not_decoded():
    from improved_diffusion.video_sampler import sample_video, default_sampling_args
    diff = _FrameIndexDiffusion()
    diff.diffusion_space = "latent"
    batch = torch.arange(12, dtype=torch.float32).view(1, 12, 1, 1, 1).expand(1, 12, 4, 2, 2).contiguous()
    args = default_sampling_args(sampling_scheme="autoreg", n_obs=2, max_frames=6, max_latent_frames=3, device="cpu")
    sample_video(args, None, diff, batch, verbose=False)
    assert diff.calls and not any(dec for _, dec in diff.calls)


def test_result_paths_and_lock(tmp_path):
    from improved_diffusion import test_util as tu
    a = NS(eval_dir=None, checkpoint_path="/scratch/vd/my-checkpoints/abcdefg/ema_0.9999_050000.pt", use_ddim=True,
           timestep_respacing="250")
    assert str(tu.get_model_results_path(a)) == "results/abcdefg/ema_0.9999_050000_ddim_respace250"
    ck = tmp_path / "checkpoints" / "run7" / "ema_latest.pt"
    ck.parent.mkdir(parents=True)
    torch.save({"step": 1234}, ck)
    a = NS(eval_dir=None, checkpoint_path=str(ck), use_ddim=False, timestep_respacing="")
    assert str(tu.get_model_results_path(a)) == "results/run7/ema_latest_1234"
    assert str(tu.get_model_results_path(NS(eval_dir="/x/y"))) == "/x/y"
    ident = tu.get_eval_run_identifier(NS(sampling_scheme="hierarchy-2", optimality="linspace-t", max_frames=20,
                                          max_latent_frames=10, T=1000, n_obs=36, dataset_partition="train"))
    assert ident == "trainset_hierarchy-2_optimal-linspace-t_20_10_1000_36"
    target = tmp_path / "model_config.json"
    with tu.Protect(target):
        assert (tmp_path / "model_config.json.lock").exists()
    img = torch.zeros(2, 3, 8, 8)
    tu.mark_as_observed(img)
    assert float(img[0, 0, 1, 3]) == 255 and float(img[0, 1, 1, 3]) == 0 and float(img[0, 0, 0, 0]) == 0


def test_video_datasets_surface(tmp_path, monkeypatch):
    from improved_diffusion import video_datasets as vd
    assert vd.default_T_dict["carla_no_traffic_2x_encoded"] == 1000 and vd.default_image_size_dict["carla_no_traffic_2x_encoded"] == 32
    assert set(vd.data_encoding_stats_dict) == {"carla_no_traffic_2x_encoded"}
    # CARLA layout: csv split files + one uint8 (T,H,W,C) tensor per video; rank sharding; test = first T frames
    monkeypatch.delenv("DATA_ROOT", raising=False)
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "datasets" / "carla" / "no-traffic"
    root.mkdir(parents=True)
    vids = {}
    for i in range(5):
        vids[i] = torch.randint(0, 256, (12, 8, 8, 3), dtype=torch.uint8)
        torch.save(vids[i], root / f"video_{i}.pt")
    (root / "video_train.csv").write_text("".join(f"some/dir/video_{i}.pt\n" for i in range(4)))
    (root / "video_test.csv").write_text("header\nsome/dir/video_4.pt\n")
    test = vd.get_test_dataset("carla_no_traffic", T=6)
    assert len(test) == 1 and test.is_test
    v, extra = test[0]
    assert extra == {} and v.shape == (6, 3, 8, 8)
    assert torch.allclose(v, -1 + 2 * (vids[4][:6].permute(0, 3, 1, 2).float() / 255))
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    train = vd.get_train_dataset("carla_no_traffic", T=12)
    assert train.fnames == ["video_1.pt", "video_3.pt"]
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    x2 = vd.get_train_dataset("carla_no_traffic_2x", T=12)          # same files, nearest-2x frames
    assert x2[0][0].shape == (12, 3, 16, 16)
    assert torch.equal(x2[1][0][:, :, ::2, ::2], vd.get_train_dataset("carla_no_traffic", T=12)[1][0])
    # DATA_ROOT: files are copied into the scratch copy on first use and read from there afterwards
    scratch = tmp_path / "scratch"
    monkeypatch.setenv("DATA_ROOT", str(scratch))
    cached = vd.get_test_dataset("carla_no_traffic", T=6)
    assert torch.equal(cached[0][0], v) and (scratch / "datasets/carla/no-traffic/video_4.pt").exists()
    assert (scratch / "datasets/carla/no-traffic/video_test.csv").exists()
    monkeypatch.delenv("DATA_ROOT")
    it = vd.load_data("carla_no_traffic", batch_size=2, T=5, num_workers=0)
    b, _ = next(it)
    assert b.shape == (2, 5, 3, 8, 8) and float(b.abs().max()) <= 1.0
    # npy datasets (mazes / minerl) and the synthetic stand-in
    mz = tmp_path / "datasets" / "gqn_mazes-torch" / "test"
    mz.mkdir(parents=True)
    arr = np.random.randint(0, 256, (7, 4, 4, 3), dtype=np.uint8)
    np.save(mz / "0.npy", arr)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    m = vd.get_test_dataset("mazes_cwvae", T=7)
    assert len(m) == 1 and torch.allclose(m[0][0], torch.from_numpy(arr).permute(0, 3, 1, 2).float() / 255 * 2 - 1)
    s = vd.get_test_dataset("synthetic_latent", T=40)
    assert s[3][0].shape == (40, 4, 16, 16) and torch.equal(s[3][0], s[3][0])
    with pytest.raises(Exception):
        vd.get_test_dataset("mazes")


def test_reference_schedules_chain_their_windows():
    """Window-level batching needs windows that neither read nor write each other's frames.  The reference's schedules
    do not offer that: in every non-adaptive scheme each window conditions on frames the PREVIOUS window generated
    (autoreg / long-range by construction; hierarchy-N infills left to right, each window observing the last frames of
    its left neighbour - sampling_schemes.py:69-230).  Checked on the reference's own window lists (schemes.json): with
    greedy grouping of consecutive same-length windows that touch none of the group's generated frames, cfg D
    (hierarchy-2, T=1000) is 97 groups of ONE window - a batched chain could not reproduce the sequential video.
    (hierarchy-5 is the only case with a few independent neighbours.)"""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "schemes.json")) as f:
        cases = json.load(f)

    def groups(windows):
        out, cur = [], []
        for j, (o, l) in enumerate(windows):
            ok = bool(cur) and len(o) + len(l) == len(windows[cur[0]][0]) + len(windows[cur[0]][1])
            for i in cur:
                ok = ok and not (set(o) & set(windows[i][1]) or set(l) & set(windows[i][1]) or set(l) & set(windows[i][0]))
            if ok:
                cur.append(j)
            else:
                if cur:
                    out.append(cur)
                cur = [j]
        return out + [cur]

    for c in cases:
        g = groups(c["windows"])
        if c["scheme"] in ("autoreg", "long-range", "hierarchy-2", "hierarchy-3", "hierarchy-4"):
            assert len(g) == len(c["windows"]), (c["scheme"], c["video_length"])
        # and directly: every window after the first observes at least one frame generated by an earlier window
        made = set()
        for j, (o, l) in enumerate(c["windows"]):
            if j > 0 and c["scheme"] in ("autoreg", "long-range", "hierarchy-2"):
                assert set(o) & made, (c["scheme"], j)
            made |= set(l)
