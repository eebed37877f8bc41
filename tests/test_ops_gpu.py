"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle.  GPU only.

Tolerances are fp32: the kernels use exact-fp32 MFMA (a k-ordered fmaf chain) and differ from
ATen only by summation order and by v_exp/v_rcp (<= 1 ulp each) inside SiLU / softmax.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import recipe, unet_oracle as uo, diffusion_oracle as do

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from improved_diffusion import _native
    _native.lib()
    return _native


def rnd(tag, *shape, scale=1.0):
    return torch.from_numpy((scale * recipe.gaussianish(tag, int(np.prod(shape)))).reshape(shape).astype(np.float32))


def cl(x):  # NCHW -> channels-last rows [N, H, W, C] contiguous, on device
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def from_cl(y, N, H, W, Cc):
    return y.view(N, H, W, Cc).permute(0, 3, 1, 2).cpu()


def close(a, b, atol, rtol=1e-4):
    a, b = a.float().cpu(), b.float().cpu()
    err = float((a - b).abs().max())
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f"max|d|={err:.3e} (scale {float(b.abs().max()):.3e})"


def gn_act(nat, a, b, C0, C1, N, P, gamma, beta, film, film_div, film_ld, act):
    """lfvdm_gn_apply: act(GroupNorm32(cat(a, b)) (* (1 + scale) + shift)) materialised as one [N*P][C0+C1] tensor."""
    out = torch.empty(N * P, C0 + C1, device="cuda")
    nat.check(nat.lib().lfvdm_gn_apply(nat.ptr(a), nat.ptr(b) if b is not None else None, C0, C1, N, P, nat.ptr(gamma), nat.ptr(beta),
                                       film.data_ptr() if film is not None else None, film_div, film_ld, 1e-5, act, nat.ptr(out),
                                       None, None, None, nat.stream()), "lfvdm_gn_apply")
    return out


def packed(nat, w):
    o = torch.empty(w.shape[0], w.shape[2] * w.shape[3], w.shape[1], device="cuda")
    nat.pack_conv_weight(w.cuda().contiguous(), o)
    return o


@pytest.mark.parametrize("N,Cin,Cout,H", [(3, 64, 64, 16), (5, 128, 96, 8), (40, 256, 128, 2), (2, 32, 4, 16),
                                          (1, 64, 192, 5)])
def test_conv3x3_plain(nat, N, Cin, Cout, H):
    x = rnd("c3/x", N, Cin, H, H)
    w = rnd("c3/w", Cout, Cin, 3, 3, scale=0.05)
    b = rnd("c3/b", Cout)
    ref = F.conv2d(x, w, b, padding=1)
    out = torch.empty(N * H * H, Cout, device="cuda")
    nat.conv_igemm(src0=cl(x), C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, W=packed(nat, w), bias=b.cuda(), Cout=Cout,
                   out=out, ldo=Cout)
    close(from_cl(out, N, H, H, Cout), ref, 2e-5)


def test_conv3x3_nchw_out(nat):
    N, Cin, Cout, H = 4, 64, 4, 16
    x, w, b = rnd("c3n/x", N, Cin, H, H), rnd("c3n/w", Cout, Cin, 3, 3, scale=0.05), rnd("c3n/b", Cout)
    out = torch.empty(N, Cout, H, H, device="cuda")
    nat.conv_igemm(src0=cl(x), C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, W=packed(nat, w), bias=b.cuda(), Cout=Cout,
                   out=out, ldo=Cout, out_mode=nat.OUT_NCHW)
    close(out, F.conv2d(x, w, b, padding=1), 2e-5)


def test_conv_stride2_and_upsample(nat):
    N, Cc, H = 3, 64, 8
    x, w, b = rnd("cs/x", N, Cc, H, H), rnd("cs/w", Cc, Cc, 3, 3, scale=0.05), rnd("cs/b", Cc)
    wp = packed(nat, w)
    out = torch.empty(N * (H // 2) ** 2, Cc, device="cuda")
    nat.conv_igemm(src0=cl(x), C0=Cc, N=N, Hs=H, Ws=H, stride=2, Ho=H // 2, Wo=H // 2, W=wp, bias=b.cuda(), Cout=Cc,
                   out=out, ldo=Cc)
    close(from_cl(out, N, H // 2, H // 2, Cc), F.conv2d(x, w, b, stride=2, padding=1), 2e-5)
    out = torch.empty(N * (2 * H) ** 2, Cc, device="cuda")
    nat.conv_igemm(src0=cl(x), C0=Cc, N=N, Hs=H, Ws=H, up=1, Ho=2 * H, Wo=2 * H, W=wp, bias=b.cuda(), Cout=Cc,
                   out=out, ldo=Cc)
    close(from_cl(out, N, 2 * H, 2 * H, Cc), F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1), 2e-5)


@pytest.mark.parametrize("N,C0,C1,Cout,H", [(4, 64, 64, 64, 16), (6, 128, 64, 128, 8), (40, 128, 128, 128, 2)])
def test_resblock_fused(nat, N, C0, C1, Cout, H):
    """gn_apply (virtual concat made real) + conv + gn_apply(FiLM) + conv(+1x1 skip segment over the two raw sources) vs the
    oracle res_block on a concat input (unet.py:194-207,460): the launch sequence of the plan's 16x16 ResBlocks."""
    Cin = C0 + C1
    xa, xb = rnd("rb/xa", N, C0, H, H), rnd("rb/xb", N, C1, H, H, scale=1.5)
    x = torch.cat([xa, xb], 1)
    T = 2
    emb = rnd("rb/emb", N // T, 256)
    emb_n = emb.repeat_interleave(T, 0)
    shapes = {"in_layers.0.weight": (Cin,), "in_layers.0.bias": (Cin,), "in_layers.2.weight": (Cout, Cin, 3, 3),
              "in_layers.2.bias": (Cout,), "emb_layers.1.weight": (2 * Cout, 256), "emb_layers.1.bias": (2 * Cout,),
              "out_layers.0.weight": (Cout,), "out_layers.0.bias": (Cout,), "out_layers.3.weight": (Cout, Cout, 3, 3),
              "out_layers.3.bias": (Cout,), "skip_connection.weight": (Cout, Cin, 1, 1), "skip_connection.bias": (Cout,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("rbt." + k, s)) for k, s in shapes.items()}
    ref = uo.res_block(sd, "p", x, emb_n)
    d = {k: v.cuda() for k, v in sd.items()}
    a, b = cl(xa), cl(xb)
    P = H * H
    film = F.linear(uo.silu(emb), sd["p.emb_layers.1.weight"], sd["p.emb_layers.1.bias"]).cuda().contiguous()
    act1 = gn_act(nat, a, b, C0, C1, N, P, d["p.in_layers.0.weight"], d["p.in_layers.0.bias"], None, 1, 0, nat.ACT_SILU)
    h1 = torch.empty(N * P, Cout, device="cuda")
    nat.conv_igemm(src0=act1, C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H,
                   W=packed(nat, sd["p.in_layers.2.weight"]), bias=d["p.in_layers.2.bias"], Cout=Cout, out=h1, ldo=Cout)
    ref_h1 = F.conv2d(uo.silu(uo.group_norm32(x, sd["p.in_layers.0.weight"], sd["p.in_layers.0.bias"])),
                      sd["p.in_layers.2.weight"], sd["p.in_layers.2.bias"], padding=1)
    close(from_cl(h1, N, H, H, Cout), ref_h1, 5e-5)
    act2 = gn_act(nat, h1, None, Cout, 0, N, P, d["p.out_layers.0.weight"], d["p.out_layers.0.bias"], film, T, 2 * Cout, nat.ACT_SILU)
    out = torch.empty(N * P, Cout, device="cuda")
    nat.conv_igemm(src0=act2, C0=Cout, N=N, Hs=H, Ws=H, Ho=H, Wo=H,
                   W=packed(nat, sd["p.out_layers.3.weight"]), bias=d["p.out_layers.3.bias"], Cout=Cout,
                   s2src0=a, s2src1=b, s2C0=C0, s2C1=C1, W2=d["p.skip_connection.weight"].view(Cout, Cin).contiguous(),
                   bias2=d["p.skip_connection.bias"], out=out, ldo=Cout)
    close(from_cl(out, N, H, H, Cout), ref, 1e-4)


def test_resblock_identity_skip(nat):
    N, Cc, H, T = 4, 64, 8, 2
    x = rnd("rbi/x", N, Cc, H, H)
    emb = rnd("rbi/emb", N // T, 128)
    shapes = {"in_layers.0.weight": (Cc,), "in_layers.0.bias": (Cc,), "in_layers.2.weight": (Cc, Cc, 3, 3),
              "in_layers.2.bias": (Cc,), "emb_layers.1.weight": (2 * Cc, 128), "emb_layers.1.bias": (2 * Cc,),
              "out_layers.0.weight": (Cc,), "out_layers.0.bias": (Cc,), "out_layers.3.weight": (Cc, Cc, 3, 3),
              "out_layers.3.bias": (Cc,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("rbi." + k, s)) for k, s in shapes.items()}
    ref = uo.res_block(sd, "p", x, emb.repeat_interleave(T, 0))
    d = {k: v.cuda() for k, v in sd.items()}
    a, P = cl(x), H * H
    film = F.linear(uo.silu(emb), sd["p.emb_layers.1.weight"], sd["p.emb_layers.1.bias"]).cuda().contiguous()
    act1 = gn_act(nat, a, None, Cc, 0, N, P, d["p.in_layers.0.weight"], d["p.in_layers.0.bias"], None, 1, 0, nat.ACT_SILU)
    h1 = torch.empty(N * P, Cc, device="cuda")
    nat.conv_igemm(src0=act1, C0=Cc, N=N, Hs=H, Ws=H, Ho=H, Wo=H,
                   W=packed(nat, sd["p.in_layers.2.weight"]), bias=d["p.in_layers.2.bias"], Cout=Cc, out=h1, ldo=Cc)
    act2 = gn_act(nat, h1, None, Cc, 0, N, P, d["p.out_layers.0.weight"], d["p.out_layers.0.bias"], film, T, 2 * Cc, nat.ACT_SILU)
    out = torch.empty(N * P, Cc, device="cuda")
    nat.conv_igemm(src0=act2, C0=Cc, N=N, Hs=H, Ws=H, Ho=H, Wo=H,
                   W=packed(nat, sd["p.out_layers.3.weight"]), bias=d["p.out_layers.3.bias"], Cout=Cc,
                   res=a, ldr=Cc, out=out, ldo=Cc)
    close(from_cl(out, N, H, H, Cc), ref, 1e-4)


@pytest.mark.parametrize("B,T,Cc,H,Cout", [(2, 3, 4, 16, 64), (1, 3, 3, 10, 128), (2, 3, 4, 2, 32), (1, 3, 4, 5, 48), (1, 3, 2, 6, 32)])
def test_conv_in(nat, B, T, Cc, H, Cout):
    """lfvdm_conv_in (compositing + indicator channel + first 3x3 conv, unet.py:441-450,310-316): image borders, maps smaller
    than a 64-pixel workgroup (pixels of several frames in one wave), ragged last workgroup, 2..8 channel quads per thread."""
    x, x0 = rnd("ci/x", B, T, Cc, H, H), rnd("ci/x0", B, T, Cc, H, H)
    obs = torch.tensor([[1., 0, 1], [0, 0, 1]])[:B].reshape(B, T, 1, 1, 1)
    w, b = rnd("ci/w", Cout, Cc + 1, 3, 3, scale=0.2), rnd("ci/b", Cout)
    comp = torch.cat([x * (1 - obs) + x0 * obs, torch.ones_like(x[:, :, :1]) * obs], 2).reshape(B * T, Cc + 1, H, H)
    ref = F.conv2d(comp, w, b, padding=1)
    out = torch.empty(B * T * H * H, Cout, device="cuda")
    nat.conv_in(x.cuda(), x0.cuda(), obs.reshape(-1).cuda().contiguous(), w.cuda(), b.cuda(), out, B * T, Cc, H, H, Cout)
    close(from_cl(out, B * T, H, H, Cout), ref, 1e-5)


def test_rowdot_time_embed(nat):
    """sinusoid -> Linear -> SiLU -> Linear, then a grouped SiLU->Linear (nn.py:105-123, unet.py:303-308)."""
    B, ch = 3, 64
    ted = 4 * ch
    t = torch.tensor([0.0, 37.5, 999.0])
    w0, b0 = rnd("rd/w0", ted, ch, scale=0.1), rnd("rd/b0", ted, scale=0.1)
    w2, b2 = rnd("rd/w2", ted, ted, scale=0.05), rnd("rd/b2", ted, scale=0.1)
    w3, b3 = rnd("rd/w3", 96, ted, scale=0.05), rnd("rd/b3", 96, scale=0.1)
    e0 = F.linear(uo.timestep_embedding(t, ch), w0, b0)
    emb = F.linear(uo.silu(e0), w2, b2)
    e3 = F.linear(uo.silu(emb), w3, b3)
    e4 = F.linear(emb, w3, b3)
    import math
    freqs = torch.exp(-math.log(10000.0) * torch.arange(ch // 2, dtype=torch.float32) / (ch // 2))
    tin = torch.cat([t, torch.zeros(8 - B), freqs]).cuda()  # timesteps | pad | freqs at offset 8
    dev = {k: v.cuda() for k, v in dict(w0=w0, b0=b0, w2=w2, b2=b2, w3=w3, b3=b3).items()}
    o0, o1 = torch.empty(B, ted, device="cuda"), torch.empty(B, ted, device="cuda")
    o3, o4 = torch.empty(B, 96, device="cuda"), torch.empty(B, 96, device="cuda")

    def job(W, b, inp, out, K, O, ldin, ldout, mode, row0):
        return nat.RowdotJob(W.data_ptr(), b.data_ptr(), inp.data_ptr(), out.data_ptr(), K, O, B, ldin, ldout, mode, row0, 0)

    j0 = nat.jobs_to_device([job(dev["w0"], dev["b0"], tin, o0, ch, ted, 8, ted, 2, 0)], "cuda")
    nat.rowdot(j0, 1, ted)
    close(o0, e0, 2e-5)
    j1 = nat.jobs_to_device([job(dev["w2"], dev["b2"], o0, o1, ted, ted, ted, ted, 1, 0)], "cuda")
    nat.rowdot(j1, 1, ted)
    close(o1, emb, 2e-5)
    j2 = nat.jobs_to_device([job(dev["w3"], dev["b3"], o1, o3, ted, 96, ted, 96, 1, 0),
                             job(dev["w3"], dev["b3"], o1, o4, ted, 96, ted, 96, 0, 96)], "cuda")
    nat.rowdot(j2, 2, 192)
    close(o3, e3, 2e-5)
    close(o4, e4, 2e-5)


@pytest.mark.parametrize("Cc,heads", [(64, 4), (128, 4), (32, 2)])
def test_rpe_nets(nat, Cc, heads):
    B, T, ted = 2, 5, 128
    fi = torch.tensor([[0, 1, 2, 3, 4], [3, 17, 18, 400, 999]])
    temb_b = rnd("rn/temb", B, ted)
    temb = temb_b.repeat_interleave(T, 0)
    jobs, refs, keep = [], [], []
    tile0 = 0
    for i in range(3):
        shapes = {"embed_distances.weight": (Cc, 3), "embed_distances.bias": (Cc,),
                  "embed_diffusion_time.weight": (Cc, ted), "embed_diffusion_time.bias": (Cc,),
                  "out.weight": (Cc, Cc), "out.bias": (Cc,)}
        sd = {"p." + k: torch.from_numpy(recipe.fill_param(f"rn{i}." + k, s)) for k, s in shapes.items()}
        rel = fi.unsqueeze(-1) - fi.unsqueeze(-2)
        refs.append(uo.rpe_net(sd, "p", temb, rel, heads).reshape(B, T, T, Cc))
        d = {k: v.cuda() for k, v in sd.items()}
        tproj = F.linear(temb_b, sd["p.embed_diffusion_time.weight"], sd["p.embed_diffusion_time.bias"]).cuda().contiguous()
        R = torch.empty(B, T, T, Cc, device="cuda")
        keep.append((d, tproj, R))
        jobs.append(nat.RpeJob(tproj.data_ptr(), d["p.embed_distances.weight"].data_ptr(), d["p.embed_distances.bias"].data_ptr(),
                               d["p.out.weight"].data_ptr(), d["p.out.bias"].data_ptr(), R.data_ptr(), Cc, tile0, Cc, 0, None))
        tile0 += (B * T * T + 31) // 32
    jd = nat.jobs_to_device(jobs, "cuda")
    nat.rpe_nets(jd, 3, tile0, fi.cuda(), B, T)
    for (d, tp, R), ref in zip(keep, refs):
        close(R, ref, 2e-5)


@pytest.mark.parametrize("N,P,Cc,heads", [(3, 256, 64, 4), (2, 64, 128, 4), (5, 4, 128, 4), (2, 100, 64, 2), (1, 256, 32, 2), (2, 64, 32, 4), (1, 70, 96, 4),
                                            (2, 256, 384, 4), (2, 64, 512, 4), (1, 50, 320, 4), (1, 33, 448, 4)])
def test_attn_spatial(nat, N, P, Cc, heads):
    qkv = rnd("as/qkv", N, P, 3 * Cc)
    Fh = Cc // heads
    q, k, v = (qkv.view(N, P, 3, heads, Fh)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    attn = torch.softmax((q * Fh ** -0.5) @ k.transpose(-1, -2), -1)
    ref = (attn @ v).permute(0, 2, 1, 3).reshape(N, P, Cc)
    o = torch.empty(N * P, Cc, device="cuda")
    a = torch.empty(N, heads, P, P, device="cuda")
    nat.attn_spatial(qkv.cuda(), o, a, N, P, Cc, heads)
    close(o.view(N, P, Cc), ref, 2e-5)
    close(a, attn, 1e-5)


@pytest.mark.parametrize("N,P,Cc,heads", [(3, 256, 64, 4), (2, 64, 128, 4), (5, 4, 128, 4), (2, 100, 64, 2), (2, 64, 32, 4), (1, 70, 96, 4),
                                            (1, 256, 384, 4), (2, 64, 512, 4), (1, 50, 320, 4), (1, 33, 448, 4), (1, 300, 64, 4)])
def test_attn_spatial_backward(nat, N, P, Cc, heads):
    """dqkv of the flash-style backward kernels vs torch autograd (fp64) of the same core (rpe.py:143-169)."""
    qkv = rnd("asb/qkv", N, P, 3 * Cc)
    d_o = rnd("asb/do", N, P, Cc)
    Fh = Cc // heads
    x = qkv.double().requires_grad_(True)
    q, k, v = (x.view(N, P, 3, heads, Fh)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    attn = torch.softmax((q * Fh ** -0.5) @ k.transpose(-1, -2), -1)
    ref_o = (attn @ v).permute(0, 2, 1, 3).reshape(N, P, Cc)
    (ref_o * d_o.double()).sum().backward()
    qc, doc = qkv.cuda().view(N * P, 3 * Cc), d_o.cuda().view(N * P, Cc)
    o = torch.empty(N * P, Cc, device="cuda")
    lse = torch.empty(N * heads, P, device="cuda")
    nat.attn_spatial(qc, o, None, N, P, Cc, heads, lse=lse)
    ref_lse = torch.logsumexp((q * Fh ** -0.5) @ k.transpose(-1, -2), -1).reshape(N * heads, P)
    close(lse, ref_lse.detach().float(), 2e-5)
    dqkv = torch.full((N * P, 3 * Cc), float("nan"), device="cuda")
    nat.attn_spatial_bwd(qc, o, doc, lse, torch.empty(N * heads, P, device="cuda"), dqkv, N, P, Cc, heads)
    g = x.grad.float().view(N * P, 3 * Cc)
    err = float((dqkv.cpu() - g).abs().max())
    assert err <= 2e-5 * (1.0 + float(g.abs().max())) + 3e-5, err


@pytest.mark.parametrize("B,T,P,Cc,heads", [(2, 20, 16, 64, 4), (2, 5, 4, 128, 4), (1, 14, 9, 64, 4), (2, 32, 3, 32, 2), (2, 4, 16, 32, 4),
                                                   (1, 20, 7, 384, 4), (1, 3, 5, 512, 4), (2, 24, 2, 64, 4), (1, 27, 3, 96, 4),
                                                   (1, 8, 37, 64, 2), (2, 1, 6, 32, 4)])
def test_temporal_attention_block(nat, B, T, P, Cc, heads):
    """gn_temporal + qkv GEMM + RPE nets + temporal core + proj GEMM vs oracle rpe_attention (rpe.py:133-174)."""
    ted = 128
    shapes = {"qkv.weight": (3 * Cc, Cc), "qkv.bias": (3 * Cc,), "proj_out.weight": (Cc, Cc), "proj_out.bias": (Cc,),
              "norm.weight": (Cc,), "norm.bias": (Cc,)}
    for r in ("rpe_q", "rpe_k", "rpe_v"):
        shapes.update({f"{r}.rpe_net.embed_distances.weight": (Cc, 3), f"{r}.rpe_net.embed_distances.bias": (Cc,),
                       f"{r}.rpe_net.embed_diffusion_time.weight": (Cc, ted), f"{r}.rpe_net.embed_diffusion_time.bias": (Cc,),
                       f"{r}.rpe_net.out.weight": (Cc, Cc), f"{r}.rpe_net.out.bias": (Cc,)})
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("ta." + k, s)) for k, s in shapes.items()}
    x = rnd("ta/x", B, P, Cc, T)  # oracle layout (B, D, C, T)
    temb_b = rnd("ta/temb", B, ted)
    fi = torch.stack([torch.arange(T), torch.sort(torch.from_numpy(np.argsort(recipe.uniform_pm1("ta/fi", 1000))[:T].copy()))[0]])[:B]
    mask = (torch.from_numpy(recipe.uniform_pm1("ta/mask", B * T)).view(B, T) > 0).float()
    ref, ref_attn = uo.rpe_attention(sd, "p", x, temb_b.repeat_interleave(T, 0), fi, mask, heads, True)
    d = {k: v.cuda() for k, v in sd.items()}
    # channels-last activation [B*T][P][C]
    xc = x.permute(0, 3, 1, 2).reshape(B * T, P, Cc).contiguous().cuda()
    xn = torch.empty_like(xc)
    nat.gn_temporal(xc, d["p.norm.weight"], d["p.norm.bias"], 1e-5, xn, B, T, P, Cc)
    M = B * T * P
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.qkv.weight"], bias=d["p.qkv.bias"],
                   Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    jobs, Rs, keep, tile0 = [], {}, [], 0
    for r in ("rpe_q", "rpe_k", "rpe_v"):
        pre = f"p.{r}.rpe_net."
        tproj = F.linear(temb_b, sd[pre + "embed_diffusion_time.weight"], sd[pre + "embed_diffusion_time.bias"]).cuda().contiguous()
        Rs[r] = torch.empty(B, T, T, Cc, device="cuda")
        keep.append(tproj)
        jobs.append(nat.RpeJob(tproj.data_ptr(), d[pre + "embed_distances.weight"].data_ptr(), d[pre + "embed_distances.bias"].data_ptr(),
                               d[pre + "out.weight"].data_ptr(), d[pre + "out.bias"].data_ptr(), Rs[r].data_ptr(), Cc, tile0, Cc, 0, None))
        tile0 += (B * T * T + 31) // 32
    nat.rpe_nets(nat.jobs_to_device(jobs, "cuda"), 3, tile0, fi.cuda(), B, T)
    o = torch.empty(M, Cc, device="cuda")
    attn = torch.empty(B * P, heads, T, T, device="cuda")
    nat.attn_temporal(qkv, Rs["rpe_q"], Rs["rpe_k"], Rs["rpe_v"], mask.cuda(), o, attn, B, T, P, Cc, heads)
    close(attn.view(B, P, heads, T, T), ref_attn, 2e-5)
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xn.view(M, Cc), ldr=Cc, out=y, ldo=Cc)
    got = y.view(B, T, P, Cc).permute(0, 2, 3, 1)  # -> (B, P, C, T)
    close(got, ref, 1e-4)


def _temporal_core_f64(qkv, Rq, Rk, Rv, mask, B, T, P, Cc, heads):
    """fp64 restatement of the temporal core (reference rpe.py:143-169): rows (b, t, p), channels [3][heads][F]."""
    Fh = Cc // heads
    scale = Fh ** -0.5
    x = qkv.view(B, T, P, 3, heads, Fh).permute(3, 0, 2, 4, 1, 5)      # 3, B, P, H, T, F
    q, k, v = x[0] * scale, x[1], x[2]
    logits = q @ k.transpose(-1, -2)
    logits = logits + torch.einsum("bdhtf,btshf->bdhts", q, Rk.view(B, T, T, heads, Fh))
    logits = logits + torch.einsum("bdhtf,btshf->bdhts", k * scale, Rq.view(B, T, T, heads, Fh)).transpose(-1, -2)
    m = mask.view(B, T)
    same = m[:, None, :] * m[:, :, None] + (1 - m[:, None, :]) * (1 - m[:, :, None])
    logits = logits.masked_fill((same == 0).view(B, 1, 1, T, T), float("-inf"))
    attn = torch.softmax(logits, dim=-1)
    out = attn @ v + torch.einsum("bdhts,btshf->bdhtf", attn, Rv.view(B, T, T, heads, Fh))
    return out.permute(0, 3, 1, 2, 4).reshape(B * T * P, Cc)


@pytest.mark.parametrize("B,T,P,Cc", [(2, 20, 256, 64), (2, 20, 64, 128), (2, 20, 4, 128), (1, 20, 16, 128), (2, 7, 6, 64), (1, 24, 3, 256),
                                      (3, 5, 1, 128), (1, 20, 256, 64), (8, 20, 64, 128)])
def test_temporal_groupnorm_with_the_qkv_projection_inside(nat, B, T, P, Cc):
    """lfvdm_gn_temporal_qkv (temporal GroupNorm + qkv Linear of the temporal attention in one launch, rpe.py:136 + :139):
    the normalised rows == lfvdm_gn_temporal's, qkv == lfvdm_gn_temporal + the 1x1 lfvdm_conv_igemm within fp32
    re-association, both == torch's group_norm + linear (fp64) on the (b, pixel) columns; bitwise reproducible."""
    L = nat.lib()
    M = B * T * P
    assert (L.lfvdm_gn_temporal_qkv_ok(B, T, P, Cc) == 0) == (Cc == 64 or 6.0 * M * Cc * Cc <= 0.5e9)     # (big projections: two launches)
    x = (rnd("tq/x", M, Cc) * 1.3 + 0.25).cuda()                     # rows (b, t, pixel); off-centre on purpose
    gam, bet = (rnd("tq/g", Cc) * 0.3 + 1.0).cuda(), (rnd("tq/b", Cc) * 0.2).cuda()
    W, bias = (rnd("tq/w", 3 * Cc, Cc) * (Cc ** -0.5)).cuda(), (rnd("tq/bias", 3 * Cc) * 0.1).cuda()
    xn = torch.empty(M, Cc, device="cuda")
    nat.gn_temporal(x, gam, bet, 1e-5, xn, B, T, P, Cc)
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=bias, Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    x0 = x.clone()
    xn1 = torch.full((M, Cc), float("nan"), device="cuda")
    qkv1 = torch.full((M, 3 * Cc), float("nan"), device="cuda")
    run = lambda: nat.check(L.lfvdm_gn_temporal_qkv(nat.ptr(x), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ptr(xn1), nat.ptr(W), nat.ptr(bias),
                                                    nat.ptr(qkv1), B, T, P, Cc, nat.stream()), "lfvdm_gn_temporal_qkv")
    run()
    assert torch.equal(x, x0), "the input is left alone"
    close(xn1, xn, 1e-5)
    close(qkv1, qkv, 2e-5)
    # torch (fp64): columns (b, pixel) as samples of a (C, T) GroupNorm
    xc = x.view(B, T, P, Cc).permute(0, 2, 3, 1).reshape(B * P, Cc, T).double().cpu()
    ref_n = F.group_norm(xc, 32, gam.double().cpu(), bet.double().cpu(), 1e-5)
    ref_n = ref_n.view(B, P, Cc, T).permute(0, 3, 1, 2).reshape(M, Cc)
    ref_q = ref_n @ W.double().cpu().t() + bias.double().cpu()
    close(xn1, ref_n, 1e-5)
    close(qkv1, ref_q, 2e-5)
    a, b2 = xn1.clone(), qkv1.clone()
    run()
    assert torch.equal(xn1, a) and torch.equal(qkv1, b2), "bitwise reproducible"
    # without the normalised rows (NULL) the projection is unchanged; aliasing them with the input is refused
    qkv2 = torch.empty_like(qkv1)
    nat.check(L.lfvdm_gn_temporal_qkv(nat.ptr(x), nat.ptr(gam), nat.ptr(bet), 1e-5, None, nat.ptr(W), nat.ptr(bias), nat.ptr(qkv2),
                                      B, T, P, Cc, nat.stream()), "lfvdm_gn_temporal_qkv")
    assert torch.equal(qkv2, qkv1)
    assert L.lfvdm_gn_temporal_qkv(nat.ptr(x), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ptr(x), nat.ptr(W), nat.ptr(bias), nat.ptr(qkv2),
                                   B, T, P, Cc, nat.stream()) != 0


@pytest.mark.parametrize("N,Cc,P", [(40, 64, 256), (3, 64, 256), (9, 128, 256), (20, 128, 256), (40, 128, 64), (5, 64, 64)])
def test_projection_with_the_next_groupnorm_inside(nat, N, Cc, P):
    """lfvdm_proj_gn (1x1 projection + bias + residual + GroupNorm32 of the sum in one launch, frames of 256 / 64 positions:
    rpe.py:171-172 then the next attention's rpe.py:136) == lfvdm_conv_igemm (1x1, res) + lfvdm_gn_apply within fp32
    re-association, == torch linear + group_norm in fp64; bitwise reproducible; aliased outputs are refused."""
    L = nat.lib()
    assert L.lfvdm_proj_gn_ok(N, P, Cc) == 0
    M = N * P
    o = (rnd("pg/o", M, Cc) * 0.8).cuda()
    res = (rnd("pg/res", M, Cc) * 1.2 + 0.3).cuda()
    W, bias = (rnd("pg/w", Cc, Cc) * (Cc ** -0.5)).cuda(), (rnd("pg/bias", Cc) * 0.1).cuda()
    gam, bet = (rnd("pg/g", Cc) * 0.3 + 1.0).cuda(), (rnd("pg/b", Cc) * 0.2).cuda()
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=bias, Cout=Cc, res=res, ldr=Cc, out=y, ldo=Cc)
    yn = torch.empty(M, Cc, device="cuda")
    nat.check(L.lfvdm_gn_apply(nat.ptr(y), None, Cc, 0, N, P, nat.ptr(gam), nat.ptr(bet), None, 1, 0, 1e-5, nat.ACT_NONE, nat.ptr(yn),
                               None, None, None, nat.stream()), "lfvdm_gn_apply")
    o0, r0 = o.clone(), res.clone()
    out = torch.full((M, Cc), float("nan"), device="cuda")
    run = lambda: nat.check(L.lfvdm_proj_gn(nat.ptr(o), nat.ptr(W), nat.ptr(bias), nat.ptr(res), nat.ptr(gam), nat.ptr(bet), 1e-5,
                                            nat.ACT_NONE, nat.ptr(out), None, N, P, Cc, nat.stream()), "lfvdm_proj_gn")
    run()
    assert torch.equal(o, o0) and torch.equal(res, r0), "the operands are left alone"
    close(out, yn, 2e-5)
    y64 = o.double().cpu() @ W.double().cpu().t() + bias.double().cpu() + res.double().cpu()
    ref = F.group_norm(y64.view(N, P, Cc).permute(0, 2, 1), 32, gam.double().cpu(), bet.double().cpu(), 1e-5).permute(0, 2, 1).reshape(M, Cc)
    close(out, ref, 2e-5)
    a = out.clone()
    run()
    assert torch.equal(out, a), "bitwise reproducible"
    # with SiLU and the raw sum (the U-Net head's GroupNorm + SiLU behind the last attention block, unet.py:418-422)
    out2 = torch.full((M, Cc), float("nan"), device="cuda")
    raw = torch.full((M, Cc), float("nan"), device="cuda")
    nat.check(L.lfvdm_proj_gn(nat.ptr(o), nat.ptr(W), nat.ptr(bias), nat.ptr(res), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ACT_SILU,
                              nat.ptr(out2), nat.ptr(raw), N, P, Cc, nat.stream()), "lfvdm_proj_gn")
    close(raw, y64, 2e-5)
    close(raw, y, 2e-5)
    close(out2, F.silu(ref), 2e-5)
    for bad in (o, res):
        assert L.lfvdm_proj_gn(nat.ptr(o), nat.ptr(W), nat.ptr(bias), nat.ptr(res), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ACT_NONE,
                               nat.ptr(bad), None, N, P, Cc, nat.stream()) != 0
        assert L.lfvdm_proj_gn(nat.ptr(o), nat.ptr(W), nat.ptr(bias), nat.ptr(res), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ACT_NONE,
                               nat.ptr(out), nat.ptr(bad), N, P, Cc, nat.stream()) != 0
    assert L.lfvdm_proj_gn(nat.ptr(o), nat.ptr(W), nat.ptr(bias), nat.ptr(res), nat.ptr(gam), nat.ptr(bet), 1e-5, nat.ACT_NONE,
                           nat.ptr(out), nat.ptr(out), N, P, Cc, nat.stream()) != 0
    assert L.lfvdm_proj_gn_ok(N, 16, Cc) != 0 and L.lfvdm_proj_gn_ok(N, 256, 96) != 0      # other maps: the GEMM's epilogue / two launches


def test_temporal_groupnorm_qkv_refuses_what_it_does_not_cover(nat):
    L = nat.lib()
    assert L.lfvdm_gn_temporal_qkv_ok(2, 20, 5, 64) != 0        # two pixels per workgroup at 64 channels: odd pixel count
    assert L.lfvdm_gn_temporal_qkv_ok(2, 25, 16, 128) != 0      # more than 24 frames
    assert L.lfvdm_gn_temporal_qkv_ok(2, 20, 16, 96) != 0       # channel counts other than 64 / 128 / 256
    assert L.lfvdm_gn_temporal_qkv_ok(2, 20, 16, 256) == 0
    assert L.lfvdm_gn_temporal_qkv_ok(2, 20, 256, 128) != 0     # covered, but a 1 GFLOP projection is faster as its own GEMM
    assert L.lfvdm_gn_temporal_qkv_ok(8, 20, 256, 64) == 0


@pytest.mark.parametrize("B,T,P,Cc,heads", [(2, 20, 16, 64, 4), (2, 5, 4, 128, 4), (1, 14, 9, 64, 4), (2, 32, 3, 32, 2), (2, 4, 16, 32, 4),
                                                   (1, 20, 7, 384, 4), (1, 3, 5, 512, 4), (2, 24, 2, 64, 4), (1, 27, 3, 96, 4),
                                                   (1, 8, 37, 64, 2), (2, 1, 6, 32, 4), (2, 20, 70, 128, 4)])
def test_attn_temporal_backward(nat, B, T, P, Cc, heads):
    """dqkv, dR_q, dR_k, dR_v of the HIP backward kernels vs fp64 autograd of the same core."""
    M = B * T * P
    qkv = rnd("tb/qkv", M, 3 * Cc)
    Rs = [0.3 * rnd(f"tb/R{i}", B, T, T, Cc) for i in range(3)]
    d_o = rnd("tb/do", M, Cc)
    mask = (torch.from_numpy(recipe.uniform_pm1("tb/mask", B * T)).view(B, T) > 0).float()
    leaves = [t.double().requires_grad_(True) for t in [qkv] + Rs]
    ref = _temporal_core_f64(leaves[0], leaves[1], leaves[2], leaves[3], mask.double(), B, T, P, Cc, heads)
    (ref * d_o.double()).sum().backward()
    g = [t.cuda().contiguous() for t in (qkv, d_o, *Rs, mask)]
    o = torch.empty(M, Cc, device="cuda")
    nat.attn_temporal(g[0], g[2], g[3], g[4], g[5], o, None, B, T, P, Cc, heads)
    close(o, ref.detach().float(), 5e-5)
    nanf = lambda *shape: torch.full(shape, float("nan"), device="cuda")
    dqkv, dRq, dRk, dRv = nanf(M, 3 * Cc), nanf(B, T, T, Cc), nanf(B, T, T, Cc), nanf(B, T, T, Cc)
    ws = torch.empty(2, B * P * heads * T * T, device="cuda")
    nat.attn_temporal_bwd(g[0], g[1], g[2], g[3], g[4], g[5], ws[0], ws[1], dqkv, dRq, dRk, dRv, B, T, P, Cc, heads)
    for name, got, want in (("dqkv", dqkv, leaves[0].grad), ("dRq", dRq, leaves[1].grad), ("dRk", dRk, leaves[2].grad),
                            ("dRv", dRv, leaves[3].grad)):
        want = want.float()
        err = float((got.cpu() - want).abs().max())
        assert err <= 3e-5 * (1.0 + float(want.abs().max())) + 3e-5, (name, err, float(want.abs().max()))


@pytest.mark.parametrize("B,T,P,Cc,heads", [(2, 20, 256, 64, 4), (2, 20, 64, 128, 4), (2, 20, 4, 128, 4), (1, 14, 256, 64, 4),
                                                   (1, 20, 33, 64, 4), (2, 17, 61, 32, 4), (3, 9, 130, 64, 2), (1, 32, 70, 128, 4),
                                                   (2, 6, 300, 32, 2), (1, 20, 256, 128, 4), (2, 20, 64, 256, 4), (1, 13, 21, 128, 2)])
def test_attn_temporal_second_generation_kernel(nat, B, T, P, Cc, heads):
    """attention_temporal2.hip (LDS-DMA staged, frame groups, XCD-aware map; head dims 8 / 16 / 32) at the network's
    real map sizes and at ragged ones - partial strips, partial frame groups, batch 3 (plain block map) - vs the fp64
    core; attention probabilities as well; and with the R tensors given as timestep tables (rsel)."""
    M = B * T * P
    qkv = rnd("t2/qkv", M, 3 * Cc)
    Rs = [0.3 * rnd(f"t2/R{i}", B, T, T, Cc) for i in range(3)]
    mask = (torch.from_numpy(recipe.uniform_pm1("t2/mask", B * T)).view(B, T) > 0).float()
    ref = _temporal_core_f64(qkv.double(), Rs[0].double(), Rs[1].double(), Rs[2].double(), mask.double(), B, T, P, Cc, heads).float()
    g = [t.cuda().contiguous() for t in (qkv, *Rs, mask)]
    o = torch.full((M, Cc), float("nan"), device="cuda")
    attn = torch.full((B * P, heads, T, T), float("nan"), device="cuda")
    nat.attn_temporal(g[0], g[1], g[2], g[3], g[4], o, attn, B, T, P, Cc, heads)
    close(o, ref, 5e-5)
    assert bool(torch.isfinite(attn).all()) and float((attn.sum(-1) - 1).abs().max()) < 1e-5
    o2 = torch.full((M, Cc), float("nan"), device="cuda")
    nat.attn_temporal(g[0], g[1], g[2], g[3], None, o2, None, B, T, P, Cc, heads)          # no mask
    ref2 = _temporal_core_f64(qkv.double(), Rs[0].double(), Rs[1].double(), Rs[2].double(), torch.ones(B, T).double(), B, T, P, Cc, heads)
    close(o2, ref2.float(), 5e-5)
    # R as tables over 3 timesteps [3][B][T][T][C]: slice rsel[b] per batch element
    n_t = 3
    tabs = [torch.stack([(0.5 + 0.25 * i) * r for i in range(n_t)]).cuda().contiguous() for r in Rs]
    sel = torch.tensor([(2 * b + 1) % n_t for b in range(B)], dtype=torch.int64, device="cuda")
    o3 = torch.full((M, Cc), float("nan"), device="cuda")
    nat.check(nat.lib().lfvdm_attn_temporal_sel(g[0].data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(),
                                                g[4].data_ptr(), o3.data_ptr(), None, B, T, P, Cc, heads, sel.data_ptr(),
                                                nat.stream()), "lfvdm_attn_temporal_sel")
    picked = [torch.stack([(0.5 + 0.25 * int(sel[b])) * r[b] for b in range(B)]) for r in Rs]
    ref3 = _temporal_core_f64(qkv.double(), picked[0].double(), picked[1].double(), picked[2].double(), mask.double(), B, T, P, Cc, heads)
    close(o3, ref3.float(), 5e-5)
    # the same tables as a rolling window of `ring` = 3 timesteps: timestep t lives in slot t % 3 (lfvdm_attn_temporal_ring)
    big = torch.tensor([int(sel[b]) + 3 * (7 + b) for b in range(B)], dtype=torch.int64, device="cuda")
    o4 = torch.full((M, Cc), float("nan"), device="cuda")
    nat.check(nat.lib().lfvdm_attn_temporal_ring(g[0].data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(),
                                                 g[4].data_ptr(), o4.data_ptr(), None, B, T, P, Cc, heads, big.data_ptr(), n_t,
                                                 nat.stream()), "lfvdm_attn_temporal_ring")
    assert torch.equal(o4, o3)


def test_spatial_attention_block(nat):
    """gn_apply + qkv GEMM + spatial core + proj GEMM (residual on the normalised tensor, rpe.py:172) vs oracle; a launch that
    still asks for the removed operand prologue (coefA / coefB) is refused."""
    N, P, Cc, heads = 3, 64, 64, 4
    shapes = {"qkv.weight": (3 * Cc, Cc), "qkv.bias": (3 * Cc,), "proj_out.weight": (Cc, Cc), "proj_out.bias": (Cc,),
              "norm.weight": (Cc,), "norm.bias": (Cc,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("sa." + k, s)) for k, s in shapes.items()}
    x = rnd("sa/x", 1, N, Cc, P)  # (B=1, D=N frames, C, T=P tokens)
    ref, _ = uo.rpe_attention(sd, "p", x, None, None, None, heads, False)
    d = {k: v.cuda() for k, v in sd.items()}
    xc = x[0].permute(0, 2, 1).contiguous().cuda()  # [N][P][C]
    M = N * P
    xn = gn_act(nat, xc, None, Cc, 0, N, P, d["p.norm.weight"], d["p.norm.bias"], None, 1, 0, nat.ACT_NONE)
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.qkv.weight"],
                   bias=d["p.qkv.bias"], Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    cA = torch.ones(N, Cc, device="cuda")
    with pytest.raises(RuntimeError):
        nat.conv_igemm(src0=xn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, coefA=cA, coefB=cA, W=d["p.qkv.weight"],
                       bias=d["p.qkv.bias"], Cout=3 * Cc, out=qkv.clone(), ldo=3 * Cc)
    o = torch.empty(M, Cc, device="cuda")
    nat.attn_spatial(qkv, o, None, N, P, Cc, heads)
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xn, ldr=Cc, out=y, ldo=Cc)
    close(y.view(N, P, Cc).permute(0, 2, 1), ref[0], 1e-4)
    # the affine residual of the epilogue (res * resA[n] + resB[n]): the raw tensor with the GroupNorm coefficients == xn
    cA, cB = torch.empty(N, Cc, device="cuda"), torch.empty(N, Cc, device="cuda")
    nat.gn_coef(xc, None, Cc, 0, N, P, d["p.norm.weight"], d["p.norm.bias"], None, 1, 0, 1e-5, cA, cB)
    y2 = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xc.view(M, Cc), ldr=Cc, resA=cA, resB=cB, out=y2, ldo=Cc)
    close(y2, y, 2e-5)


def test_diffusion_ops(nat):
    B, T, Cc, H = 3, 4, 4, 8
    tab = do.Tables(do.linear_betas(1000))
    x0, noise, x, eps = (rnd(f"df/{n}", B, T, Cc, H, H) for n in ("x0", "noise", "x", "eps"))
    t = torch.tensor([999, 0, 417])
    dev = lambda a: torch.from_numpy(a).float().cuda()
    out = torch.empty(B, T, Cc, H, H, device="cuda")
    nat.q_sample(x0.cuda(), noise.cuda(), t.cuda(), dev(tab.sqrt_alphas_cumprod), dev(tab.sqrt_one_minus_alphas_cumprod), out)
    close(out, do.q_sample(tab, x0, t, noise), 1e-6)
    smp, pred, mean = (torch.empty_like(out) for _ in range(3))
    nat.p_sample(x.cuda(), eps.cuda(), noise.cuda(), t.cuda(), dev(tab.sqrt_recip_alphas_cumprod),
                 dev(tab.sqrt_recipm1_alphas_cumprod), dev(tab.posterior_mean_coef1), dev(tab.posterior_mean_coef2),
                 dev(tab.fixed_large_log_variance), True, smp, pred, mean)
    ref_s, ref_p = do.p_sample(tab, eps, x, t, noise)
    close(smp, ref_s, 1e-5)
    close(pred, ref_p, 1e-4)  # 157x amplification at t=999 before the clamp
    close(mean, do.p_mean_variance(tab, eps, x, t)["mean"], 1e-5)
    mask = torch.tensor([[1., 0, 1, 1], [0, 0, 0, 0], [1, 1, 1, 1]])
    mo = torch.empty(B, device="cuda")
    nat.masked_mse(x.cuda(), eps.cuda(), mask.cuda(), mo, B, T, Cc * H * H)
    close(mo, do.masked_mean_flat((x - eps) ** 2, mask.view(B, T, 1, 1, 1)), 1e-6, rtol=1e-5)


@pytest.mark.parametrize("shape", [(3, 4, 4, 16, 16), (2, 20, 4, 16, 16), (2, 5, 3, 7, 5)])
@pytest.mark.parametrize("with_mask", [True, False])
def test_masked_mse_forward_and_backward(nat, shape, with_mask):
    """_autograd.masked_mse (gaussian_diffusion.py:787-788 mean_flat((target - pred)^2 * mask)): value and gradient w.r.t.
    the prediction vs torch autograd; frame sizes that are / are not multiples of 4."""
    from improved_diffusion._autograd import masked_mse
    B, T = shape[:2]
    tgt, prd = rnd("mm/t", *shape), rnd("mm/p", *shape)
    mask = (torch.rand(B, T, 1, 1, 1) < 0.6).float() if with_mask else None
    pr = prd.double().requires_grad_(True)
    d2 = (tgt.double() - pr) ** 2
    ref = (d2 * mask.double() if with_mask else d2).flatten(1).mean(1)
    w = torch.tensor([0.7, -1.3, 2.1])[:B].double()
    (ref * w).sum().backward()
    pg = prd.cuda().requires_grad_(True)
    out = masked_mse(tgt.cuda(), pg, mask.cuda() if with_mask else None)
    (out * w.float().cuda()).sum().backward()
    close(out, ref.float(), 1e-6, rtol=1e-5)
    close(pg.grad, pr.grad.float(), 1e-7, rtol=1e-5)


def test_conv_splitk_is_exact_and_deterministic(nat):
    """Every legal (tile shape, K-chunk, split-K) variant of one small-M launch gives the same conv, and the
    split-K variants (ordered last-arriver reduction) are bitwise reproducible run to run."""
    import ctypes as C
    N, Cin, Cout, H = 40, 128, 128, 2
    x, w, b = rnd("sk/x", N, Cin, H, H), rnd("sk/w", Cout, Cin, 3, 3, scale=0.05), rnd("sk/b", Cout)
    res = rnd("sk/r", N * H * H, Cout)
    ref = F.conv2d(x, w, b, padding=1) + res.view(N, H, H, Cout).permute(0, 3, 1, 2)
    out = torch.empty(N * H * H, Cout, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    cnt = torch.zeros(1024, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), bias=b.cuda(), res=res.cuda())   # the struct holds raw pointers
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, Cout=Cout, ldr=Cout, out=out, ldo=Cout, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    split = [codes[i] for i in range(n) if (codes[i] - 1) >> 5 > 0]
    assert n >= 8 and len(split) >= 3, "split-K candidates expected for a small-M layer"
    for code in [codes[i] for i in range(n)]:
        a.tune = code
        out.zero_()
        nat.conv_igemm_struct(a)
        close(from_cl(out, N, H, H, Cout), ref, 5e-5)
        first = out.clone()
        nat.conv_igemm_struct(a)
        assert torch.equal(out, first), f"tune code {code} is not reproducible"


# ------------------------------------------------------------------------------------------- round-1 additions
@pytest.mark.parametrize("N,P,C0,C1,film,act", [(4, 256, 64, 0, False, 1), (6, 64, 128, 64, False, 1), (4, 16, 128, 0, True, 1),
                                                (3, 4, 256, 128, False, 1), (2, 256, 64, 0, False, 0), (2, 2500, 96, 32, True, 1),
                                                (2, 256, 128, 64, True, 1), (2, 300, 64, 32, False, 1), (2, 16, 512, 256, True, 1),
                                                (4, 100, 256, 256, False, 1), (2, 37, 512, 0, True, 0)])
def test_gn_apply(nat, N, P, C0, C1, film, act):
    """lfvdm_gn_apply: act(GroupNorm32(cat(a, b)) * (1 + scale) + shift) materialised (nn.py:17-19, unet.py:199-203)
    plus the coefficient / statistics side outputs, vs torch group_norm in fp64."""
    C, T = C0 + C1, 2
    a = rnd("ga/a", N * P, C0) * 1.3 + 0.2
    b = rnd("ga/b", N * P, C1) if C1 else None
    gamma, beta = 1 + 0.1 * rnd("ga/g", C), 0.1 * rnd("ga/be", C)
    fm = 0.3 * rnd("ga/film", N // T, 2 * C) if film else None
    x = torch.cat([a] + ([b] if C1 else []), dim=1).double().view(N, P, C).permute(0, 2, 1)           # N, C, P
    ref = F.group_norm(x, 32, gamma.double(), beta.double(), eps=1e-5)
    if film:
        f = fm.double().repeat_interleave(T, dim=0)
        ref = ref * (1 + f[:, :C, None]) + f[:, C:, None]
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(N * P, C)
    out = torch.full((N * P, C), float("nan"), device="cuda")
    cA, cB, st = torch.empty(N, C, device="cuda"), torch.empty(N, C, device="cuda"), torch.empty(N, 32, 2, device="cuda")
    g = [t.cuda() if t is not None else None for t in (a, b, gamma, beta, fm)]
    nat.check(nat.lib().lfvdm_gn_apply(nat.ptr(g[0]), nat.ptr(g[1]), C0, C1, N, P, nat.ptr(g[2]), nat.ptr(g[3]), nat.ptr(g[4]),
                                       T if film else 1, 2 * C if film else 0, 1e-5, act, nat.ptr(out), nat.ptr(cA), nat.ptr(cB),
                                       nat.ptr(st), nat.stream()), "lfvdm_gn_apply")
    close(out, ref.float(), 3e-5)
    mean = x.reshape(N, 32, -1).mean(-1)
    close(st[..., 0], mean.float(), 1e-5)
    # the coefficients reproduce the same affine map
    pre = (x.permute(0, 2, 1) * cA.cpu().double()[:, None, :] + cB.cpu().double()[:, None, :]).reshape(N * P, C)
    close(F.silu(pre) if act else pre, ref.float(), 5e-5)


@pytest.mark.parametrize("N,P,C0,C1,film,act", [(2, 2500, 96, 32, True, 1), (4, 1061, 64, 0, False, 1), (2, 4096, 128, 64, True, 1),
                                                (3, 16384, 128, 0, False, 1), (2, 300, 512, 256, False, 0),
                                                (4, 256, 64, 0, False, 1)])
def test_gn_apply_large_maps(nat, N, P, C0, C1, film, act):
    """lfvdm_gn_apply_ws: the chunked two-launch GroupNorm for maps whose (sample, 8 groups) slice exceeds one workgroup's
    registers (chunk statistics + fixed-order combination), ragged last chunks and non-power-of-two channel quads included;
    the last case fits and must take the single-launch kernel (workspace size 0)."""
    C, T = C0 + C1, 2
    a = rnd("gl/a", N * P, C0) * 1.3 + 0.7          # a mean well away from zero: a naive sum of squares would show
    b = rnd("gl/b", N * P, C1) if C1 else None
    gamma, beta = 1 + 0.1 * rnd("gl/g", C), 0.1 * rnd("gl/be", C)
    fm = 0.3 * rnd("gl/film", N // T, 2 * C) if film else None
    x = torch.cat([a] + ([b] if C1 else []), dim=1).double().view(N, P, C).permute(0, 2, 1)
    ref = F.group_norm(x, 32, gamma.double(), beta.double(), eps=1e-5)
    if film:
        f = fm.double().repeat_interleave(T, dim=0)
        ref = ref * (1 + f[:, :C, None]) + f[:, C:, None]
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(N * P, C)
    L = nat.lib()
    need = int(L.lfvdm_gn_apply_ws_floats(C, N, P))
    assert (need == 0) == (P == 256)
    ws = torch.full((max(need, 1),), float("nan"), device="cuda")
    out = torch.full((N * P, C), float("nan"), device="cuda")
    cA, cB, st = torch.empty(N, C, device="cuda"), torch.empty(N, C, device="cuda"), torch.empty(N, 32, 2, device="cuda")
    g = [t.cuda() if t is not None else None for t in (a, b, gamma, beta, fm)]
    args = (nat.ptr(g[0]), nat.ptr(g[1]), C0, C1, N, P, nat.ptr(g[2]), nat.ptr(g[3]), nat.ptr(g[4]), T if film else 1,
            2 * C if film else 0, 1e-5, act, nat.ptr(out), nat.ptr(cA), nat.ptr(cB), nat.ptr(st))
    nat.check(L.lfvdm_gn_apply_ws(*args, nat.ptr(ws), need, nat.stream()), "lfvdm_gn_apply_ws")
    close(out, ref.float(), 3e-5)
    grp = x.reshape(N, 32, -1)
    close(st[..., 0], grp.mean(-1).float(), 1e-5)
    close(st[..., 1], (1.0 / torch.sqrt(grp.var(-1, unbiased=False) + 1e-5)).float(), 1e-5)
    # same values as the single-launch kernel (which walks the whole slice in one workgroup), and reproducible
    out1 = torch.empty_like(out)
    nat.check(L.lfvdm_gn_apply(*args[:13], nat.ptr(out1), None, None, None, nat.stream()), "lfvdm_gn_apply")
    close(out, out1, 2e-6)
    out2 = torch.empty_like(out)
    nat.check(L.lfvdm_gn_apply_ws(*args[:13], nat.ptr(out2), None, None, None, nat.ptr(ws), need, nat.stream()), "lfvdm_gn_apply_ws")
    assert torch.equal(out, out2)
    if need:
        assert L.lfvdm_gn_apply_ws(*args, None, 0, nat.stream()) != 0, "a missing workspace must be refused"


@pytest.mark.parametrize("M,K", [(3, 256), (2, 1024), (6, 512)])
def test_rowdot_backward(nat, M, K):
    """lfvdm_rowdot_bwd (grouped nn.Linear backward) vs autograd: two jobs sharing one input, one with SiLU in front.
    The 17 tasks make one workgroup that spans both jobs (per-task atomics) and one inside a job (workgroup-level sum of the
    input-gradient partials); M = 6 takes the path for more than four batch rows."""
    x = rnd("rb/x", M, K)
    Ws = [0.1 * rnd("rb/w0", 96, K), 0.1 * rnd("rb/w1", 40, K)]
    douts = [rnd("rb/d0", M, 96), rnd("rb/d1", M, 40)]
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in Ws]
    act = F.silu(xr)
    act.retain_grad()
    y0, y1 = act @ wr[0].t(), xr @ wr[1].t()
    (y0 * douts[0].double()).sum().backward(retain_graph=True, inputs=[act, wr[0]])
    g_act = act.grad.clone()
    (y1 * douts[1].double()).sum().backward(inputs=[xr, wr[1]])
    dev = lambda t: t.cuda().contiguous()
    xd, Wd, dd = dev(x), [dev(w) for w in Ws], [dev(d) for d in douts]
    dW = [torch.zeros_like(w) for w in Wd]
    db = [torch.zeros(w.shape[0], device="cuda") for w in Wd]
    din = [torch.zeros(M, K, device="cuda"), torch.zeros(M, K, device="cuda")]
    jobs = [nat.RowdotBwdJob(Wd[0].data_ptr(), xd.data_ptr(), dd[0].data_ptr(), dW[0].data_ptr(), db[0].data_ptr(),
                             din[0].data_ptr(), K, 96, M, K, 96, K, 1, 0),
            nat.RowdotBwdJob(Wd[1].data_ptr(), xd.data_ptr(), dd[1].data_ptr(), dW[1].data_ptr(), db[1].data_ptr(),
                             din[1].data_ptr(), K, 40, M, K, 40, K, 0, 12)]
    table = nat.jobs_to_device(jobs, "cuda")
    nat.check(nat.lib().lfvdm_rowdot_bwd(table.data_ptr(), 2, 12 + 5, nat.stream()), "lfvdm_rowdot_bwd")
    close(dW[0], wr[0].grad.float(), 2e-5)
    close(dW[1], wr[1].grad.float(), 2e-5)
    close(db[0], douts[0].sum(0), 1e-5)
    close(db[1], douts[1].sum(0), 1e-5)
    close(din[0], g_act.float(), 2e-5)          # gradient w.r.t. the ACTIVATED input of job 0
    close(din[1], xr.grad.float(), 2e-5)


def test_rpe_front_and_backward(nat):
    """lfvdm_rpe_front(_bwd): hidden layer of an RPENet (rpe.py:20-31) with a strided time projection."""
    B, T, C = 2, 5, 64
    rows = B * T * T
    tp_full = rnd("rf/tp", B, 3 * C)                      # the kernel reads columns [C, 2C) with row stride 3C
    feats, wd, bd, d_act = rnd("rf/f", rows, 3).abs(), 0.5 * rnd("rf/wd", C, 3), 0.1 * rnd("rf/bd", C), rnd("rf/da", rows, C)
    tpr = tp_full[:, C:2 * C].double().requires_grad_(True)
    wdr, bdr = wd.double().requires_grad_(True), bd.double().requires_grad_(True)
    hid = tpr.repeat_interleave(T * T, 0) + feats.double() @ wdr.t() + bdr
    ref = F.silu(hid)
    (ref * d_act.double()).sum().backward()
    g = {k: v.cuda().contiguous() for k, v in dict(tp=tp_full, f=feats, wd=wd, bd=bd, da=d_act).items()}
    tp_view = g["tp"][:, C:2 * C]
    act = torch.empty(rows, C, device="cuda")
    nat.check(nat.lib().lfvdm_rpe_front(tp_view.data_ptr(), tp_view.stride(0), nat.ptr(g["f"]), nat.ptr(g["wd"]), nat.ptr(g["bd"]),
                                        nat.ptr(act), B, T * T, C, nat.stream()), "lfvdm_rpe_front")
    close(act, ref.detach().float(), 2e-5)
    dtp, dwd, dbd = torch.zeros(B, 2 * C, device="cuda"), torch.zeros(C, 3, device="cuda"), torch.zeros(C, device="cuda")
    slot = dtp[:, C:]
    nat.check(nat.lib().lfvdm_rpe_front_bwd(tp_view.data_ptr(), tp_view.stride(0), nat.ptr(g["f"]), nat.ptr(g["wd"]),
                                            nat.ptr(g["bd"]), nat.ptr(g["da"]), slot.data_ptr(), slot.stride(0), nat.ptr(dwd),
                                            nat.ptr(dbd), B, T * T, C, nat.stream()), "lfvdm_rpe_front_bwd")
    close(slot, tpr.grad.float(), 1e-4)
    close(dwd, wdr.grad.float(), 1e-4)
    close(dbd, bdr.grad.float(), 1e-4)
    assert float(dtp[:, :C].abs().max()) == 0.0


def test_grouped_pack_and_unpack(nat):
    """lfvdm_pack_conv_weights == per-weight packing; lfvdm_unpack_conv_grads folds and re-zeroes the accumulators."""
    ws = [rnd("gp/w0", 64, 32, 3, 3).cuda(), rnd("gp/w1", 96, 64, 1, 1).cuda(), rnd("gp/w2", 32, 128, 3, 3).cuda(),
          rnd("gp/w3", 4, 40, 3, 3).cuda(), rnd("gp/w4", 72, 5, 3, 3).cuda()]
    jobs, outs, blk = [], [], 0
    for w in ws:
        for tr in (0, 1):
            Cout, Cin, k, _ = w.shape
            o = torch.full((Cin, k * k, Cout) if tr else (Cout, k * k, Cin), float("nan"), device="cuda")
            jobs.append(nat.PackJob(w.data_ptr(), o.data_ptr(), Cout, Cin, k * k, tr, blk, 0))
            outs.append((w, o, tr))
            blk += ((Cout + 31) // 32) * ((Cin + 31) // 32)
    table = nat.jobs_to_device(jobs, "cuda")
    nat.check(nat.lib().lfvdm_pack_conv_weights(table.data_ptr(), len(jobs), blk, nat.stream()), "lfvdm_pack_conv_weights")
    for w, o, tr in outs:
        ref = torch.empty_like(o)
        (nat.pack_conv_weight_t if tr else nat.pack_conv_weight)(w, ref)
        assert torch.equal(o, ref)
        if not tr:
            assert torch.equal(o, w.permute(0, 2, 3, 1).reshape(o.shape))
    # unpack: g += unpack(gp); gp = 0
    gps = [rnd("gp/g0", 64, 9, 32).cuda(), rnd("gp/g2", 32, 9, 128).cuda()]
    gs = [rnd("gp/a0", 64, 32, 3, 3).cuda(), rnd("gp/a2", 32, 128, 3, 3).cuda()]
    want = [g + gp.view(gp.shape[0], 3, 3, gp.shape[2]).permute(0, 3, 1, 2) for g, gp in zip(gs, gps)]
    uj, row0 = [], 0
    for g, gp in zip(gs, gps):
        uj.append(nat.UnpackJob(gp.data_ptr(), g.data_ptr(), g.shape[0], g.shape[1], 9, row0))
        row0 += g.shape[0]
    ut = nat.jobs_to_device(uj, "cuda")
    nat.check(nat.lib().lfvdm_unpack_conv_grads(ut.data_ptr(), 2, row0, 9 * 128, nat.stream()), "lfvdm_unpack_conv_grads")
    for g, w_, gp in zip(gs, want, gps):
        assert torch.allclose(g, w_, atol=1e-6) and float(gp.abs().max()) == 0.0


def test_gn_param_grads(nat):
    """lfvdm_gn_param_grads vs the closed-form sums it implements (FiLM and plain)."""
    N, T, C = 6, 3, 96
    sums, gamma, beta = rnd("gg/s", N, C, 2), 1 + 0.1 * rnd("gg/g", C), 0.1 * rnd("gg/b", C)
    film = 0.3 * rnd("gg/f", N // T, 2 * C)
    s1, s2 = sums[..., 0].double(), sums[..., 1].double()
    sc1 = 1 + film[:, :C].double().repeat_interleave(T, 0)
    for use_film in (True, False):
        dg0, db0 = rnd("gg/dg", C), rnd("gg/db", C)
        dg, db = dg0.cuda().clone(), db0.cuda().clone()
        dfilm = torch.full((N // T, 2 * C), float("nan"), device="cuda") if use_film else None
        g = [t.cuda() for t in (sums, gamma, beta, film)]
        nat.check(nat.lib().lfvdm_gn_param_grads(nat.ptr(g[0]), nat.ptr(g[1]), nat.ptr(g[2]), nat.ptr(g[3]) if use_film else None,
                                                 2 * C if use_film else 0, T, nat.ptr(dg), nat.ptr(db), nat.ptr(dfilm),
                                                 2 * C if use_film else 0, N, C, nat.stream()), "lfvdm_gn_param_grads")
        if use_film:
            close(dg, (dg0.double() + (s2 * sc1).sum(0)).float(), 2e-5)
            close(db, (db0.double() + (s1 * sc1).sum(0)).float(), 2e-5)
            dsc = (s2 * gamma.double() + s1 * beta.double()).view(N // T, T, C).sum(1)
            close(dfilm, torch.cat([dsc, s1.view(N // T, T, C).sum(1)], 1).float(), 2e-5)
        else:
            close(dg, (dg0.double() + s2.sum(0)).float(), 2e-5)
            close(db, (db0.double() + s1.sum(0)).float(), 2e-5)


@pytest.mark.parametrize("B,T,P,C", [(2, 20, 16, 128), (1, 14, 5, 256), (2, 20, 4, 64), (1, 32, 3, 96), (2, 7, 6, 384)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_gn_temporal_backward(nat, B, T, P, C, accumulate):
    """lfvdm_gn_temporal_bwd (autograd of the GroupNorm over (C/32, T) per (b, pixel), rpe.py:135-137): dx, dgamma, dbeta vs
    torch autograd in fp64 - the register-resident kernel (C <= 256) and the general one (96 / 384 channels), ragged
    workgroups (B*P not a multiple of 4), dx written or accumulated."""
    x, dy = rnd("gtb/x", B * T * P, C) * 1.2 + 0.3, rnd("gtb/dy", B * T * P, C)
    gamma, beta = 1 + 0.1 * rnd("gtb/g", C), 0.1 * rnd("gtb/b", C)
    xr = x.double().view(B, T, P, C).permute(0, 2, 3, 1).reshape(B * P, C, T).requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.group_norm(xr, 32, gr, br, eps=1e-5)
    y.backward(dy.double().view(B, T, P, C).permute(0, 2, 3, 1).reshape(B * P, C, T))
    want_dx = xr.grad.view(B, P, C, T).permute(0, 3, 1, 2).reshape(B * T * P, C)
    dx0 = rnd("gtb/dx0", B * T * P, C)
    dx = dx0.cuda().clone()
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    g = [t.cuda().contiguous() for t in (x, dy, gamma)]
    nat.check(nat.lib().lfvdm_gn_temporal_bwd(nat.ptr(g[0]), nat.ptr(g[1]), nat.ptr(g[2]), 1e-5, nat.ptr(dx), nat.ptr(dg), nat.ptr(db),
                                              B, T, P, C, accumulate, nat.stream()), "lfvdm_gn_temporal_bwd")
    close(dx, (want_dx + (dx0.double() if accumulate else 0)).float(), 5e-5)
    close(dg, gr.grad.float(), 1e-4 * max(1.0, float(gr.grad.abs().max())))
    close(db, br.grad.float(), 1e-4 * max(1.0, float(br.grad.abs().max())))


def test_p_sample_rng_draws_standard_normal_noise(nat):
    """lfvdm_p_sample_rng: the update of lfvdm_p_sample with the noise drawn in the kernel (Philox4x32-10 + Box-Muller,
    keyed by a per-chain seed, counter = (element, batch row, timestep)).  The values it reports are the ones it used, they
    are standard normal, a (seed, t, element) triple is reproducible, other timesteps / seeds / rows are independent, t = 0
    adds no noise (gaussian_diffusion.py:396-400)."""
    from scipy import stats
    B, inner, S = 4, 262144 + 3, 1000
    x, eps = rnd("pr/x", B, inner).cuda(), rnd("pr/e", B, inner).cuda()
    tabs = [torch.rand(S, device="cuda") + 0.5 for _ in range(4)] + [torch.randn(S, device="cuda") * 0.3]
    seed = torch.tensor([0x1234567, ], dtype=torch.int64, device="cuda")

    def run(t, sd, want_noise=True):
        tt = torch.tensor(t, dtype=torch.int64, device="cuda")
        out, nz = torch.empty(B, inner, device="cuda"), torch.empty(B, inner, device="cuda") if want_noise else None
        nat.p_sample_rng(x, eps, nz, tt, *tabs, True, out, sd, None, None)
        return out, nz, tt

    out, nz, tt = run([7, 7, 500, 0], seed)
    ref = torch.empty_like(out)
    nat.p_sample(x, eps, nz, tt, *tabs, True, ref)                        # the plain kernel on the reported noise
    assert torch.equal(out, ref)
    z = nz[:3].flatten().double().cpu().numpy()                            # (row 3 is t = 0: its noise is not used)
    assert abs(z.mean()) < 4e-3 and abs(z.var() - 1) < 6e-3 and abs(stats.kurtosis(z)) < 0.03 and abs(stats.skew(z)) < 0.01
    assert stats.kstest(z[:200000], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 4e-3                    # neighbours (Box-Muller pairs included)
    assert abs(float(torch.corrcoef(torch.stack([nz[0], nz[1]]))[0, 1])) < 6e-3      # same t, different rows
    assert abs(float(torch.corrcoef(torch.stack([nz[0], nz[2]]))[0, 1])) < 6e-3      # different t
    out2, nz2, _ = run([7, 7, 500, 0], seed)
    assert torch.equal(out, out2) and torch.equal(nz, nz2)                 # reproducible
    _, nz3, _ = run([7, 7, 500, 0], seed + 1)
    assert abs(float(torch.corrcoef(torch.stack([nz[0], nz3[0]]))[0, 1])) < 6e-3     # another chain
    out0, _, _ = run([7, 7, 500, 0], seed, want_noise=False)
    assert torch.equal(out0, out)                                          # the noise output is optional
    mean = torch.empty_like(out)
    nat.p_sample(x, eps, torch.zeros_like(x), tt, *tabs, True, torch.empty_like(out), None, mean)
    assert torch.equal(out[3], mean[3])                                    # t = 0: the sample is the posterior mean


@pytest.mark.parametrize("B,T,H,W,C,Cout", [(2, 20, 16, 16, 64, 4), (1, 3, 8, 12, 128, 3), (2, 2, 4, 4, 256, 4)])
def test_conv_out_psample_equals_conv_then_update(nat, B, T, H, W, C, Cout):
    """lfvdm_conv_out_psample = the output convolution (unet.py:402,462-464) + lfvdm_p_sample_rng in one launch: its eps
    equals the dense convolution, its noise is the SAME stream as lfvdm_p_sample_rng's for the chain's seed (bitwise), and
    its sample / pred_xstart are bitwise what the two-launch form gives on that eps; with injected noise it uses that noise."""
    N = B * T
    a = rnd("co/a", N, C, H, W)
    w, bias = rnd("co/w", Cout, C, 3, 3, scale=0.05), rnd("co/b", Cout)
    ref_eps = F.conv2d(a, w, bias, padding=1).view(B, T, Cout, H, W)
    x = rnd("co/x", B, T, Cout, H, W).cuda()
    S = 1000
    tabs = [torch.rand(S, device="cuda") + 0.5 for _ in range(4)] + [torch.randn(S, device="cuda") * 0.3]
    t = torch.tensor(([7, 0] * B)[:B], dtype=torch.int64, device="cuda")
    seed = torch.tensor([0x7654321], dtype=torch.int64, device="cuda")
    act, wp = cl(a), packed(nat, w)
    eps, nz, out, pred = (torch.empty_like(x) for _ in range(4))
    nat.conv_out_psample(act, wp, bias.cuda(), eps, x, None, nz, t, *tabs, True, out, seed, pred)
    err = float((eps.cpu() - ref_eps).abs().max())
    assert err < 5e-5, err
    out2, nz2, pred2 = (torch.empty_like(x) for _ in range(3))
    nat.p_sample_rng(x.view(B, -1), eps.view(B, -1), nz2.view(B, -1), t, *tabs, True, out2.view(B, -1), seed, pred2.view(B, -1))
    assert torch.equal(nz, nz2) and torch.equal(out, out2) and torch.equal(pred, pred2)
    given = rnd("co/n", B, T, Cout, H, W).cuda()
    out3, out4 = torch.empty_like(x), torch.empty_like(x)
    nat.conv_out_psample(act, wp, bias.cuda(), None, x, given, None, t, *tabs, True, out3, None)
    nat.p_sample(x.view(B, -1), eps.view(B, -1), given.view(B, -1), t, *tabs, True, out4.view(B, -1))
    assert torch.equal(out3, out4)
    xin = x.clone()                                     # in place, as the sampler runs it
    nat.conv_out_psample(act, wp, bias.cuda(), None, xin, None, None, t, *tabs, True, xin, seed)
    assert torch.equal(xin, out)


def test_sampler_tick(nat):
    t = torch.tensor([5, 0, 999], dtype=torch.int64, device="cuda")
    table = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.25
    mt = torch.zeros(3, device="cuda")
    nat.check(nat.lib().lfvdm_sampler_tick(t.data_ptr(), table.data_ptr(), mt.data_ptr(), 3, nat.stream()), "lfvdm_sampler_tick")
    assert t.tolist() == [4, 0, 998] and mt.tolist() == [1.0, 0.0, 249.5]


@pytest.mark.parametrize("N,Cin,Cout,H,k,stride", [(3, 64, 64, 16, 3, 1), (2, 128, 192, 5, 3, 1), (5, 64, 64, 8, 3, 2),
                                                   (7, 128, 32, 6, 1, 1), (40, 128, 128, 4, 3, 1), (4, 64, 4, 16, 3, 1),
                                                   (21, 256, 128, 16, 3, 1)])
def test_conv_every_tune_code(nat, N, Cin, Cout, H, k, stride):
    """Every legal launch variant - register-staged and LDS-DMA (buffer_load ... lds, 2 and 3 stages), both K-chunk
    widths, split-K and tail-split - computes the same convolution (odd sizes: ragged last tiles, image borders)."""
    import ctypes as C
    pad = 1 if k == 3 else 0
    x, w, b = rnd("et/x", N, Cin, H, H), rnd("et/w", Cout, Cin, k, k, scale=0.05), rnd("et/b", Cout)
    ref = F.conv2d(x, w, b, padding=pad, stride=stride)
    Ho = ref.shape[2]
    out = torch.empty(N * Ho * Ho, Cout, device="cuda")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), bias=b.cuda())
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=Ho, Wo=Ho, Cout=Cout, out=out, ldo=Cout, ksize=k, stride=stride, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    dma = [codes[i] for i in range(n) if (codes[i] - 1) >> 8]
    assert n > 0 and dma, "LDS-DMA variants expected for a plain convolution"
    for code in [0] + [codes[i] for i in range(n)]:
        a.tune = code
        out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        got = from_cl(out, N, Ho, Ho, Cout)
        err = float((got.cpu() - ref).abs().max())
        assert err < 5e-5, f"tune code {code}: max|d| = {err:.3e}"
    assert int(cnt.abs().sum()) == 0, "split-K tickets must be left at zero"


@pytest.mark.parametrize("N,Cin,Cout,H,up", [(3, 64, 64, 8, 1), (2, 128, 96, 5, 1), (40, 128, 128, 2, 1), (3, 64, 64, 7, 2),
                                             (5, 128, 64, 4, 2)])
def test_conv_upsampled_source_every_tune_code(nat, N, Cin, Cout, H, up):
    """3x3 conv over a nearest-2x upsampled source (Upsample, unet.py:60-83) and over a zero-inserted one (the data
    gradient of a stride-2 conv as a transposed convolution): every launch variant, including the LDS-DMA loop whose
    per-lane source offsets are rebuilt per tap for these two modes."""
    import ctypes as C
    x, w, b = rnd("eu/x", N, Cin, H, H), rnd("eu/w", Cout, Cin, 3, 3, scale=0.05), rnd("eu/b", Cout)
    if up == 1:
        xin = F.interpolate(x, scale_factor=2, mode="nearest")
    else:
        xin = torch.zeros(N, Cin, 2 * H, 2 * H)
        xin[:, :, ::2, ::2] = x
    ref = F.conv2d(xin, w, b, padding=1)
    Ho = 2 * H
    out = torch.empty(N * Ho * Ho, Cout, device="cuda")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), bias=b.cuda())
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=Ho, Wo=Ho, Cout=Cout, out=out, ldo=Cout, ksize=3, up=up, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    dma = [codes[i] for i in range(n) if (codes[i] - 1) >> 8]
    assert n > 0 and dma, "LDS-DMA variants expected for an upsampled source"
    for code in [0] + [codes[i] for i in range(n)]:
        a.tune = code
        out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        err = float((from_cl(out, N, Ho, Ho, Cout).cpu() - ref).abs().max())
        assert err < 5e-5, f"tune code {code}: max|d| = {err:.3e}"
    assert int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("N,Cin,Cout,H", [(3, 64, 64, 7), (40, 128, 128, 2), (2, 96, 160, 16)])
def test_conv_zero_inserted_source_with_residual(nat, N, Cin, Cout, H):
    """The zero-inserted source is computed by output parity classes (rows enumerated on the source grid, 1 / 2 / 2 / 4
    live taps): the residual and the output rows go through the class -> output-row map, so check a launch with a
    residual - what the data gradient of a Downsample conv with a skip connection's gradient riding along issues -
    over every launch variant, against a dense convolution of the explicitly zero-filled image."""
    import ctypes as C
    x, w = rnd("ez/x", N, Cin, H, H), rnd("ez/w", Cout, Cin, 3, 3, scale=0.05)
    Ho = 2 * H
    r = rnd("ez/r", N, Cout, Ho, Ho)
    xin = torch.zeros(N, Cin, Ho, Ho)
    xin[:, :, ::2, ::2] = x
    ref = F.conv2d(xin, w, None, padding=1) + r
    out = torch.empty(N * Ho * Ho, Cout, device="cuda")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), res=cl(r))
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=Ho, Wo=Ho, Cout=Cout, out=out, ldo=Cout, ksize=3, up=2, ldr=Cout, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    assert n > 0
    for code in [0] + [codes[i] for i in range(n)]:
        a.tune = code
        out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        err = float((from_cl(out, N, Ho, Ho, Cout).cpu() - ref).abs().max())
        assert err < 5e-5, f"tune code {code}: max|d| = {err:.3e}"
    assert int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("N,C0,C1,S0,S1,Cout,H", [(5, 64, 32, 64, 32, 64, 8), (3, 128, 0, 96, 32, 96, 5), (40, 64, 64, 0, 0, 128, 4),
                                                  (2, 32, 32, 32, 0, 32, 16)])
def test_conv_concat_and_skip_segment_every_tune_code(nat, N, C0, C1, S0, S1, Cout, H):
    """Virtual concat (src0 | src1) 3x3 conv + fused 1x1 skip segment on a second raw concat (s2src0 | s2src1) +
    residual (the second conv of a channel-changing ResBlock, unet.py:173-207): every launch variant, including the
    general LDS-DMA loop with its per-chunk descriptor selects, gives the same result."""
    import ctypes as C
    Cin, C2 = C0 + C1, S0 + S1
    x, w, b = rnd("cs2/x", N, Cin, H, H), rnd("cs2/w", Cout, Cin, 3, 3, scale=0.05), rnd("cs2/b", Cout)
    ref = F.conv2d(x, w, b, padding=1)
    keep = dict(src0=cl(x[:, :C0]), W=packed(nat, w), bias=b.cuda())
    kw = dict(C0=C0, C1=C1)
    if C1:
        keep["src1"] = cl(x[:, C0:])
    if C2:
        s, w2, b2 = rnd("cs2/s", N, C2, H, H), rnd("cs2/w2", Cout, C2, scale=0.05), rnd("cs2/b2", Cout)
        ref = ref + F.conv2d(s, w2.view(Cout, C2, 1, 1), b2)
        keep.update(s2src0=cl(s[:, :S0]), W2=w2.cuda().contiguous(), bias2=b2.cuda())
        kw.update(s2C0=S0, s2C1=S1)
        if S1:
            keep["s2src1"] = cl(s[:, S0:])
    else:
        r = rnd("cs2/r", N * H * H, Cout)
        ref = ref + r.view(N, H, H, Cout).permute(0, 3, 1, 2)
        keep["res"] = r.cuda()
        kw["ldr"] = Cout
    out = torch.empty(N * H * H, Cout, device="cuda")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    a = nat.fill_conv_args(N=N, Hs=H, Ws=H, Ho=H, Wo=H, Cout=Cout, out=out, ldo=Cout, **kw, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    assert n > 0 and any((codes[i] - 1) >> 8 for i in range(n))
    for code in [0] + [codes[i] for i in range(n)]:
        a.tune = code
        out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        err = float((from_cl(out, N, H, H, Cout).cpu() - ref).abs().max())
        assert err < 5e-5, f"tune code {code}: max|d| = {err:.3e}"


@pytest.mark.parametrize("N,C0,C1,Cout,H,k,stride,up", [(10, 128, 0, 128, 16, 3, 1, 0), (5, 64, 0, 96, 5, 3, 1, 0),
                                                        (6, 64, 64, 160, 8, 1, 1, 0), (4, 64, 0, 64, 8, 3, 2, 0),
                                                        (3, 128, 0, 128, 4, 3, 1, 1), (40, 256, 0, 128, 2, 3, 1, 0),
                                                        (2, 32, 0, 32, 6, 3, 1, 0)])
@pytest.mark.parametrize("variant", ["dma3", "dma2", "regs"])
def test_conv_wgrad_matches_autograd(nat, monkeypatch, N, C0, C1, Cout, H, k, stride, up, variant):
    """lfvdm_conv_wgrad (weight + bias gradient of Conv2d / Linear, train_util.py:328 loss.backward()) vs torch
    autograd: the LDS-DMA kernels (2 / 3 stages) and the register-staged ones, concat sources, stride 2, nearest
    upsampling, ragged M, filter counts that do not fill the last tile."""
    import os
    for key in ("LFVDM_WGRAD_NO_DMA", "LFVDM_WGRAD_STAGES"):
        monkeypatch.delenv(key, raising=False)
    if variant == "regs":
        monkeypatch.setenv("LFVDM_WGRAD_NO_DMA", "1")
    else:
        monkeypatch.setenv("LFVDM_WGRAD_STAGES", variant[-1])
    Cin = C0 + C1
    x = rnd("wg/x", N, Cin, H, H)
    w = (rnd("wg/w", Cout, Cin, k, k, scale=0.05)).requires_grad_(True)
    b = rnd("wg/b", Cout).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y = F.conv2d(xin, w, b, padding=1 if k == 3 else 0, stride=stride)
    Ho = y.shape[2]
    dout = rnd("wg/d", N, Cout, Ho, Ho)
    y.backward(dout)
    gp = torch.zeros(Cout, k * k, Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    kw = dict(src0=cl(x[:, :C0]), C0=C0, C1=C1, N=N, Hs=H, Ws=H, Ho=Ho, Wo=Ho, ksize=k, stride=stride, up=up,
              res=cl(dout), ldr=Cout, out=gp, bias=db, Cout=Cout)
    if C1:
        kw["src1"] = cl(x[:, C0:])
    nat.conv_wgrad(**kw)
    got_w = gp.view(Cout, k, k, Cin).permute(0, 3, 1, 2).cpu()
    scale = float(w.grad.abs().max())
    assert float((got_w - w.grad).abs().max()) < 2e-5 * max(1.0, scale), float((got_w - w.grad).abs().max())
    assert float((db.cpu() - b.grad).abs().max()) < 2e-5 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize("N,Cin,Cout,H,film,skip_raw,k", [(6, 64, 128, 8, True, True, 3), (8, 128, 128, 4, True, False, 3),
                                                         (40, 128, 64, 2, False, False, 3), (4, 64, 256, 8, True, True, 3),
                                                         (6, 64, 64, 4, False, False, 1)])
def test_conv_fused_output_groupnorm_every_tune_code(nat, N, Cin, Cout, H, film, skip_raw, k):
    """Conv + the NEXT layer's GroupNorm32(+FiLM)+SiLU in the conv's epilogue (lfvdm_conv_args gn_*): the second
    normalisation of a ResBlock, unet.py:199-203.  Every offered tile holds whole samples and groups; each variant must
    match conv2d -> group_norm -> scale/shift -> silu in fp64."""
    import ctypes as C
    T = 2
    x, w, b = rnd("gnf/x", N, Cin, H, H), rnd("gnf/w", Cout, Cin, k, k, scale=0.05), rnd("gnf/b", Cout)
    gamma, beta = 1 + 0.1 * rnd("gnf/g", Cout), 0.1 * rnd("gnf/be", Cout)
    fm = 0.3 * rnd("gnf/film", N // T, 2 * Cout) if film else None
    raw = F.conv2d(x.double(), w.double(), b.double(), padding=1 if k == 3 else 0)
    ref = F.group_norm(raw, 32, gamma.double(), beta.double(), eps=1e-5)
    if film:
        f = fm.double().repeat_interleave(T, dim=0)
        ref = ref * (1 + f[:, :Cout, None, None]) + f[:, Cout:, None, None]
    ref = F.silu(ref).float()
    out = torch.empty(N * H * H, Cout, device="cuda")
    gn_out = torch.empty(N * H * H, Cout, device="cuda")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), bias=b.cuda(), gn_gamma=gamma.cuda(), gn_beta=beta.cuda(),
                gn_film=fm.cuda() if film else None)
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, Cout=Cout, out=out, ldo=Cout, ksize=k, gn_out=gn_out,
                           gn_film_div=T, gn_act=nat.ACT_SILU, gn_skip_raw=skip_raw, **keep)
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    assert n > 0
    # both forms of the epilogue: registers + lane butterflies (P and Cout/32 powers of two: every case here) and the general
    # LDS-tile form (gn_general, what other shapes get)
    for code, general in [(c, g) for c in [0] + [codes[i] for i in range(n)] for g in (0, 1)]:
        a.tune, a.gn_general = code, general
        out.fill_(float("nan")); gn_out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        err = float((from_cl(gn_out, N, H, H, Cout).cpu() - ref).abs().max())
        assert err < 1e-4, f"tune code {code}, general form {general}: fused GroupNorm max|d| = {err:.3e}"
        if skip_raw:
            assert bool(torch.isnan(out).all()), "raw output must not be written with gn_skip_raw"
        else:
            assert float((from_cl(out, N, H, H, Cout).cpu() - raw.float()).abs().max()) < 5e-5
    # a 16x16 map (256 rows per sample) fits no tile: the launch must be refused, not silently wrong
    x2 = rnd("gnf/x2", 2, Cin, 16, 16)
    o2 = torch.empty(2 * 256, Cout, device="cuda")
    with pytest.raises(RuntimeError):
        nat.conv_igemm(src0=cl(x2), C0=Cin, N=2, Hs=16, Ws=16, Ho=16, Wo=16, W=keep["W"], bias=keep["bias"], Cout=Cout, out=o2,
                       ldo=Cout, ksize=k, gn_out=torch.empty_like(o2), gn_gamma=keep["gn_gamma"], gn_beta=keep["gn_beta"])


# ----------------------------------------------------------------------------------------------------------------------
# Op-level fixtures generated by the REFERENCE's own modules (tests/golden/ops.npz, oracle/make_golden.py::gen_ops):
# the attention and resampling ops are pinned to the reference directly, not only through whole-network goldens.
def _ops_golden():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def test_temporal_attention_ops_match_the_reference_fixture(nat):
    """gn_temporal + qkv GEMM + RPE nets + temporal core + proj GEMM == reference RPEAttention._forward (rpe.py:133-174,
    temporal instance with rpe_q/k/v, two-clique mask, sparse frame indices): output AND attention probabilities."""
    g = _ops_golden()
    B, P, Cc, T, heads, ted = 2, 6, 64, 5, 4, 128
    shapes = {"qkv.weight": (3 * Cc, Cc), "qkv.bias": (3 * Cc,), "proj_out.weight": (Cc, Cc), "proj_out.bias": (Cc,),
              "norm.weight": (Cc,), "norm.bias": (Cc,)}
    for r in ("rpe_q", "rpe_k", "rpe_v"):
        shapes.update({f"{r}.rpe_net.embed_distances.weight": (Cc, 3), f"{r}.rpe_net.embed_distances.bias": (Cc,),
                       f"{r}.rpe_net.embed_diffusion_time.weight": (Cc, ted), f"{r}.rpe_net.embed_diffusion_time.bias": (Cc,),
                       f"{r}.rpe_net.out.weight": (Cc, Cc), f"{r}.rpe_net.out.bias": (Cc,)})
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("ops/att_temporal." + k, s)) for k, s in shapes.items()}
    x = rnd("ops/att_temporal/x", B, P, Cc, T)                       # reference layout (B, D = pixels, C, T)
    temb_b = torch.from_numpy(g["att_temb"]).view(B, T, ted)[:, 0].contiguous()
    fi = torch.from_numpy(g["rpe_fi"])
    mask = torch.tensor([[1., 1, 1, 0, 0], [1, 0, 1, 1, 0]])
    d = {k: v.cuda() for k, v in sd.items()}
    xc = x.permute(0, 3, 1, 2).reshape(B * T, P, Cc).contiguous().cuda()
    xn = torch.empty_like(xc)
    nat.gn_temporal(xc, d["p.norm.weight"], d["p.norm.bias"], 1e-5, xn, B, T, P, Cc)
    M = B * T * P
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.qkv.weight"], bias=d["p.qkv.bias"],
                   Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    jobs, Rs, keep, tile0 = [], {}, [], 0
    for r in ("rpe_q", "rpe_k", "rpe_v"):
        pre = f"p.{r}.rpe_net."
        tproj = F.linear(temb_b, sd[pre + "embed_diffusion_time.weight"], sd[pre + "embed_diffusion_time.bias"]).cuda().contiguous()
        Rs[r] = torch.empty(B, T, T, Cc, device="cuda")
        keep.append(tproj)
        jobs.append(nat.RpeJob(tproj.data_ptr(), d[pre + "embed_distances.weight"].data_ptr(), d[pre + "embed_distances.bias"].data_ptr(),
                               d[pre + "out.weight"].data_ptr(), d[pre + "out.bias"].data_ptr(), Rs[r].data_ptr(), Cc, tile0, Cc, 0, None))
        tile0 += (B * T * T + 31) // 32
    nat.rpe_nets(nat.jobs_to_device(jobs, "cuda"), 3, tile0, fi.cuda(), B, T)
    o = torch.empty(M, Cc, device="cuda")
    attn = torch.empty(B * P, heads, T, T, device="cuda")
    nat.attn_temporal(qkv, Rs["rpe_q"], Rs["rpe_k"], Rs["rpe_v"], mask.cuda(), o, attn, B, T, P, Cc, heads)
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xn.view(M, Cc), ldr=Cc, out=y, ldo=Cc)
    close(y.view(B, T, P, Cc).permute(0, 2, 3, 1), torch.from_numpy(g["att_temporal_y"]), 1e-4)
    close(attn.view(B, P, heads, T, T), torch.from_numpy(g["att_temporal_attn"]).view(B, P, heads, T, T), 2e-5)


def test_spatial_attention_ops_match_the_reference_fixture(nat):
    """GroupNorm + qkv GEMM + spatial core + proj GEMM == reference RPEAttention._forward (spatial instance: no RPE, no
    mask; rpe.py:133-174): output AND attention probabilities."""
    g = _ops_golden()
    B, D, Cc, P, heads = 2, 6, 64, 5, 4                    # reference x: (B, D = frames, C, T = tokens)
    shapes = {"qkv.weight": (3 * Cc, Cc), "qkv.bias": (3 * Cc,), "proj_out.weight": (Cc, Cc), "proj_out.bias": (Cc,),
              "norm.weight": (Cc,), "norm.bias": (Cc,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("ops/att_spatial." + k, s)) for k, s in shapes.items()}
    x = rnd("ops/att_spatial/x", B, D, Cc, P)
    d = {k: v.cuda() for k, v in sd.items()}
    N, M = B * D, B * D * P
    xc = x.reshape(N, Cc, P).permute(0, 2, 1).contiguous().cuda()          # [N][P][C]
    xn = torch.empty(M, Cc, device="cuda")
    nat.check(nat.lib().lfvdm_gn_apply(nat.ptr(xc), None, Cc, 0, N, P, nat.ptr(d["p.norm.weight"]), nat.ptr(d["p.norm.bias"]),
                                       None, 1, 0, 1e-5, nat.ACT_NONE, nat.ptr(xn), None, None, None, nat.stream()), "lfvdm_gn_apply")
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.qkv.weight"], bias=d["p.qkv.bias"],
                   Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    o = torch.empty(M, Cc, device="cuda")
    attn = torch.empty(N, heads, P, P, device="cuda")
    nat.attn_spatial(qkv, o, attn, N, P, Cc, heads)
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xn, ldr=Cc, out=y, ldo=Cc)
    close(y.view(B, D, P, Cc).permute(0, 1, 3, 2), torch.from_numpy(g["att_spatial_y"]), 1e-4)
    close(attn.view(B, D, heads, P, P), torch.from_numpy(g["att_spatial_attn"]).view(B, D, heads, P, P), 2e-5)


def test_resampling_convs_match_the_reference_fixture(nat):
    """Downsample (3x3 stride 2) and Upsample (nearest x2 + 3x3) == the reference modules (unet.py:60-114)."""
    g = _ops_golden()
    N, Cc, H = 2, 32, 8
    for kind, key in (("down", "op"), ("up", "conv")):
        w = torch.from_numpy(recipe.fill_param(f"ops/{kind}.{key}.weight", (Cc, Cc, 3, 3)))
        b = torch.from_numpy(recipe.fill_param(f"ops/{kind}.{key}.bias", (Cc,)))
        x = rnd(f"ops/{kind}/x", N, Cc, H, H)
        Ho = H // 2 if kind == "down" else 2 * H
        out = torch.empty(N * Ho * Ho, Cc, device="cuda")
        nat.conv_igemm(src0=cl(x), C0=Cc, N=N, Hs=H, Ws=H, up=0 if kind == "down" else 1, stride=2 if kind == "down" else 1,
                       Ho=Ho, Wo=Ho, W=packed(nat, w), bias=b.cuda(), Cout=Cc, out=out, ldo=Cc)
        close(from_cl(out, N, Ho, Ho, Cc), torch.from_numpy(g[f"{kind}_y"]), 2e-5)


@pytest.mark.parametrize("N,P,Cc,heads", [(3, 256, 64, 4), (5, 64, 128, 4), (9, 4, 128, 4), (2, 36, 64, 4), (1, 9, 128, 2), (10, 100, 64, 2),
                                          (2, 64, 64, 1)])
def test_spatial_attention_with_fused_projection(nat, N, P, Cc, heads):
    """lfvdm_attn_spatial_fused (qkv projection + flash attention in one launch) == GroupNorm + qkv Linear + attention +
    proj of the oracle's spatial RPEAttention (rpe.py:133-174), and == the two-launch form within fp32 re-association."""
    L = nat.lib()
    assert L.lfvdm_attn_spatial_fused_ok(N, P, Cc, heads) == 0
    shapes = {"qkv.weight": (3 * Cc, Cc), "qkv.bias": (3 * Cc,), "proj_out.weight": (Cc, Cc), "proj_out.bias": (Cc,),
              "norm.weight": (Cc,), "norm.bias": (Cc,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("saf." + k, s)) for k, s in shapes.items()}
    x = rnd("saf/x", 1, N, Cc, P)                                    # (B = 1, D = N frames, C, tokens)
    ref, _ = uo.rpe_attention(sd, "p", x, None, None, None, heads, False)
    d = {k: v.cuda() for k, v in sd.items()}
    M = N * P
    xc = x[0].permute(0, 2, 1).contiguous().cuda()                  # [N][P][C]
    xn = torch.empty(M, Cc, device="cuda")
    nat.check(L.lfvdm_gn_apply(nat.ptr(xc), None, Cc, 0, N, P, nat.ptr(d["p.norm.weight"]), nat.ptr(d["p.norm.bias"]),
                               None, 1, 0, 1e-5, nat.ACT_NONE, nat.ptr(xn), None, None, None, nat.stream()), "lfvdm_gn_apply")
    o = torch.full((M, Cc), float("nan"), device="cuda")
    nat.check(L.lfvdm_attn_spatial_fused(nat.ptr(xn), nat.ptr(d["p.qkv.weight"]), nat.ptr(d["p.qkv.bias"]), nat.ptr(o), N, P, Cc, heads,
                                         nat.stream()), "lfvdm_attn_spatial_fused")
    # two-launch form on the same operands
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    nat.conv_igemm(src0=xn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.qkv.weight"], bias=d["p.qkv.bias"],
                   Cout=3 * Cc, out=qkv, ldo=3 * Cc)
    o2 = torch.empty(M, Cc, device="cuda")
    nat.attn_spatial(qkv, o2, None, N, P, Cc, heads)
    close(o, o2, 2e-5)
    o1 = o.clone()
    nat.check(L.lfvdm_attn_spatial_fused(nat.ptr(xn), nat.ptr(d["p.qkv.weight"]), nat.ptr(d["p.qkv.bias"]), nat.ptr(o), N, P, Cc, heads,
                                         nat.stream()), "lfvdm_attn_spatial_fused")
    assert torch.equal(o, o1), "bitwise reproducible"
    y = torch.empty(M, Cc, device="cuda")
    nat.conv_igemm(src0=o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=d["p.proj_out.weight"], bias=d["p.proj_out.bias"],
                   Cout=Cc, res=xn, ldr=Cc, out=y, ldo=Cc)
    close(y.view(N, P, Cc).permute(0, 2, 1), ref[0], 1e-4)


def test_spatial_fused_refuses_what_does_not_fit(nat):
    L = nat.lib()
    assert L.lfvdm_attn_spatial_fused_ok(40, 256, 128, 4) != 0       # 16x16 frame at 128 channels: 128 KB of tokens alone
    assert L.lfvdm_attn_spatial_fused_ok(40, 256, 96, 4) != 0        # head dim 24
    assert L.lfvdm_attn_spatial_fused_ok(40, 64, 128, 4) == 0


# ------------------------------------------------------------------------------------------- round-4 additions
@pytest.mark.parametrize("mode", ["atomics", "deterministic", "autograd"])
@pytest.mark.parametrize("N,P,C0,C1,film,act,adds", [(2, 2500, 96, 32, True, 1, 2), (4, 1061, 64, 0, False, 1, 0),
                                                     (2, 4096, 128, 128, True, 1, 1), (2, 16384, 128, 0, False, 1, 2),
                                                     (2, 300, 512, 256, False, 0, 1), (4, 1024, 256, 0, True, 1, 0),
                                                     (2, 256, 384, 0, False, 1, 1), (4, 256, 64, 0, False, 1, 1)])
def test_gn_backward_large_maps(nat, N, P, C0, C1, film, act, adds, mode, monkeypatch):
    """lfvdm_gn_bwd_ws: the chunked two-launch GroupNorm(+FiLM)(+SiLU) backward for maps whose (sample, 8 groups) slice
    exceeds one workgroup's registers (pixel space, wide concats) - ragged last chunks, concat split, one / two extra
    gradients folded into dx - in the three gradient delivery modes, against fp64 autograd of
    silu(group_norm(x) * (1 + scale) + shift) (reference nn.py:17-19, unet.py:199-203).  The last case fits one
    workgroup and must take the single-launch kernels (workspace size 0)."""
    from improved_diffusion import _backward as bw
    monkeypatch.setenv("LFVDM_DETERMINISTIC", "1" if mode == "deterministic" else "0")
    C, T = C0 + C1, 2
    need = int(nat.lib().lfvdm_gn_bwd_ws_floats(C, N, P))
    assert (need == 0) == (P == 256 and C == 64)
    a = (rnd("gbl/a", N * P, C0) * 1.3 + 0.7)
    b = rnd("gbl/b", N * P, C1) if C1 else None
    gamma, beta = 1 + 0.1 * rnd("gbl/g", C), 0.1 * rnd("gbl/be", C)
    fm = 0.3 * rnd("gbl/film", N // T, 2 * C) if film else None
    da = rnd("gbl/da", N * P, C)
    extra = [rnd(f"gbl/add{i}", N * P, C) for i in range(adds)]
    # fp64 reference
    xd = torch.cat([a] + ([b] if C1 else []), dim=1).double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    fd = fm.double().requires_grad_(True) if film else None
    y = F.group_norm(xd.view(N, P, C).permute(0, 2, 1), 32, gd, bd, eps=1e-5)
    if film:
        f = fd.repeat_interleave(T, dim=0)
        y = y * (1 + f[:, :C, None]) + f[:, C:, None]
    if act:
        y = F.silu(y)
    (y.permute(0, 2, 1).reshape(N * P, C) * da.double()).sum().backward()
    dx_ref = xd.grad + sum(e.double() for e in extra) if extra else xd.grad
    # product path
    gpar = torch.nn.Parameter(gamma.cuda()); bpar = torch.nn.Parameter(beta.cuda())
    ac, bc = a.cuda(), (b.cuda() if C1 else None)
    fc = fm.cuda() if film else None
    _, cA, cB, st = bw._gn_apply(ac, bc, C0, C1, N, P, gpar.detach(), bpar.detach(), fc, T, act)
    ex = [e.cuda() for e in extra]
    kw = dict(add=ex[0] if adds >= 1 else None, add2=ex[1] if adds >= 2 else None)
    runs = []
    for _ in range(2):
        gpar.grad = None; bpar.grad = None
        dxa, dxb, dg, db, dfilm = bw._gn_backward(da.cuda(), ac, bc, C0, C1, N, P, cA, cB, st, act, gpar, bpar, fc, T,
                                                  inplace=(mode != "autograd"), **kw)
        if mode != "autograd":
            dg, db = gpar.grad, bpar.grad
        torch.cuda.synchronize()
        runs.append((dxa.clone(), None if dxb is None else dxb.clone(), dg.clone(), db.clone(), None if dfilm is None else dfilm.clone()))
    dxa, dxb, dg, db, dfilm = runs[0]
    dx = torch.cat([dxa] + ([dxb] if C1 else []), dim=1)
    scale = float(dx_ref.abs().max())
    close(dx, dx_ref.float(), 3e-5 * max(scale, 1.0))
    close(dg, gd.grad.float(), 2e-4 * float(gd.grad.abs().max()))
    close(db, bd.grad.float(), 2e-4 * float(bd.grad.abs().max()))
    if film:
        close(dfilm, fd.grad.float(), 2e-4 * float(fd.grad.abs().max()))
    # dx is deterministic in every mode; the parameter gradients are when no float atomics are involved
    assert torch.equal(runs[0][0], runs[1][0]) and (dxb is None or torch.equal(runs[0][1], runs[1][1]))
    if mode != "atomics":
        assert torch.equal(runs[0][2], runs[1][2]) and torch.equal(runs[0][3], runs[1][3])


@pytest.mark.parametrize("N,C0,C1,Cout,H,k", [(10, 128, 0, 128, 16, 3), (3, 128, 128, 256, 8, 3), (4, 256, 0, 128, 8, 1),
                                              (2, 128, 0, 160, 12, 3), (1, 64, 64, 96, 32, 3), (2, 128, 0, 128, 64, 3)])
def test_conv_wgrad_every_tune_code(nat, N, C0, C1, Cout, H, k):
    """Every launch code the weight-gradient tuner may pick for a layer shape (_native._wgrad_codes: 64- / 128-filter tiles,
    the 128 x 128 tile of code 3, two / three LDS-DMA stages, 64-row chunks, M slices, and - 3x3 layers on maps of 8, 16 or a
    multiple of 32 pixels per row - the tap-fused kernels with three / nine taps per workgroup) gives the same dW and db as
    torch autograd; the tap-fused kernels also through the deterministic slabs, bitwise reproducibly."""
    import ctypes as C
    Cin = C0 + C1
    x = rnd("wgc/x", N, Cin, H, H)
    w = (rnd("wgc/w", Cout, Cin, k, k, scale=0.05)).requires_grad_(True)
    b = rnd("wgc/b", Cout).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1 if k == 3 else 0)
    dout = rnd("wgc/d", N, Cout, H, H)
    y.backward(dout)
    gp = torch.zeros(Cout, k * k, Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    keep = dict(src0=cl(x[:, :C0]), res=cl(dout))
    if C1:
        keep["src1"] = cl(x[:, C0:])
    a = nat.fill_conv_args(C0=C0, C1=C1, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, ldr=Cout, out=gp, bias=db, Cout=Cout, **keep)
    codes = nat._wgrad_codes(a)
    tiles = {(c - 1) & 3 for c in codes if ((c - 1) >> 2) & 3}
    assert ({1, 2, 3} if Cin % 128 == 0 and C0 % 128 == 0 and Cout >= 128 else {1}) <= tiles, tiles
    taps_codes = [c for c in codes if ((c - 1) >> 2) & 3 == 0]
    eligible = k == 3 and (H * H) % 32 == 0 and (H % 32 == 0 or H in (8, 16))
    assert bool(taps_codes) == eligible and (not eligible or {(c - 1) & 3 for c in taps_codes} == {1, 2})
    scale = max(1.0, float(w.grad.abs().max()))
    for code in [0] + codes:
        a.tune = code
        gp.zero_(); db.zero_()
        nat.check(nat.lib().lfvdm_conv_wgrad(C.byref(a), nat.stream()), "lfvdm_conv_wgrad")
        got_w = gp.view(Cout, k, k, Cin).permute(0, 3, 1, 2).cpu()
        err = float((got_w - w.grad).abs().max())
        assert err < 2e-5 * scale, f"tune code {code}: dW max|d| = {err:.3e}"
        assert float((db.cpu() - b.grad).abs().max()) < 2e-5 * max(1.0, float(b.grad.abs().max())), f"tune code {code}: db"
    # deterministic mode (partial tiles stored to slab rows, summed in slice order) for the tap-fused kernels
    ws = torch.empty(64 << 20, device="cuda")
    a.splitk_ws, a.splitk_ws_floats = ws.data_ptr(), ws.numel()
    for code in taps_codes[:2] + taps_codes[-2:]:
        a.tune = code
        runs = []
        for _ in range(2):
            gp.zero_(); db.zero_()
            nat.check(nat.lib().lfvdm_conv_wgrad(C.byref(a), nat.stream()), "lfvdm_conv_wgrad")
            runs.append((gp.clone(), db.clone()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), f"tune code {code}: not reproducible"
        got_w = gp.view(Cout, k, k, Cin).permute(0, 3, 1, 2).cpu()
        assert float((got_w - w.grad).abs().max()) < 2e-5 * scale, f"tune code {code} (deterministic)"
        assert float((db.cpu() - b.grad).abs().max()) < 2e-5 * max(1.0, float(b.grad.abs().max())), f"tune code {code}: db (deterministic)"


def test_tensors_above_one_gibibyte_take_the_range_paths(nat):
    """Operands of 2^30 bytes and more (pixel space at large batch: 64 frames x 128 x 128 x 256 channels = 1 GiB): the LDS-DMA
    kernels address with 32-bit byte offsets below 2^30, so lfvdm_conv_igemm cuts the batch into sample ranges and
    lfvdm_conv_wgrad falls back to the register-staged kernels with 64-bit addressing.  Reference: torch's own convolution
    and its autograd ON THE GPU (fp32) - the CPU would need minutes at this size.  Also the full pixel-space weight-gradient
    shape of BASELINE.json configs[4] (20 x 128 x 128, 128 -> 128: M = 327680) through the tap-fused and tiled kernels."""
    import ctypes as C
    torch.backends.cudnn.allow_tf32 = False
    g = torch.Generator(device="cuda").manual_seed(5)
    # ---- forward + weight gradient above 2^30 bytes
    N, H, Cin, Cout = 64, 128, 256, 64
    x = torch.randn(N, H, H, Cin, device="cuda", generator=g)                   # channels-last rows: exactly 2^30 bytes
    assert x.numel() * 4 >= 1 << 30
    w = (0.02 * torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g)).requires_grad_(True)
    b = torch.randn(Cout, device="cuda", generator=g).requires_grad_(True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)                      # (N, Cout, H, W) view of channels-last data
    out = torch.empty(N * H * H, Cout, device="cuda")
    wp = packed(nat, w.detach().cpu())
    nat.conv_igemm(src0=x.view(-1, Cin), C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, W=wp, bias=b.detach(), Cout=Cout, out=out, ldo=Cout)
    got = out.view(N, H, H, Cout).permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    err = float((got - ref.detach()).abs().max())
    assert err < 2e-4 * scale, (err, scale)
    dout = torch.randn(N * H * H, Cout, device="cuda", generator=g)
    ref.backward(dout.view(N, H, H, Cout).permute(0, 3, 1, 2))
    gp = torch.zeros(Cout, 9, Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    nat.conv_wgrad(src0=x.view(-1, Cin), C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=3, res=dout, ldr=Cout, out=gp, bias=db, Cout=Cout)
    got_w = gp.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    sw = float(w.grad.abs().max())
    assert float((got_w - w.grad).abs().max()) < 5e-4 * sw, (float((got_w - w.grad).abs().max()), sw)
    assert float((db - b.grad).abs().max()) < 5e-4 * float(b.grad.abs().max())
    del x, ref, out, got, dout
    torch.cuda.empty_cache()
    # ---- the full pixel-space weight-gradient shape, every kernel family the tuner may pick for it
    N, Cin, Cout = 20, 128, 128
    x = torch.randn(N, H, H, Cin, device="cuda", generator=g)
    w = (0.02 * torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g)).requires_grad_(True)
    dout = torch.randn(N * H * H, Cout, device="cuda", generator=g)
    F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1).backward(dout.view(N, H, H, Cout).permute(0, 3, 1, 2))
    sw = float(w.grad.abs().max())
    gp = torch.zeros(Cout, 9, Cin, device="cuda")
    a = nat.fill_conv_args(src0=x.view(-1, Cin), C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=3, res=dout, ldr=Cout, out=gp, Cout=Cout)
    codes = nat._wgrad_codes(a)
    pick = {}
    for c in codes:                                     # one code per (tile, stage field): largest M-slice count of each
        pick[((c - 1) & 3, ((c - 1) >> 2) & 3)] = c
    assert {(1, 0), (2, 0), (3, 1)} <= set(pick), sorted(pick)
    for key, code in sorted(pick.items()):
        a.tune = code
        gp.zero_()
        nat.check(nat.lib().lfvdm_conv_wgrad(C.byref(a), nat.stream()), "lfvdm_conv_wgrad")
        err = float((gp.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2) - w.grad).abs().max())
        assert err < 5e-4 * sw, (key, code, err, sw)


def test_conv_wgrad_oihw_output_every_tune_code(nat):
    """out_mode = 1 (dW written in the parameter's own OIHW layout, include/lfvdm_hip.h): the tap-fused kernels only know the
    packed accumulator layout, so the tuner must not offer them here and a tap code set by hand must run another kernel -
    every code gives torch's dW directly in OIHW."""
    import ctypes as C
    N, Cin, Cout, H, k = 4, 128, 128, 16, 3
    x = rnd("wgo/x", N, Cin, H, H)
    w = (rnd("wgo/w", Cout, Cin, k, k, scale=0.05)).requires_grad_(True)
    b = rnd("wgo/b", Cout).requires_grad_(True)
    F.conv2d(x, w, b, padding=1).backward(rnd("wgo/d", N, Cout, H, H))
    dout = rnd("wgo/d", N, Cout, H, H)
    g = torch.zeros(Cout, Cin, k, k, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    keep = dict(src0=cl(x), res=cl(dout))
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, ldr=Cout, out=g, bias=db, Cout=Cout, out_mode=1, **keep)
    packed_codes = nat._wgrad_codes(nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, ldr=Cout, out=g, bias=db,
                                                       Cout=Cout, **keep))
    taps = [c for c in packed_codes if ((c - 1) >> 2) & 3 == 0]
    codes = nat._wgrad_codes(a)
    assert taps and not [c for c in codes if ((c - 1) >> 2) & 3 == 0], "tap-fused codes are offered for the packed layout only"
    scale = max(1.0, float(w.grad.abs().max()))
    for code in [0] + codes + taps[:2]:
        a.tune = code
        g.zero_(); db.zero_()
        nat.check(nat.lib().lfvdm_conv_wgrad(C.byref(a), nat.stream()), "lfvdm_conv_wgrad")
        err = float((g.cpu() - w.grad).abs().max())
        assert err < 2e-5 * scale, f"tune code {code}: OIHW dW max|d| = {err:.3e}"
        assert float((db.cpu() - b.grad).abs().max()) < 2e-5 * max(1.0, float(b.grad.abs().max())), f"tune code {code}: db"


@pytest.mark.parametrize("N,Cin,C0,C1,H", [(40, 128, 128, 128, 2), (8, 256, 128, 128, 4), (6, 64, 64, 64, 4), (4, 128, 256, 256, 2)])
def test_concat_groupnorm_half_by_half_every_tune_code(nat, N, Cin, C0, C1, H):
    """The decoder's first normalisation, GroupNorm32 + SiLU over concat(h, skip) (unet.py:460 + :152-155), evaluated half by
    half: h = conv3x3(x) normalised in the producing GEMM's epilogue with the CONCAT's group width (lfvdm_conv_args.gn_gw,
    written into the left columns of the [M][C0 + C1] operand: gn_ld), the skip half by lfvdm_gn_apply_part.  No group
    straddles the concat when (C0 + C1) / 32 divides both halves, so the result must equal F.group_norm of the materialised
    concat in fp64 - for every tile code the tuner may pick, in both forms of the epilogue."""
    import ctypes as C
    x, w, b = rnd("cat/x", N, Cin, H, H), rnd("cat/w", C0, Cin, 3, 3, scale=0.05), rnd("cat/b", C0)
    skip = rnd("cat/skip", N, C1, H, H)
    Cc = C0 + C1
    gw = Cc // 32
    gamma, beta = 1 + 0.1 * rnd("cat/g", Cc), 0.1 * rnd("cat/be", Cc)
    h = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.silu(F.group_norm(torch.cat([h, skip.double()], 1), 32, gamma.double(), beta.double(), eps=1e-5)).float()
    M, P = N * H * H, H * H
    act = torch.full((M, Cc), float("nan"), device="cuda")
    raw = torch.empty(M, C0, device="cuda")
    g_dev, b_dev = gamma.cuda(), beta.cuda()
    sk = cl(skip)
    nat.check(nat.lib().lfvdm_gn_apply_part(sk.data_ptr(), C1, N, P, gw, g_dev.data_ptr() + 4 * C0, b_dev.data_ptr() + 4 * C0, 1e-5,
                                            nat.ACT_SILU, act.data_ptr() + 4 * C0, Cc, nat.stream()), "lfvdm_gn_apply_part")
    ws = torch.empty(1 << 22, device="cuda")
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    keep = dict(src0=cl(x), W=packed(nat, w), bias=b.cuda())
    a = nat.fill_conv_args(C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, Cout=C0, out=raw, ldo=C0, gn_out=act, gn_gamma=g_dev, gn_beta=b_dev,
                           gn_film_div=1, gn_act=nat.ACT_SILU, gn_skip_raw=0, **keep)
    a.gn_gw, a.gn_ld = gw, Cc
    a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
    codes = (C.c_int * 256)()
    n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
    assert n > 0
    right = act[:, C0:].clone()
    for code, general in [(c, g) for c in [0] + [codes[i] for i in range(n)] for g in (0, 1)]:
        a.tune, a.gn_general = code, general
        act[:, :C0].fill_(float("nan"))
        nat.conv_igemm_struct(a)
        assert torch.equal(act[:, C0:], right), "the GEMM's epilogue must not touch the skip half's columns"
        err = float((from_cl(act, N, H, H, Cc).cpu() - ref).abs().max())
        assert err < 1e-4, f"tune code {code}, general form {general}: max|d| = {err:.3e}"
        assert float((from_cl(raw, N, H, H, C0).cpu() - h.float()).abs().max()) < 5e-5
