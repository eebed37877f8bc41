/*
 * lfvdm_hip.h — C ABI of the MI355X (gfx950) native layer under the `improved_diffusion`
 * hot path (frame-conditioned video U-Net forward/backward + Gaussian-diffusion step math).
 *
 * The reference (plai-group/latent-flexible-video-diffusion-modeling) has NO native layer:
 * its boundary is the Python module surface.  Every entry point below
 * therefore replaces a stock ATen op sequence at the cited reference site and is what a
 * ctypes / cffi stub in the reference's Python would bind.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless the name says otherwise
 *     (`i64` = int64_t, `i32` = int32_t); nothing is allocated or freed here;
 *   - every launcher is stream-ordered on `stream` (a hipStream_t passed as void*), is
 *     re-entrant and hipGraph-capturable (no sync, no malloc).  The library keeps no data
 *     state; the only host-side state is a per-(kernel instance, device) record of the
 *     dynamic-LDS limit that has been raised with hipFuncSetAttribute (atomic fast path,
 *     mutex-guarded slow path: safe from several host threads and several devices), and a
 *     handful of LFVDM_CONV_* / LFVDM_WGRAD_* environment variables - A/B and tuning aids
 *     for developers, read at most once per process (thread-safe static initialisation),
 *     never required, and documented where they are read;
 *   - return value: 0 = launched, non-zero = LFVDM_E_* (invalid shape / unsupported config /
 *     launch error); the Python shim turns non-zero into RuntimeError;
 *   - activations are channels-last: [N][H][W][C] with N = B*T, n = b*T + t
 *     (the reference's (B*T, C, H, W) is converted once at the model input and once at the
 *     output: lfvdm_conv_in / LFVDM_OUT_NCHW).
 */
#ifndef LFVDM_HIP_H
#define LFVDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFVDM_OK 0
#define LFVDM_E_SHAPE 1   /* operand shapes violate the kernel's assumptions */
#define LFVDM_E_LAUNCH 2  /* hipGetLastError() != hipSuccess after the launch */
#define LFVDM_E_UNSUPPORTED 3

#define LFVDM_ACT_NONE 0
#define LFVDM_ACT_SILU 1

#define LFVDM_OUT_ROWS 0 /* out[m*ldo + co]                        (channels-last rows) */
#define LFVDM_OUT_NCHW 1 /* out[(n*Cout + co)*Ho*Wo + pix]          (reference frame layout) */

int lfvdm_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d 3x3 (unet.py:76,108,155,169,313,402), the 1x1 skip conv (unet.py:180),
 * nn.Linear qkv / proj_out (rpe.py:111-112,139,171) and - fused into the launch - the nearest-2x
 * Upsample (unet.py:85-87), the skip concat (unet.py:460), the residual add (unet.py:207,
 * rpe.py:172) and, in the epilogue, the NEXT GroupNorm32 (+FiLM)(+SiLU) (gn_* fields).
 *
 *   out[m][co] = bias[co] + sum_{tap,ci} src[n, iy, ix, ci] * W[co][tap*Cin + ci]     (zero outside the padded image)
 *              (+ bias2[co] + sum_ci src2[m][ci] * W2[co][ci])            second segment
 *              (+ res[m][co] * resA[n][co] + resB[n][co])                 residual
 * Operands are RAW: coefA / coefB / act (the GroupNorm affine + SiLU applied while staging, rounds 1-2) must be
 * NULL / 0 for lfvdm_conv_igemm (LFVDM_E_UNSUPPORTED otherwise) - normalise with lfvdm_gn_apply or a producer's gn_*
 * epilogue.  lfvdm_conv_wgrad still honours them (f(v) = act(v * coefA[n][ci] + coefB[n][ci])).
 * ------------------------------------------------------------------------------------- */
typedef struct lfvdm_conv_args {
    /* main segment: virtual concat of src0 (C0 ch) and src1 (C1 ch, may be 0/NULL) */
    const float* src0;
    const float* src1;
    int32_t C0, C1;
    int32_t N;        /* samples (B*T) */
    int32_t Hs, Ws;   /* spatial size of the stored source */
    int32_t up;       /* 1: source is nearest-upsampled x2 on the fly; 2: zero-insertion x2 (the data
                       * gradient of a stride-2 conv is this conv on the transposed-flipped weights) */
    int32_t stride;   /* 1 or 2 */
    int32_t ksize;    /* 3 (pad 1) or 1 (pad 0) */
    int32_t Ho, Wo;   /* output spatial size */
    const float* coefA; /* [N][C0+C1] or NULL (identity) */
    const float* coefB;
    int32_t act;      /* LFVDM_ACT_* applied after the affine */
    const float* W;   /* packed [Cout][ksize*ksize][C0+C1] (see lfvdm_pack_conv_weight) */
    const float* bias;/* [Cout] or NULL */
    int32_t Cout;
    /* optional second segment: 1x1 on a raw (un-normalised) concat source at output res */
    const float* s2src0;
    const float* s2src1;
    int32_t s2C0, s2C1;
    const float* W2;  /* [Cout][s2C0+s2C1] */
    const float* bias2;
    /* optional residual, rows [M][ldr]; resA/resB [N][Cout] or NULL (plain add) */
    const float* res;
    int32_t ldr;
    const float* resA;
    const float* resB;
    /* output */
    float* out;
    int32_t ldo;      /* row stride for LFVDM_OUT_ROWS */
    int32_t out_mode; /* LFVDM_OUT_* */
    int32_t tune;     /* 0: built-in makespan model picks tile shape / K-chunk (never split-K; LDS-DMA staging
                       * whenever the operands are raw); otherwise a code returned by lfvdm_conv_igemm_candidates
                       * (set by an autotuner for a fixed shape).  Codes are opaque to callers and tied to
                       * lfvdm_abi_version(): 1 + tile id + 16*(64-channel chunks) + 32*log2(split-K) +
                       * 256*(0 register staging | 1 LDS-DMA 2 stages | 2 LDS-DMA 3 stages) */
    /* optional workspace enabling deterministic split-K over workgroups for small-M layers: slabs of partial
     * tiles + one arrival ticket per output tile.  The tickets must be ZERO before the first launch; every
     * launch leaves them zero again (the last slice to arrive resets its tile's ticket) */
    float* splitk_ws;
    int32_t* splitk_cnt;
    int64_t splitk_ws_floats;
    int64_t splitk_cnt_ints;
    /* optional fused GroupNorm(32 groups, nn.py:17-19) + FiLM (unet.py:199-203) + activation of the OUTPUT - the
     * normalisation that sits in front of the NEXT layer - evaluated in this launch's epilogue (LFVDM_OUT_ROWS only):
     *   gn_out[m][co] = act((out - mean) * rstd * gamma[co] + beta[co]) * (1 + film[n / gn_film_div][co]) + film[..][Cout + co])
     * with mean / rstd over the Ho*Wo rows x Cout/32 channels of (sample n, group).  Needs Cout % 32 == 0 and a tile
     * that holds whole samples and whole groups (Ho*Wo divides the tile's rows; lfvdm_conv_igemm_candidates only
     * offers such tiles, and the launch returns LFVDM_E_UNSUPPORTED if none exists).  gn_skip_raw != 0: `out`
     * itself is not written (nobody else reads the raw tensor). */
    const float* gn_gamma;
    const float* gn_beta;
    const float* gn_film;   /* [N / gn_film_div][gn_film_ld] (scale | shift) or NULL */
    float* gn_out;          /* [M][Cout]; NULL = no fused normalisation */
    int32_t gn_film_ld;
    int32_t gn_film_div;
    int32_t gn_act;         /* LFVDM_ACT_* */
    int32_t gn_skip_raw;
    float gn_eps;
    int32_t gn_general;     /* non-zero: the general (LDS tile) form of the fused GroupNorm even where the register form
                             * applies (P and Cout/32 powers of two) - A/B and test aid, same statistics */
    /* the fused normalisation as ONE HALF of a GroupNorm over a channel concat (the first normalisation of a decoder
     * ResBlock, unet.py:460 + :152-155: 32 groups over C0 + C1 channels, whose group boundaries never straddle the concat
     * when (C0 + C1) / 32 divides C0): gn_gw = channels per group (0: Cout / 32), gn_ld = row stride of gn_out (0: Cout; the
     * concat's width when gn_out is the left part of the consumer's operand - the other half comes from lfvdm_gn_apply_part);
     * gn_gamma / gn_beta point at this half's first channel */
    int32_t gn_gw;
    int32_t gn_ld;
} lfvdm_conv_args;

int lfvdm_conv_igemm(const lfvdm_conv_args* a, void* stream);
/* template instance (NT = 32-column tiles per wave, nwaves = K-split waves per workgroup) that
 * lfvdm_conv_igemm picks for these arguments; profiling aid, launches nothing. */
int lfvdm_conv_igemm_config(const lfvdm_conv_args* a, int* nt, int* nwaves);
/* legal `tune` codes for these arguments (tile configuration x K-chunk width x split-K factor x operand staging);
 * up to ~100 codes: pass max_codes >= 256 */
int lfvdm_conv_igemm_candidates(const lfvdm_conv_args* a, int* codes, int max_codes);

/* Several small weight-gradient launches as ONE (the wave-private kernel: a wave owns a 32 x 32 tile of dW for a
 * slice of M): job i covers wave tasks [task0, task0 + ntasks), ntasks = taps * (Cin/32) * ceil(Cout/32) * msplit;
 * jobs sorted by task0.  Same argument meaning as lfvdm_conv_wgrad. */
typedef struct lfvdm_wgrad_job {
    lfvdm_conv_args a;
    int32_t msplit;
    int32_t task0;
} lfvdm_wgrad_job;
int lfvdm_conv_wgrad_grouped(const lfvdm_wgrad_job* jobs_dev, int njobs, int total_tasks, void* stream);

/* OIHW [Cout][Cin][k][k] -> [Cout][k*k][Cin] (k in {1,3}); state-dict layout stays OIHW. */
int lfvdm_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int ksize, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward of the convolution / linear (autograd of nn.Conv2d / nn.Linear in the reference,
 * train_util.py:328 `loss.backward()`).
 *   data gradient   : lfvdm_conv_igemm on dout with the weights packed by lfvdm_pack_conv_weight_t
 *                     (Wt[ci][tap][co] = W[co][ci][flip(tap)]); stride-2 convs use up = 2.
 *   weight gradient : lfvdm_conv_wgrad.  `a` describes the FORWARD operand (src*, C*, N, Hs, Ws, up,
 *                     stride, ksize, Ho, Wo, coefA/B, act); a->res = dout rows [M][ldr]; a->out = packed
 *                     gradient [Cout][k*k][Cin] (a->out_mode = 0) or the OIHW parameter gradient itself
 *                     (a->out_mode = 1), ACCUMULATED with float atomics; a->bias = bias gradient [Cout]
 *                     (accumulated) or NULL.
 *   lfvdm_unpack_conv_grad: packed [Cout][k*k][Cin] -> OIHW gradient (accumulate = 1: +=).
 * ------------------------------------------------------------------------------------- */
int lfvdm_conv_wgrad(const lfvdm_conv_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * Deterministic gradients (LFVDM_DETERMINISTIC=1 on the Python side; reference behaviour: loss.backward() on the CPU
 * path is run-to-run reproducible, train_util.py:328).  By default the partial sums of different workgroups - weight /
 * bias gradients over slices of the output pixels, GroupNorm parameter gradients over samples, the input gradient of
 * the embedding projections over output rows, the RPE hidden-layer gradients over row tiles - are combined with float
 * atomics: the order, hence the rounding, changes from run to run.  With a workspace the partials are STORED to slabs
 * (one row per contributor, zero-filled first where a contributor does not cover every element) and added in
 * contributor order by an extra launch: bitwise reproducible, at the price of the slab traffic.
 *   lfvdm_conv_wgrad     : a->splitk_ws = slab, a->splitk_ws_floats = its capacity (>= Cout*k*k*Cin + Cout; the number
 *                          of pixel slices is reduced to what fits)
 *   lfvdm_*_det          : the entry points below with (det_ws, det_ws_floats) appended
 *   lfvdm_det_reduce     : dst[i] += slab[0][i] + slab[1][i] + ... + slab[parts-1][i], i < n  (the ordered sum itself)
 * One workspace per device serves every launch (they are stream-ordered).
 * ------------------------------------------------------------------------------------- */
int lfvdm_det_reduce(float* dst, const float* slab, long n, long parts, void* stream);
int lfvdm_pack_conv_weight_t(const float* w_oihw, float* w_packed_t, int Cout, int Cin, int ksize, void* stream);
int lfvdm_unpack_conv_grad(const float* g_packed, float* g_oihw, int Cout, int Cin, int ksize, int accumulate, void* stream);

/* Grouped weight packing (one launch per training step instead of two per convolution): job = one OIHW weight ->
 * [Cout][k*k][Cin] (transposed = 0, as lfvdm_pack_conv_weight) or [Cin][k*k][Cout] flipped (transposed = 1, as
 * lfvdm_pack_conv_weight_t).  blk0 = first workgroup of the job (one workgroup per tile of 32 filters x 32 input
 * channels: ceil(Cout/32)*ceil(Cin/32) per job), jobs sorted by blk0; k*k <= 9. */
typedef struct lfvdm_pack_job {
    const float* src;
    float* dst;
    int32_t Cout, Cin, taps, transposed, blk0;
    int32_t ld;      /* stride of the destination's fastest axis (0 = Cin / Cout): a larger value leaves zero-padded
                        channels, written once by the caller - the 5-channel input conv and the 4-filter output conv
                        are staged as 32-channel operands */
} lfvdm_pack_job;
int lfvdm_pack_conv_weights(const lfvdm_pack_job* jobs_dev, int njobs, int total_blocks, void* stream);

/* Grouped version for a training step: every job folds one packed gradient [Cout][k*k][Cin] into the OIHW
 * parameter gradient (g += unpack(gp)) and zeroes gp for the next step.  row0 = first workgroup (filter row) of
 * the job, jobs sorted by row0; total_rows = sum of Cout; max_row_floats = max k*k*max(Cin, ldp) (<= 16384). */
typedef struct lfvdm_unpack_job {
    float* gp;
    float* g;
    int32_t Cout, Cin, taps, row0;
    int32_t ldp;     /* channel stride of gp (0 = Cin); > Cin for accumulators of zero-padded operands */
    int32_t pad_;
} lfvdm_unpack_job;
int lfvdm_unpack_conv_grads(const lfvdm_unpack_job* jobs_dev, int njobs, int total_rows, int max_row_floats, void* stream);

/* ---------------------------------------------------------------------------------------
 * Model prologue: input compositing + indicator channel + 3x3 input conv in one kernel
 * (unet.py:441-450 and input_blocks.0, unet.py:310-316).
 *   x, x0: (B,T,C,H,W) frame layout; obs: (B*T) floats; w: OIHW [Cout][C+1][3][3]
 *   out: channels-last [B*T][H][W][Cout]
 * ------------------------------------------------------------------------------------- */
int lfvdm_conv_in(const float* x, const float* x0, const float* obs, const float* w, const float* bias,
                  float* out, int N, int C, int H, int W, int Cout, void* stream);
/* lfvdm_conv_in with the sampler's clock riding in the same launch (one extra workgroup; nothing in the first conv reads
 * the timestep): exactly lfvdm_sampler_tick_fetch's effect on (t, model_t, rows) - see there for the arguments. */
int lfvdm_conv_in_tick(const float* x, const float* x0, const float* obs, const float* w, const float* bias, float* out,
                       int N, int C, int H, int W, int Cout, int64_t* t, const float* model_timestep_table, float* model_t,
                       int B, const float* rows_all, int rows_ld, float* rows, int row_floats, void* stream);

/* ---------------------------------------------------------------------------------------
 * GroupNorm(32) statistics -> per-(sample, channel) affine coefficients, optionally folded
 * with the FiLM scale/shift of the ResBlock (nn.py:17-19,95-102; unet.py:199-203):
 *   y = x*A + B,  A = rstd*gamma*(1+scale), B = (beta - mean*rstd*gamma)*(1+scale) + shift
 * src is a virtual concat [N][P][C0+C1]; film: [Bf][2*C] rows (scale | shift) indexed by
 * n / film_div (the embedding is per batch element, n = b*T + t), or NULL.
 * ------------------------------------------------------------------------------------- */
int lfvdm_gn_coef(const float* src0, const float* src1, int C0, int C1, int N, int P,
                  const float* gamma, const float* beta, const float* film, int film_div, int film_ld,
                  float eps, float* coefA, float* coefB, void* stream);

/* GroupNorm(+FiLM)(+activation) APPLIED: out[n][p][c] = act(x * A + B) as one contiguous [N*P][C0+C1] tensor (the
 * virtual concat is materialised), nn.py:17-19 + SiLU nn.py:12-14 + unet.py:199-203.  coefA/coefB/stats are
 * optional outputs (NULL to skip).  On gfx950 fp32 MFMA and VALU share the vector ALUs, so evaluating the
 * normalisation/activation once here is cheaper than re-evaluating it per tap inside the consuming GEMM. */
int lfvdm_gn_apply(const float* src0, const float* src1, int C0, int C1, int N, int P,
                   const float* gamma, const float* beta, const float* film, int film_div, int film_ld,
                   float eps, int act, float* out, float* coefA, float* coefB, float* stats, void* stream);
/* A PART of a GroupNorm over a wider tensor: C channels of src [N][P][C] normalised in groups of `cg` channels (2, 4, 8, 16;
 * C % 16 == 0, P <= 256: the one-wave kernel), gamma / beta pointing at the part's first channel, written to
 * out[(n*P + p) * ldo + c].  With lfvdm_conv_args.gn_gw / gn_ld this evaluates the GroupNorm over a channel concat half by
 * half - each half where its tensor is produced - instead of materialising the concat first. */
int lfvdm_gn_apply_part(const float* src, int C, int N, int P, int cg, const float* gamma, const float* beta, float eps, int act,
                        float* out, int ldo, void* stream);
/* lfvdm_gn_apply for LARGE maps (pixel space, 32x32 latents): when a (sample, 8 groups) slice does not fit the registers of
 * one workgroup, the slice is cut into chunks of positions owned by separate workgroups - chunk statistics (exact two-pass
 * mean / M2 per group) into `ws`, then every workgroup combines the partials of its groups in a fixed order (Chan et al.)
 * and applies the affine to its chunk: two launches of thousands of workgroups instead of one of N*4 (0.1 vs 1-2 ms per
 * GroupNorm at 20 x 128 x 128 x 128).  ws: lfvdm_gn_apply_ws_floats(C0 + C1, N, P) floats of scratch; 0 means the slice
 * fits and the call is lfvdm_gn_apply.  Deterministic; same outputs (coefA / coefB / stats optional). */
long lfvdm_gn_apply_ws_floats(int C, int N, int P);
int lfvdm_gn_apply_ws(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                      const float* beta, const float* film, int film_div, int film_ld, float eps, int act, float* out,
                      float* coefA, float* coefB, float* stats, float* ws, long ws_floats, void* stream);

/* Same as lfvdm_gn_coef, additionally writing (mean, rstd) per (sample, group) to stats[N][32][2] for the backward. */
int lfvdm_gn_coef_stats(const float* src0, const float* src1, int C0, int C1, int N, int P,
                        const float* gamma, const float* beta, const float* film, int film_div, int film_ld,
                        float eps, float* coefA, float* coefB, float* stats, void* stream);

/* GroupNorm(+FiLM)(+SiLU) backward (autograd of nn.GroupNorm / SiLU in the reference's loss.backward()).
 * da [N*P][C0+C1] is the gradient w.r.t. act(x*A+B) (the conv data gradient); coefA/coefB/stats come from
 * lfvdm_gn_coef_stats.  _stats writes sums[N][C][2] = (sum_p dz, sum_p dz*xhat); _apply writes dx, split over
 * the two concat sources (out1 may be NULL when C1 == 0), overwriting (acc = 0) or accumulating (acc = 1). */
int lfvdm_gn_bwd_stats(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                       const float* coefA, const float* coefB, const float* stats, int act, float* sums, void* stream);
int lfvdm_gn_bwd_apply(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                       const float* coefA, const float* coefB, const float* stats, const float* sums, int act,
                       float* out0, float* out1, int acc0, int acc1, void* stream);
/* lfvdm_gn_bwd_apply + the GroupNorm(+FiLM) parameter gradients in the same launch: dgamma / dbeta [C] and dfilm
 * [N/T][2C] (when film != NULL; zero it first) are ACCUMULATED with float atomics (same sums as lfvdm_gn_param_grads,
 * summation order not fixed).  add (optional, rows [N*P][add_ld], add_ld >= C0+C1): a second gradient of the same
 * input - e.g. the skip path of a ResBlock - added to dx on the way out. */
int lfvdm_gn_bwd_apply_params(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                              const float* coefA, const float* coefB, const float* stats, const float* sums, int act,
                              float* out0, float* out1, int acc0, int acc1, const float* gamma, const float* beta,
                              const float* film, int film_ld, int T, float* dgamma, float* dbeta, float* dfilm,
                              int dfilm_ld, const float* add, int add_ld, void* stream);
/* lfvdm_gn_bwd_stats + lfvdm_gn_bwd_apply_params in ONE launch (the training path): the per-channel sums stay in the
 * workgroup that owns the (sample, 8 groups) slice; dx is written (not accumulated) to out0 / out1.  add2 (optional, like
 * add): a third gradient of the same input - what the decoder's skip connection sends back to an encoder output. */
int lfvdm_gn_bwd_fused(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                       const float* coefA, const float* coefB, const float* stats, int act, float* out0, float* out1,
                       const float* gamma, const float* beta, const float* film, int film_ld, int T, float* dgamma,
                       float* dbeta, float* dfilm, int dfilm_ld, const float* add, int add_ld, const float* add2,
                       int add2_ld, void* stream);
/* lfvdm_gn_bwd_fused for the deterministic mode: the same ONE launch (statistics + dx, add / add2 folded in), but the
 * per-(sample, channel) sums are STORED to sums[N][C][2] (each pair by exactly one workgroup) instead of being added to the
 * parameter gradients with float atomics; lfvdm_gn_param_grads then reduces them in sample order. */
int lfvdm_gn_bwd_fused_sums(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                            const float* coefA, const float* coefB, const float* stats, int act, float* out0, float* out1,
                            const float* add, int add_ld, const float* add2, int add2_ld, float* sums, void* stream);
/* Large-map form of the GroupNorm backward (what lfvdm_gn_apply_ws is to the forward; reference nn.py:17-19 through
 * unet.py:194-207,399-403 at pixel-space map sizes): a workgroup owns a chunk of positions of a (sample, 8 groups) slice,
 * chunk sums -> fixed-order combination in every consumer workgroup -> dx: two launches of thousands of workgroups instead
 * of N*4 workgroups whatever P is.  ws: lfvdm_gn_bwd_ws_floats(C0 + C1, N, P) floats of scratch; 0 means the slice is small
 * and the single-workgroup kernels (lfvdm_gn_bwd_fused / _stats + _apply) are the ones to call.  dx is written (not
 * accumulated) to out0 / out1 with add / add2 folded in.  The per-(sample, channel) sums go to sums_out[N][C][2] when it is
 * not NULL (fixed order: deterministic mode, autograd delivery) and / or - dgamma, dbeta not NULL - into the parameter /
 * FiLM gradients with float atomics exactly as lfvdm_gn_bwd_fused does.  dx itself is deterministic either way. */
long lfvdm_gn_bwd_ws_floats(int C, int N, int P);
int lfvdm_gn_bwd_ws(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                    const float* coefA, const float* coefB, const float* stats, int act, float* out0, float* out1,
                    const float* gamma, const float* beta, const float* film, int film_ld, int T, float* dgamma,
                    float* dbeta, float* dfilm, int dfilm_ld, const float* add, int add_ld, const float* add2, int add2_ld,
                    float* sums_out, float* ws, long ws_floats, void* stream);
/* GroupNorm(+FiLM) parameter gradients from the sums of lfvdm_gn_bwd_stats: dgamma / dbeta [C] are ACCUMULATED
 * (+=, fixed order), dfilm [N/T][2C] (d scale | d shift of unet.py:199-203; row strides film_ld / dfilm_ld) is
 * written when film != NULL. */
int lfvdm_gn_param_grads(const float* sums, const float* gamma, const float* beta, const float* film, int film_ld, int T,
                         float* dgamma, float* dbeta, float* dfilm, int dfilm_ld, int N, int C, void* stream);
/* Temporal GroupNorm backward: dx from dy; dgamma/dbeta [C] are ACCUMULATED with float atomics. */
int lfvdm_gn_temporal_bwd(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* dgamma,
                          float* dbeta, int B, int T, int P, int C, int accumulate, void* stream);
/* deterministic form: det_ws >= 2 * ceil(B*P / 4) * C floats */
int lfvdm_gn_temporal_bwd_det(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* dgamma,
                              float* dbeta, int B, int T, int P, int C, int accumulate, float* det_ws, int64_t det_ws_floats,
                              void* stream);

/* Temporal GroupNorm of rpe.py:135-137: statistics over (C/32 channels x T frames) for each
 * (b, pixel); writes the normalised tensor (it is also the residual of rpe.py:172).
 * x, y: [B*T][P][C]. */
int lfvdm_gn_temporal(const float* x, const float* gamma, const float* beta, float eps, float* y,
                      int B, int T, int P, int C, void* stream);

/* Temporal GroupNorm + the temporal attention's qkv projection in ONE launch (rpe.py:136 `x = self.norm(x)` + :139
 * `qkv = self.qkv(x)`, temporal instance): x [(b*T + t)*P + p][C] = the block input, gamma / beta / eps = norm.weight /
 * norm.bias / norm.eps (GroupNorm32(32, C) over C/32 channels x T frames per (b, pixel), nn.py:93-101), Wqkv [3C][C] /
 * bqkv [3C] = qkv.weight / qkv.bias (nn.Linear), qkv [B*T*P][3C].  xn_out [B*T*P][C] (may be NULL, must not alias x)
 * receives the normalised rows - the residual of the block's output projection (rpe.py:172).  Replaces lfvdm_gn_temporal
 * + lfvdm_conv_igemm (1x1) in the sampler plan; same arithmetic per element as those two (fp32 MFMA, two-pass statistics),
 * a different summation order.  _ok: LFVDM_OK for C = 64 / 128 / 256, T <= 24, P a multiple of the pixel strip (2 at
 * C = 64) and - at 128 / 256 channels - a projection of at most 0.5 GFLOP (above that the two launches are faster), else
 * LFVDM_E_UNSUPPORTED (keep the two launches); the launch itself accepts every covered shape. */
int lfvdm_gn_temporal_qkv_ok(int B, int T, int P, int C);
int lfvdm_gn_temporal_qkv(const float* x, const float* gamma, const float* beta, float eps, float* xn_out,
                          const float* Wqkv, const float* bqkv, float* qkv, int B, int T, int P, int C, void* stream);

/* 1x1 projection + bias + residual + the next GroupNorm in ONE launch, for frames of 256 or 64 positions (16x16, 8x8): the
 * temporal attention's `x + self.proj_out(out)` (rpe.py:171-172; o [N*P][C] = the attention output, W [C][C] / bias [C] =
 * proj_out.weight / .bias, res [N*P][C] = the temporally normalised tokens) followed by the spatial attention's
 * `self.norm` (rpe.py:136; GroupNorm32(32, C) per frame in fp32, nn.py:93-101; gamma / beta / eps), or - act =
 * LFVDM_ACT_SILU - the spatial attention's output projection followed by the U-Net head's GroupNorm + SiLU
 * (unet.py:418-422).  out [N*P][C] = act(GroupNorm(sum)); raw_out [N*P][C] (may be NULL) = the sum itself, for consumers
 * that also read it; out / raw_out must not alias o, res or each other.  Replaces lfvdm_conv_igemm (1x1) + lfvdm_gn_apply
 * where the GroupNorm does not fit the GEMM's epilogue (a frame is more rows than a tile) or costs more there than here (8x8).  _ok: LFVDM_OK
 * for P = 256 / 64 and C = 64 / 128, else LFVDM_E_UNSUPPORTED. */
int lfvdm_proj_gn_ok(int N, int P, int C);
int lfvdm_proj_gn(const float* o, const float* W, const float* bias, const float* res, const float* gamma, const float* beta,
                  float eps, int act, float* out, float* raw_out, int N, int P, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Small-M grouped linear ("row-dot"): out[m][o] = sum_k actin(in[m][k]) * W[o][k] + b[o],
 * m < M <= 8.  One launch evaluates a whole table of jobs (time_embed.{0,2}, every
 * ResBlock.emb_layers.1, every RPENet.embed_diffusion_time: nn.py:105-123, unet.py:303-308,
 * 157-163,196; rpe.py:29).  Job tables live in device memory (see lfvdm_rowdot_job).
 * in_mode: 0 = plain, 1 = SiLU(in), 2 = in is a scalar timestep per row and the operand is its
 * sinusoidal embedding [cos | sin] of width K (nn.py:105-123).
 * ------------------------------------------------------------------------------------- */
typedef struct lfvdm_rowdot_job {
    const float* W;   /* [O][K] */
    const float* b;   /* [O] */
    const float* in;  /* [M][ldin] (or [M] timesteps for in_mode 2) */
    float* out;       /* [M][ldout] */
    int32_t K, O, M, ldin, ldout, in_mode;
    int32_t row0;     /* first global output-row index of this job (prefix sum of O) */
    int32_t pad_;
} lfvdm_rowdot_job;

int lfvdm_rowdot(const lfvdm_rowdot_job* jobs_dev, int njobs, int total_rows, void* stream);
/* out = silu(in) elementwise (n a multiple of 4), with the exact silu of in_mode 1 above: a launch over many batch rows
 * materialises it once and runs in in_mode 0 (bitwise the same results). */
int lfvdm_silu(const float* in, float* out, int64_t n, void* stream);

/* Backward of the same grouped linears (autograd of nn.Linear, train_util.py:328): per job
 *   dW[o][k] += sum_m dout[m][o] * actin(in[m][k]);  db[o] += sum_m dout[m][o]   (single writer, plain +=)
 *   din[m][k] += sum_o dout[m][o] * W[o][k]   (gradient w.r.t. the ACTIVATED input, float atomics; NULL = skip)
 * One wave per 8 output rows: task0 = first task of the job (prefix sum of ceil(O/8)), K % 4 == 0. */
typedef struct lfvdm_rowdot_bwd_job {
    const float* W;    /* [O][K] */
    const float* in;   /* as in the forward job */
    const float* dout; /* [M][lddout] */
    float* dW;         /* [O][K] accumulated */
    float* db;         /* [O] accumulated, or NULL */
    float* din;        /* [M][lddin] accumulated, or NULL */
    int32_t K, O, M, ldin, lddout, lddin, in_mode, task0;
} lfvdm_rowdot_bwd_job;
int lfvdm_rowdot_bwd(const lfvdm_rowdot_bwd_job* jobs_dev, int njobs, int total_tasks, void* stream);
/* deterministic form: din_base / din_n = the array every job's `din` rows live in (NULL: no job has a din);
 * det_ws >= total_tasks * din_n floats */
int lfvdm_rowdot_bwd_det(const lfvdm_rowdot_bwd_job* jobs_dev, int njobs, int total_tasks, float* din_base, int64_t din_n,
                         float* det_ws, int64_t det_ws_floats, void* stream);

/* ---------------------------------------------------------------------------------------
 * All RPENet output projections of one forward in one launch (rpe.py:20-31):
 *   R[b,t,s,:] = Wout * SiLU(tproj[b,:] + Wd * feats(fi[b,t]-fi[b,s]) + bd) + bout
 * tproj already holds embed_diffusion_time(emb)+its bias (from lfvdm_rowdot).
 * ------------------------------------------------------------------------------------- */
typedef struct lfvdm_rpe_job {
    const float* tproj; /* [B][C] */
    const float* Wd;    /* [C][3] */
    const float* bd;    /* [C] */
    const float* Wout;  /* [C][C] */
    const float* bout;  /* [C] */
    float* R;           /* [B][T][T][C] */
    int32_t C;
    int32_t tile0;      /* first workgroup index of this job */
    int32_t tproj_ld;   /* row stride of tproj (>= C) */
    int32_t pad_;
    float* act;         /* optional [B*T*T][C]: the hidden activations SiLU(...) (operand of the output layer's weight
                         * gradient in training); NULL = not stored */
} lfvdm_rpe_job;

int lfvdm_rpe_nets(const lfvdm_rpe_job* jobs_dev, int njobs, int total_tiles, const int64_t* frame_indices_i64,
                   int B, int T, void* stream);
/* The same, told the largest `C` among the jobs (<= 512): the launch then reserves LDS for that width only, which lets
 * several workgroups share a CU.  lfvdm_rpe_nets assumes 512. */
int lfvdm_rpe_nets_maxc(const lfvdm_rpe_job* jobs_dev, int njobs, int total_tiles, const int64_t* frame_indices_i64,
                        int B, int T, int max_channels, void* stream);

/* Backward of the RPE networks of a training step in ONE launch (autograd of rpe.py:20-31 through the output layer's
 * input and the hidden layer): per job, from dR [B*T*T][C],
 *   d_act = dR * Wout  (Wout_t = Wout transposed, [C in][C out] as packed by lfvdm_pack_conv_weight_t),
 *   dhid = d_act * silu'(hid)  (hid recomputed from tproj, Wd, bd and the frame indices),
 *   dtproj[b][c] += sum_{t,s} dhid,  dWd[c][j] += sum dhid * feats[j],  dbd[c] += sum dhid      (float atomics).
 * The output layer's own weight / bias gradient is a lfvdm_conv_wgrad_grouped job on the stored activations.
 * Needs T*T >= 32 (a 32-row tile then spans at most two batch elements) and C % 32 == 0. */
typedef struct lfvdm_rpe_bwd_job {
    const float* tproj; /* [B][C], row stride tproj_ld */
    const float* Wd;    /* [C][3] */
    const float* bd;    /* [C] */
    const float* Wout_t;/* [C][C]: Wout_t[c][o] = Wout[o][c] */
    const float* dR;    /* [B*T*T][C] */
    float* dtproj;      /* [B][C], row stride dtproj_ld, accumulated */
    float* dWd;         /* [C][3], accumulated */
    float* dbd;         /* [C], accumulated */
    int32_t C;
    int32_t tile0;
    int32_t tproj_ld;
    int32_t dtproj_ld;
} lfvdm_rpe_bwd_job;

int lfvdm_rpe_nets_bwd(const lfvdm_rpe_bwd_job* jobs_dev, int njobs, int total_tiles, const int64_t* frame_indices_i64,
                       int B, int T, void* stream);
/* deterministic form: det_ws >= total_tiles * 5 * 512 floats; every job has total_tiles / njobs tiles */
int lfvdm_rpe_nets_bwd_det(const lfvdm_rpe_bwd_job* jobs_dev, int njobs, int total_tiles, const int64_t* frame_indices_i64,
                           int B, int T, float* det_ws, int64_t det_ws_floats, void* stream);

/* Hidden layer of one RPENet for the training path (rpe.py:20-31 before the output layer) and its backward:
 *   act[r][c] = silu(tproj[b][c] + Wd[c][0..2] . feats[r][0..2] + bd[c]),  rows r = (b, t, s), rows_per_b = T*T,
 *   tproj = embed_diffusion_time(emb) [B][C], feats [B*T*T][3] (log1p(relu(d)), log1p(relu(-d)), d == 0).
 * tproj / dtproj have row strides tproj_ld / dtproj_ld.
 * bwd: dtproj [B][C], dWd [C][3] and dbd [C] are ACCUMULATED with float atomics (dtproj must be zeroed). */
int lfvdm_rpe_front(const float* tproj, int tproj_ld, const float* feats, const float* Wd, const float* bd, float* act, int B,
                    int rows_per_b, int C, void* stream);
int lfvdm_rpe_front_bwd(const float* tproj, int tproj_ld, const float* feats, const float* Wd, const float* bd,
                        const float* d_act, float* dtproj, int dtproj_ld, float* dWd, float* dbd, int B, int rows_per_b,
                        int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Attention cores (rpe.py:143-169).  qkv rows are token-major [M][3C] with the reference's
 * channel order [3][heads][F]; o rows are [M][C].
 *   spatial : tokens of one frame n attend to each other (no mask, no RPE)
 *   temporal: the T frames of one (b, pixel) attend with RPE biases and the two-clique mask
 * attn_out (optional, may be NULL): softmax probabilities for logging
 *   spatial [N][heads][P][P], temporal [B*P][heads][T][T].
 * ------------------------------------------------------------------------------------- */
/* lse_out (optional, may be NULL): log-sum-exp of every query row [N][heads][P], kept for the backward. */
int lfvdm_attn_spatial(const float* qkv, float* o, float* attn_out, float* lse_out, int N, int P, int C, int heads,
                       void* stream);

/* Spatial attention WITH its qkv projection (rpe.py:139 + :143-169, spatial instance): xn = the normalised tokens
 * [N*P][C] (what qkv's nn.Linear reads), Wqkv [3C][C] / bqkv [3C] = qkv.weight / qkv.bias, o [N*P][C].  One workgroup
 * per (frame, head) stages the frame and the head's filter rows in LDS, projects on MFMA and runs the flash loop over
 * resident keys: replaces lfvdm_conv_igemm (qkv) + lfvdm_attn_spatial in the sampler plan.
 * lfvdm_attn_spatial_fused_ok: LFVDM_OK if the shape is covered (head dim 16 / 32 / 64, C a multiple of 64, frame +
 * filters + q/k/v tiles within 160 KB of LDS), else LFVDM_E_UNSUPPORTED (use the two-launch form). */
int lfvdm_attn_spatial_fused_ok(int N, int P, int C, int heads);
int lfvdm_attn_spatial_fused(const float* xn, const float* Wqkv, const float* bqkv, float* o, int N, int P, int C, int heads,
                             void* stream);

/* Backward of the spatial core (autograd of rpe.py:143-169 with attn_mask = None, no RPE): from qkv, the
 * forward output o, its gradient d_o and the saved lse, writes dqkv [M][3C] (same layout as qkv).
 * delta_ws: workspace of N*heads*P floats (rowdot(o, d_o)).  S and P are recomputed tile by tile. */
int lfvdm_attn_spatial_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* delta_ws,
                           float* dqkv, int N, int P, int C, int heads, void* stream);

int lfvdm_attn_temporal(const float* qkv, const float* Rq, const float* Rk, const float* Rv,
                        const float* mask /* [B][T] or NULL */, float* o, float* attn_out,
                        int B, int T, int P, int C, int heads, void* stream);
/* The same with R_q / R_k / R_v given as tables over the sampler's timesteps, [n_t][B][T][T][C]: batch element b uses
 * slice rsel[b] (the device-side step counter); rsel == NULL is lfvdm_attn_temporal. */
int lfvdm_attn_temporal_sel(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                            float* o, float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel,
                            void* stream);
/* The same with the tables held as a ROLLING WINDOW of `ring` timesteps ([ring][B][T][T][C]): batch row b reads slice
 * rsel[b] % ring (ring = 0: lfvdm_attn_temporal_sel).  The sampler refills half a ring at a time between graph launches
 * (Plan.ensure_R): 1/8 of the whole-chain tables' memory at ring = 128 for a 1000-step chain, same values. */
int lfvdm_attn_temporal_ring(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                             float* o, float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel,
                             int ring, void* stream);

/* Backward of the temporal core: from qkv, d_o (gradient of the core output), the three R tensors and the mask
 * writes dqkv [M][3C] and dRq / dRk / dRv [B][T][T][C] (every element written, no atomics).
 * ws_p, ws_ds: workspaces of B*P*heads*T*T floats each (rows of P and dS, recomputed from the inputs). */
int lfvdm_attn_temporal_bwd(const float* qkv, const float* d_o, const float* Rq, const float* Rk, const float* Rv,
                            const float* mask, float* ws_p, float* ws_ds, float* dqkv, float* dRq, float* dRk,
                            float* dRv, int B, int T, int P, int C, int heads, void* stream);

/* ---------------------------------------------------------------------------------------
 * Diffusion step math (gaussian_diffusion.py).  Tables are device fp32 arrays gathered by
 * the int64 timestep t[b]; tensors are (B, inner) contiguous.
 * ------------------------------------------------------------------------------------- */
/* q_sample, :200-218 */
int lfvdm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp,
                   const float* sqrt_1macp, float* out, int B, int inner, void* stream);
/* p_mean_variance (eps, fixed variance) + p_sample update, :290-346,369-401.
 * pred_xstart / mean_out may be NULL. */
int lfvdm_p_sample(const float* x, const float* eps, const float* noise, const int64_t* t,
                   const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, const float* coef1,
                   const float* coef2, const float* log_var, int clip, float* sample, float* pred_xstart,
                   float* mean_out, int B, int inner, void* stream);
/* The same update with the noise drawn INSIDE the kernel (the sampler's replayed step: one launch less than
 * th.randn + lfvdm_p_sample): Philox4x32-10 keyed by seed[0] (device int64, one value per chain), counter = (element quad,
 * batch row, timestep t[b]) -> Box-Muller; a (seed, timestep, element) triple always gives the same value.  noise_out
 * (optional) receives the standard normal values that were used. */
int lfvdm_p_sample_rng(const float* x, const float* eps, float* noise_out, const int64_t* t, const float* sqrt_recip_acp,
                       const float* sqrt_recipm1_acp, const float* coef1, const float* coef2, const float* log_var,
                       int clip, float* sample, float* pred_xstart, float* mean_out, int B, int inner,
                       const int64_t* seed, void* stream);
/* The U-Net's 3x3 output convolution (unet.py:399-403,462-464) and the update above in ONE launch - the last two
 * launches of a replayed sampling step: eps = conv(act) + bias from the channels-last rows act [B*T*H*W][C] = SiLU(GN(h))
 * and the packed filters Wp [Cout][9][C] (lfvdm_pack_conv_weight), then lfvdm_p_sample_rng's arithmetic with the same
 * noise stream (noise_in != NULL: that noise instead, seed unused).  eps_out / noise_out / pred_xstart / mean_out may be
 * NULL; x and sample may alias.  _ok: 0 if the shape is covered (Cout 3 or 4, C 64 / 128 / 256, W % 4 == 0). */
int lfvdm_conv_out_psample_ok(int N, int H, int W, int C, int Cout);
int lfvdm_conv_out_psample(const float* act, const float* Wp, const float* bias, float* eps_out, const float* x,
                           const float* noise_in, float* noise_out, const int64_t* t, const float* sqrt_recip_acp,
                           const float* sqrt_recipm1_acp, const float* coef1, const float* coef2, const float* log_var,
                           int clip, float* sample, float* pred_xstart, float* mean_out, int B, int T, int H, int W, int C,
                           int Cout, const int64_t* seed, void* stream);
/* Sampler clock of the captured denoising step (the loop `for i in indices: t = th.tensor([i]*B)` of
 * gaussian_diffusion.py:509-512 and _WrappedModel's timestep map, respace.py:117-122, kept on the device):
 * t[b] <- max(t[b] - 1, 0);  model_t[b] <- model_timestep_table[t[b]]. */
int lfvdm_sampler_tick(int64_t* t, const float* model_timestep_table, float* model_t, int B, void* stream);
/* The same clock, plus: rows[b*rows_ld + 0:row_floats] <- rows_all[(t[b]*B + b)*rows_ld + 0:row_floats] (multiples of 4;
 * B <= 64; both buffers have rows of rows_ld floats).
 * rows_all is a table over all timesteps of the chain of everything the network derives from the timestep alone
 * (time-embedding MLP -> FiLM rows of every ResBlock, unet.py:303-308,157-163,199-203), built once per chain by the
 * per-step kernels on a virtual batch; the per-step embedding launches disappear from the denoising step. */
int lfvdm_sampler_tick_fetch(int64_t* t, const float* model_timestep_table, float* model_t, int B, const float* rows_all,
                             int rows_ld, float* rows, int row_floats, void* stream);
/* masked mean of squared error, :787-788 + nn.py:86-92: out[b] = mean_inner((a-b)^2 * mask[b,frame]).
 * mask is (B, T) (broadcast over the per-frame block of `frame_inner` elements) or NULL. */
int lfvdm_masked_mse(const float* a, const float* b, const float* mask, float* out, int B, int T,
                     int frame_inner, void* stream);
/* its backward w.r.t. pred: dpred = -2 (target - pred) * mask[b][t] * g[b] / (T * frame_inner); mask may be NULL. */
int lfvdm_masked_mse_bwd(const float* target, const float* pred, const float* mask, const float* g, float* dpred, int B, int T,
                         int frame_inner, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused AdamW + EMA + gradient-norm over one flat fp32 arena (train_util.py:346-357, nn.py:55-65).
 * torch.optim.AdamW semantics (decoupled weight decay, bias correction); ema <- rate*ema + (1-rate)*p.
 * ------------------------------------------------------------------------------------- */
typedef struct lfvdm_adamw_args {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema[4];
    float ema_rate[4];
    int32_t n_ema;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay;
    float bias_corr1, bias_corr2_sqrt; /* 1 - beta1^t, sqrt(1 - beta2^t) */
    float grad_scale;                  /* applied to g first (1/world_size after a SUM all-reduce) */
    float* grad_sqsum;                 /* optional: += sum (grad_scale*g)^2 */
    const int32_t* skip_flag;          /* optional: a non-zero word (the `timed_out` word of lfvdm_flag_wait) turns the
                                          launch into a no-op: nothing is read or written */
    const int32_t* skip_flag2;         /* optional second word, same meaning: the element of the gradient arena behind the
                                          last bucket that lfvdm_flag_wait2 raises to 1.0f and the bucket's SUM all-reduce
                                          carries to every rank (read as bits: any non-zero value skips) */
} lfvdm_adamw_args;

int lfvdm_adamw_ema(const lfvdm_adamw_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training-batch assembly on the device (TrainLoop.prepare_training_batch, train_util.py:224-241, with the masks of
 * sample_all_masks, :193-222).  The host samples the index table (same sequence of random draws as the reference);
 * the gather and the mask / index tensors are produced here, inside the captured training step.
 *   pool  [B][Tp][frame_elems]  candidate frames of each batch element (e.g. video1 frames followed by the frames of
 *                               the padding video), frame_elems = C*H*W, a multiple of 4
 *   table [B][F][4] int32       {row in the pool, frame index (-> frame_indices), observed 0/1, latent 0/1}
 *   -> batch [B][F][frame_elems], frame_indices [B][F] int64, obs_mask / latent_mask [B][F] fp32 (the reference's
 *      (B, F, 1, 1, 1) tensors).
 * ------------------------------------------------------------------------------------- */
int lfvdm_prepare_batch(const float* pool, const int32_t* table, float* batch, int64_t* frame_indices, float* obs_mask,
                        float* latent_mask, int B, int F, int Tp, int frame_elems, void* stream);

/* Input compositing of the TRAINING forward (unet.py:441-450) as channels-last rows [N*H*W][ld] for the first 3x3 conv:
 * channels 0..Cx-1 = x*(1-obs[n]) + x0*obs[n], channel Cx = obs[n] (indicator), zero up to ld (>= Cx+1, multiple of 4).
 * x, x0: (N, Cx, H, W); obs: [N].  (Inference fuses this into lfvdm_conv_in; training keeps the rows for the weight
 * gradient of the first conv.) */
int lfvdm_compose_rows(const float* x, const float* x0, const float* obs, float* rows, int N, int Cx, int H, int W, int ld,
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * Persistent level chain: a run of CONSECUTIVE launches of one forward pass - implicit GEMMs and small-map GroupNorms of
 * the low-resolution levels (unet.py:194-207 ResBlocks, :91-114 / :60-88 resampling convs, the 1x1 projections of
 * :223-243) - executed as ONE launch of <= 256 co-resident workgroups.  Every stage keeps the tile decomposition, K-slice
 * order and epilogue of its stand-alone kernel (bitwise the same results); what replaces the launch boundaries is a
 * point-to-point dependency per work item: an item lists the output tiles of earlier stages whose bytes it reads, polls
 * their FLAGS (one lane per flag, sc1 loads, s_sleep between polls) and only then stages its activations - with sc1
 * loads, from tensors the producers published with sc1 (write-through) stores, so no release / acquire fence sits on the
 * path (0.7-1.15 us per hop against 3.9 us for a counter barrier among 256 workgroups: tools/grid_barrier_bench).  The
 * filter pieces of an item's first K chunks, which depend on nothing the chain produces, are in flight before it waits.
 * A flag holds the GENERATION of the launch that completed its tile; the generation advances once per launch (last
 * workgroup out), so nothing is reset between replays.
 *
 * Every wait is bounded: a poller that has waited longer than `timeout_s` of wall clock raises ctl[LFVDM_CHAIN_CTL_ABORT]
 * and leaves; every poller also watches that word, so all workgroups reach the end of the kernel.  The host reads the word
 * (it stays raised), falls back to the per-launch plan and reports the failure.
 *
 * Use: fill kind + conv / gn of every stage (conv.tune = a code whose tile is <1,1,4,1> with 32-channel chunks and a
 * plain split-K factor; outputs of different stages must be DIFFERENT buffers - there is no launch boundary to order a
 * reuse), call lfvdm_chain_plan (host only: work items, flags, dependency lists, split-K workspace offsets), point each
 * conv.splitk_ws / splitk_cnt at ws + ws_off / cnt + cnt_off of zero-initialised tickets, copy stages and deps to the
 * device, zero flags[n_flags] and ctl[LFVDM_CHAIN_CTL_INTS] once, launch lfvdm_level_chain with the planned grid.
 * ------------------------------------------------------------------------------------- */
#define LFVDM_CHAIN_CONV 0
#define LFVDM_CHAIN_GN 1
/* Round 6 - SAMPLE-LOCAL stage (csrc/conv_local_body.h): the same lfvdm_conv_args, decomposed for maps of <= 16 pixels
 * (Ho * Wo a power of two) where a 3x3 convolution only mixes the pixels of ONE sample and GroupNorm32 is per sample and
 * group (unet.py:194-207, nn.py:17-19): work item = (16 * cfg output rows = whole samples, 16 filters = whole groups, ALL
 * of K) - no split-K over workgroups, no slab / ticket seam, no cross-wave normalisation.  The item's filter slice
 * [16][K] depends on nothing the chain computes and is fetched into LDS before the item waits (for a workgroup's next item:
 * while it finishes the current one); only the item's activation rows are read behind the flag hop.  K order differs
 * from the tile kernels: results agree to rounding, not bitwise.  cfg = row tiles per item (1 | 2); conv.tune is ignored.
 * Needs Cout % 16 == 0, (C0 + C1) % 64 == 0, (s2C0 + s2C1) % 64 == 0, front + filter slice <= 159.5 KB of LDS. */
#define LFVDM_CHAIN_LOCAL 2
#define LFVDM_CHAIN_MAX_DEPS 64
#define LFVDM_CHAIN_CTL_EPOCH 0    /* int index: generation of the last completed launch */
#define LFVDM_CHAIN_CTL_EXIT 32    /* workgroups that have left the current launch */
#define LFVDM_CHAIN_CTL_ABORT 64   /* non-zero: a wait timed out */
#define LFVDM_CHAIN_CTL_INTS 96

typedef struct lfvdm_gn_args {       /* lfvdm_gn_apply's arguments (one-wave form: P <= 256, (C0+C1)/32 in {2,4,8,16}) */
    const float* src0;
    const float* src1;
    int32_t C0, C1, N, P;
    const float* gamma;
    const float* beta;
    const float* film;
    int32_t film_div, film_ld;
    float eps;
    int32_t act;
    float* out;              /* first element this stage writes */
    int32_t cg;              /* channels per group (0: (C0 + C1) / 32): a PART of a wider normalisation (lfvdm_gn_apply_part) */
    int32_t ldo;             /* row stride of out (0: C0 + C1) */
    float* out_base;         /* the buffer `out` points into (NULL: out) and its first column there: what readers of the */
    int32_t out_col;         /* buffer are matched against when dependencies are planned */
    int32_t pad_;
} lfvdm_gn_args;

typedef struct lfvdm_chain_stage {
    int32_t kind;            /* LFVDM_CHAIN_CONV | LFVDM_CHAIN_GN | LFVDM_CHAIN_LOCAL (caller also sets cfg = row tiles) */
    /* filled by lfvdm_chain_plan: */
    int32_t n_items;         /* work items (conv: the XCD-aware flat grid of the stand-alone launch, padding included) */
    int32_t flag_base;       /* flags[flag_base + output unit] (conv: output tile; gn: work item) */
    int32_t n_flags;
    int32_t dep_base;        /* deps[dep_base + item * dep_stride] = count, followed by `count` flag indices;  local: SIX counts -
                              * one list per wave (the producers of the channel quarter that wave stages and multiplies) and one
                              * per row tile (the producers of its residual rows) - followed by the six lists */
    int32_t dep_stride;
    int32_t cfg;             /* conv: kernel-body instance;  local: row tiles of 16 rows per item (1 | 2), set by the caller */
    int32_t kz;              /* conv: K slices over workgroups;  local: floats of LDS in front of the filter slice */
    int32_t nt2;             /* conv: filter tiles;  local: filter slices (Cout / 16) */
    int32_t wg_off;          /* work item i runs on workgroup wg_lo + (i + wg_off) mod wg_count: stages that depend on nothing inside
                              * the chain (the skip half of a concat GroupNorm) are put on the workgroups the GEMM stages leave
                              * idle; consecutive sample-local stages are rotated over their workgroups */
    int32_t side;            /* set by the CALLER: 1 = off the chain's critical path (the skip side of a decoder concat: its operands
                              * come from earlier launches, its consumer sits deep in the chain).  If a chain has such stages,
                              * lfvdm_chain_plan reserves the top 5/16 of the grid for them: a workgroup walks its items in stage
                              * order and waits where an item's producers are not done - side work must not sit in front of a
                              * main-path item on the same workgroup */
    int32_t wg_lo;           /* planner: the stage's workgroups are [wg_lo, wg_lo + wg_count) */
    int32_t wg_count;
    int64_t ws_off;          /* conv: this stage's slab region (floats) and ticket region (ints) in the chain's workspace */
    int64_t cnt_off;
    lfvdm_conv_args conv;
    lfvdm_gn_args gn;
} lfvdm_chain_stage;

/* Host-side planning (no GPU work).  deps: caller's array of deps_cap ints.  -> LFVDM_E_UNSUPPORTED if a stage cannot run
 * in a chain (tile configuration, layout, more than LFVDM_CHAIN_MAX_DEPS producers for an item, a buffer written twice).
 * *grid on ENTRY: the most workgroups the device can keep resident for this chain (lfvdm_chain_capacity; <= 0: the 256 of
 * a whole MI355X) - the planned grid never exceeds it (work items stride over the grid, so any grid is legal). */
int lfvdm_chain_plan(lfvdm_chain_stage* stages, int n_stages, int32_t* deps, int64_t deps_cap, int64_t* deps_used,
                     int32_t* n_flags, int64_t* ws_floats, int64_t* cnt_ints, int32_t* grid, int32_t* lds_bytes);
/* LFVDM_OK if this launch could be a chain stage with this tune code (tile / chunk / split-K family the chain kernel holds) */
int lfvdm_chain_conv_ok(const lfvdm_conv_args* a);
int lfvdm_chain_gn_ok(int C0, int C1, int N, int P);
/* LFVDM_OK if this launch could be a LFVDM_CHAIN_LOCAL stage with `row_tiles` (1 | 2) tiles of 16 rows per item */
int lfvdm_chain_local_ok(const lfvdm_conv_args* a, int row_tiles);
/* Workgroups of the chain kernel the CURRENT device can hold at once with lds_bytes of dynamic LDS (CU count x occupancy):
 * the bound of a chain's grid - its waits only end if every workgroup is running.  < 0: the query failed.
 * lfvdm_level_chain refuses (LFVDM_E_UNSUPPORTED) a grid above it. */
int lfvdm_chain_capacity(int lds_bytes);
int lfvdm_level_chain(const lfvdm_chain_stage* stages_dev, int n_stages, const int32_t* deps_dev, int32_t* flags, int32_t* ctl,
                      int grid, int lds_bytes, double timeout_s, void* stream);

/* ---------------------------------------------------------------------------------------
 * Device-side semaphores between a replayed hipGraph and another stream: host-side plumbing of the bucketed gradient
 * exchange that stands in for DistributedDataParallel's overlapped buckets (train_util.py:116-125,309-313).
 *   lfvdm_flag_add : *flag += 1 once everything enqueued before it on `stream` has completed.  An ordinary kernel node
 *                    when captured, so every replay of the backward graph bumps the counter at that point.
 *   lfvdm_flag_wait: `stream` proceeds when *flag - target >= 0 (one parked wave polls with device-scope loads), or
 *                    after `timeout_s` seconds, in which case *timed_out is set to 1 (the host must treat that as an
 *                    error).  Issued on the exchange's side stream in front of a bucket's all-reduce.
 * (An external event-record node would do the same; the HIP runtime of PyTorch-ROCm 2.10 refuses it under capture.)
 * ------------------------------------------------------------------------------------- */
int lfvdm_flag_add(int32_t* flag, void* stream);
int lfvdm_flag_wait(const int32_t* flag, int32_t target, double timeout_s, int32_t* timed_out, void* stream);
/* the same, and on a timeout *timed_out_f32 = 1.0f as well (optional; see lfvdm_adamw_args.skip_flag2) */
int lfvdm_flag_wait2(const int32_t* flag, int32_t target, double timeout_s, int32_t* timed_out, float* timed_out_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LFVDM_HIP_H */
