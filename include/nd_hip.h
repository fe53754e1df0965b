/*
 * nd_hip.h -- C ABI of libnd_hip.so: the MI355X (gfx950) kernels behind the nice-diffusion sampling hot path.
 *
 * The reference (edogariu/nice-diffusion) has no FFI: its hot path is a chain of PyTorch ops.  Each entry point
 * below replaces one such chain (cited as /root/reference file:line) and is what a ctypes/cffi binding binds.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer (tensor.data_ptr()); no ownership is transferred;
 *   - activations are fp32 NHWC ("pixel-major"): element (img, y, x, c) lives at ((img*H + y)*W + x)*ld + c,
 *     with ld >= C the pixel stride in floats; ld and every channel count handed to a vector kernel are
 *     multiples of 4 and base pointers are 16-byte aligned;
 *   - every launch is asynchronous on `stream` (a hipStream_t), allocation-free and sync-free, so that a
 *     caller may capture it into a hipGraph;
 *   - return value: 0 = ok, < 0 = error (ND_E_*), message via nd_last_error() (thread-local); nothing throws.
 */
#ifndef ND_HIP_H
#define ND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nd_stream_t; /* hipStream_t */

#define ND_OK 0
#define ND_E_ARG (-1)     /* bad argument (shape / alignment / unsupported combination) */
#define ND_E_LAUNCH (-2)  /* HIP launch error */
#define ND_E_ARCH (-3)    /* device is not gfx950 */

/* flags for nd_conv_nhwc */
#define ND_CONV_IN_UP2X 1   /* input is stored at half resolution and read through nearest-2x upsampling */
#define ND_CONV_RES_UP2X 2  /* residual is stored at half resolution and read through nearest-2x upsampling */
#define ND_CONV_SILU_OUT 4  /* apply SiLU to the result (after bias; residual must be NULL) */
#define ND_CONV_GN_SILU 8   /* with gnA/gnB: SiLU after the fused GroupNorm affine of the INPUT */
#define ND_CONV_OUT_F32 16  /* nd_conv_bf16_nhwc only: write the fp32 accumulators (out is float*) instead of bf16 */

/* element type of activation tensors where an entry point takes `dtype` (weights/bias/statistics keep their own types) */
#define ND_DT_F32 0
#define ND_DT_BF16 1

/* flags for nd_groupnorm_apply_nhwc */
#define ND_GN_SILU 1        /* SiLU after the affine */
#define ND_GN_POOL2 2       /* write the 2x2 average of the activated values ([NI,H/2,W/2,C]) */

/* sampler variance kinds for nd_ddpm_step (reference: VarType, diffusion.py:552-572) */
#define ND_VAR_FIXED 0          /* 'small' / 'large': log-variance comes from coef column 6 */
#define ND_VAR_LEARNED 1        /* model output second half is the log-variance */
#define ND_VAR_LEARNED_INTERP 2 /* second half interpolates between coef columns 6 (min) and 7 (max) */

#define ND_COEF_COLS 8 /* per-step fp32 coefficient row: see nd_ddim_step / nd_ddpm_step */

int nd_version(void);
/* 16 hex digits of the SHA-256 over the library's sources (every .hip / .h / .inc file under csrc/ and this header, in name
 * order; the Makefile computes it and every change of one character of any of them changes it).  Measured artefacts --
 * tune caches, PMC tables, bench lines -- are stamped with it, so "taken on another build" is detected, not remembered. */
const char* nd_build_id(void);
/* "" for the product build; otherwise the space-separated NAME=value list of the timing-only / diagnostic macros
 * (ND_F4ABL_*, ND_HABL_*, ND_WABL_*, ND_*_DIAG, ...; csrc/nd_variant_flags.inc) some translation unit was compiled
 * with.  Such a library computes wrong results by construction; the Python host refuses it unless ND_ALLOW_ABLATION=1. */
const char* nd_build_flags(void);
const char* nd_last_error(void);
/* gcnArchName of the current device (e.g. "gfx950:sramecc+:xnack-"); "" on error. */
const char* nd_device_arch(void);

/* 64-bit digest of a list of device buffers (ptrs / nbytes are DEVICE arrays of nseg entries; sizes are multiples of 4
 * bytes): *out = sum over words of mix(word, position in the concatenation).  Used by the host side to notice weights
 * that were rewritten in place (p.data.copy_(), EMA swaps landing on recycled addresses) under a cached launch plan,
 * whose repacked weight copies would otherwise go stale silently.  Reads every byte once (HBM-bound). */
int nd_checksum_segments(const void* const* ptrs, const int64_t* nbytes, int nseg, uint64_t* out, nd_stream_t stream);

/* ---- K1: sinusoidal timestep embedding (model.py:514-523): out[b] = [cos(t*f) | sin(t*f) | 0 pad] ----------
 * freqs [dim/2] is the constant fp32 table exp(arange(dim/2) * -(ln 10000)/(dim/2)) (model.py:516-517). */
int nd_timestep_embed(const int64_t* t, const float* freqs, int B, int dim, float* out, int ld_out,
                      nd_stream_t stream);

/* emb[b,:] += table[y[b],:] (y may be NULL), then silu_out[b,:] = silu(emb[b,:])  (model.py:459, :197 input) */
int nd_embedding_add_silu(float* emb, const float* table, const int64_t* y, int num_rows, int B, int D,
                          float* silu_out, nd_stream_t stream);

/* ---- K5/K6/K2: convolution as an implicit GEMM on fp32 MFMA ------------------------------------------------
 * out[img,y,x,n] = bias[n] + rowbias[img,n] + sum_{tap,c} in(img, y+dy-1, x+dx-1, c) * w[tap][n][c] + residual
 *   ksize 3: nn.Conv2d 3x3 stride 1 pad 1 (model.py:173,177,367,448,72);  ksize 1: nn.Conv2d 1x1 / nn.Conv1d k=1 /
 *   nn.Linear (model.py:169,247,253,180,349-351) -- for those pass NI=1, H=1, W=M.
 *   The input is the channel concatenation of x0 (C0 channels, stride ldx0) and x1 (C1, ldx1) (torch.cat,
 *   model.py:474); x1 may be NULL with C1 = 0.
 *   w is the weight tensor over the C0+C1 concatenated input channels in MFMA-fragment order (nd_repack_conv_weight).
 *   bias [N] | NULL;  rowbias [NI][ld_rowbias] | NULL (per-image bias: the timestep embedding of model.py:205);
 *   residual [NI,H,W,N] stride ldr | NULL (model.py:211, :291).  `variant` < 0 selects the tile shape automatically.
 *   gnA/gnB [NI][ld_gn] | NULL: GroupNorm(+AdaGN)(+SiLU with ND_CONV_GN_SILU) of the INPUT folded into the loader,
 *   in' = act(in * gnA[img][c] + gnB[img][c]) (coefficients from nd_groupnorm_coeffs; zero padding stays zero, as
 *   the reference pads the normalised tensor).  Needs H*W >= the pixel tile (one image per block).
 */
int nd_conv_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                 const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                 const float* residual, int ldr, float* out, int ldo,
                 int NI, int H, int W, int N, int ksize, int flags, int variant,
                 const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream);

/* Number of tile-shape variants nd_conv_nhwc accepts for `variant` (0 .. n-1). */
int nd_conv_num_variants(void);
/* The variant nd_conv_nhwc picks for a shape when `variant` < 0 (host-only query; < 0 on error), and a variant's
 * block tile (pixels x output channels) and thread count.  Used by bench.py to attribute launches to kernels. */
/* 1x1 convolution on a flat pixel list by the two-blocks-per-CU GEMM (variant 14 of nd_conv_nhwc; its restrictions apply)
 * that also leaves the per-channel partial statistics of its output behind: chstats [NI][H*W/128][sum | sum of squares][N]
 * in fp32, one row per 128-pixel run of an image, every row written by every launch (no zeroing, no atomics);
 * nd_groupnorm_stats_from_partials folds them -- the attention block's output projection + residual (model.py:291) feeds
 * the next block's in_norm (model.py:190), which then needs no pass over the tensor.  rows per image:
 * nd_conv1x1_stats_rows (0: the shape cannot -- H*W % 128, NI*H*W % 256, N % 4 must be 0); ldo must equal N. */
int nd_conv1x1_stats_rows(int NI, int H, int W, int N);
int nd_conv1x1_stats_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                          const float* w, const float* bias, const float* residual, int ldr, float* out, int ldo,
                          int NI, int H, int W, int N, int flags, float* chstats, nd_stream_t stream);
/* ---- K5 at the two ends of the UNet (csrc/nd_conv_edge.hip).
 * First convolution, Conv2d(in_channels <= 4, N, 3, padding=1) on the NHWC4 input (model.py:427-431): K = 4 channels x 9 taps
 * on v_mfma_f32_16x16x4_f32 with the pixel's four channels as the k index; W % 16 == 0, N % 16 == 0, N <= 256, ldx == 4.
 * w: nd_repack_conv_first_weight(OIHW [N][C0][3][3]) -> nd_conv_first_weight_floats(N) floats.  chstats | NULL: per-channel
 * partial statistics of the output, [NI][rows][sum | sum of squares][N] fp32 with rows = nd_conv3x3_first_stats_rows()
 * (0: shape not supported), every row written by every launch; folded by nd_groupnorm_stats_from_partials. */
int64_t nd_conv_first_weight_floats(int N);
int nd_repack_conv_first_weight(const float* w_oihw, float* out, int N, int C0, nd_stream_t stream);
int nd_conv3x3_first_stats_rows(int NI, int H, int W, int N);
int nd_conv3x3_first_nhwc(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo, int NI, int H,
                          int W, int N, float* chstats, nd_stream_t stream);
/* Last convolution, Conv2d(C, N <= 7, 3, padding=1) behind GroupNorm + SiLU (model.py:446-449), second half: the caller
 * runs ONE 1x1 convolution P[px][tap * N + n] = sum_c act(x[px][c]) w[n][c][ky][kx], tap = 3 ky + kx (nd_conv_nhwc with the
 * norm in its loader and the [9 N (padded)][C] weight), and this entry adds the nine shifted taps:
 * out[y][x][n] = bias[n] + sum_tap P[y + ky - 1][x + kx - 1][tap * N + n], zero outside the image.  ldp >= 9 N, even. */
int nd_conv3x3_taps_gather_nhwc(const float* P, int ldp, const float* bias, float* out, int ldo, int NI, int H, int W, int N,
                                nd_stream_t stream);
/* The same convolution split over K, for layers whose output has too few tiles to fill the chip and whose contraction is
 * long (7x7 .. 16x16 maps at small batch; the reference runs them as ordinary Conv2d, model.py:166-182): `splits` (2..16)
 * block rows each run a range of whole 32-channel chunks of the (concatenated) input and leave raw accumulators in
 * `workspace` (nd_conv_splitk_workspace_floats() floats, 16-byte aligned); a second launch adds them IN SPLIT ORDER and
 * applies bias / per-image bias / residual / SiLU -- deterministic, no atomics.  conv_mfma_kernel variants (0..8) only,
 * named explicitly; no fused GroupNorm, no ND_CONV_RES_UP2X; C0 + C1 a multiple of 32, N and the strides of 4. */
int nd_conv_splitk_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                        const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                        const float* residual, int ldr, float* out, int ldo,
                        int NI, int H, int W, int N, int ksize, int flags, int variant, int splits,
                        float* workspace, nd_stream_t stream);
/* The Winograd form split over K (conv_wino4_kernel, variant 12, only): 3x3 layers on small maps at large batch whose
 * output tiles do not fill the chip evenly; same scheme, restrictions and workspace (ksize 3) as nd_conv_splitk_nhwc. */
int nd_conv3x3_winograd_splitk_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                    const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                                    const float* residual, int ldr, float* out, int ldo,
                                    int NI, int H, int W, int N, int flags, int variant, int splits,
                                    float* workspace, nd_stream_t stream);
int64_t nd_conv_splitk_workspace_floats(int NI, int H, int W, int N, int C, int ksize, int splits);
int nd_conv_select_variant(int NI, int H, int W, int N, int ksize, int flags, int has_rowbias);
int nd_conv_variant_info(int variant, int* bm, int* bn, int* threads);

/* Winograd F(2x2,3x3) form of the same 3x3 stride-1 pad-1 convolution (2.25x fewer matrix instructions; fp32
 * throughout, the transforms only use 0, +-1, +-1/2).  Same arguments and fused options as nd_conv_nhwc with ksize 3;
 * H and W must be even; `w` must come from nd_repack_conv_weight_winograd (nd_conv_winograd_weight_floats floats);
 * `variant` in [0, nd_conv_winograd_num_variants()).  Callers pick direct vs Winograd per shape by measurement. */
int nd_conv_winograd_num_variants(void);
/* The same convolution by the position-split kernel (variant nd_conv_winograd_stats_variant()) which, besides the
 * output, leaves PARTIAL statistics of it behind: pstats[img][mb][q][0|1][n] = sum | sum of squares of out[img][.][.][n]
 * over the pixels that m block mb (of *mbi per image) and pixel-wave q (of 4) produced -- plain stores, every entry is
 * written by every launch (no zeroing, no atomics); nd_conv_winograd_stats_floats gives the size and *mbi.  ldo must
 * equal N.  nd_groupnorm_stats_from_partials folds them into the per-group statistics of the next GroupNorm, which
 * therefore needs no pass over the tensor. */
int nd_conv_winograd_stats_variant(void);
/* The same for any variant that can (conv_wino16_kernel, conv_wino4_kernel): nd_conv_winograd_stats_rows gives the rows
 * per image of chstats [NI][rows][sum | sum of squares][N] that variant writes (0: it cannot), every entry written by
 * every launch; nd_groupnorm_stats_from_partials folds them. */
int nd_conv_winograd_stats_rows(int variant, int NI, int H, int W);
int nd_conv3x3_winograd_vstats_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                    const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                                    const float* residual, int ldr, float* out, int ldo,
                                    int NI, int H, int W, int N, int flags, int variant,
                                    float* chstats, nd_stream_t stream);
int64_t nd_conv_winograd_stats_floats(int NI, int H, int W, int N, int* mbi);
int nd_conv3x3_winograd_stats_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                   const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                                   const float* residual, int ldr, float* out, int ldo,
                                   int NI, int H, int W, int N, int flags, float* pstats, nd_stream_t stream);
/* A Winograd variant's block shape: output pixels and output channels per block, threads, 32-channel sub-chunks per LDS
 * chunk and whether the A operand is prefetched (the kernel's template arguments; used by bench.py to name kernels). */
int nd_conv_winograd_variant_info(int variant, int* bm, int* bn, int* threads, int* nsub, int* apf);
/* The variant's kernel as a profiler prints it, e.g. "nd::conv_wino16_kernel<1>" ("" for a bad variant). */
const char* nd_conv_winograd_variant_name(int variant);
int64_t nd_conv_winograd_weight_floats(int N, int C);
/* Upper bound, in floats, of what a launch of `variant` (< 0: any variant) may read of the packed tensor: the chunks it
 * consumes plus its read-ahead (csrc/nd_weight_stream.h).  Always <= nd_conv_winograd_weight_floats(N, C). */
int64_t nd_conv_winograd_max_weight_read(int variant, int N, int C);
int nd_repack_conv_weight_winograd(const float* w_oihw, float* w_out, int N, int C, nd_stream_t stream);
int nd_conv3x3_winograd_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                             const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                             const float* residual, int ldr, float* out, int ldo,
                             int NI, int H, int W, int N, int flags, int variant,
                             const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream);

/* Winograd F(4x4,3x3) form of the same 3x3 stride-1 pad-1 convolution (model.py:173-177,194,209): 36 transform positions
 * per 4x4 output tile = a quarter of the direct convolution's multiplies, 0.5625 x the matrix instructions of the F(2x2,3x3)
 * form; fp32 throughout (exact-fp32 MFMA; data-side transform constants 1, 2, 4, 5, 8, weight-side 1/4 .. 1/24 folded at
 * repack in float64).  Numerically the looser of the two forms (about 5x the rounding error of F(2x2,3x3): 7e-6 per forward
 * of the 64x64 preset, 1.4e-4 after its 25-step chain against the reference's 1e-3; profiles/r05_f4_numerics_preset64.txt),
 * so callers choose it per layer (the plan tuner does; ND_WINOGRAD_F4=0 keeps F(2x2,3x3) everywhere).
 * conv_wf4_kernel: 6 waves, 256 output pixels x 48 channels per block, two blocks per CU.  Restrictions: ONE source tensor
 * (no concatenation) of whole 32-channel chunks; H and W multiples of 4, at least 12x12 (16x16-pixel blocks) or exactly 8x8
 * (four images per block); flags IN_UP2X / RES_UP2X / SILU_OUT; no fused GroupNorm.  `w` from
 * nd_repack_conv_weight_winograd_f4 (nd_conv_winograd_f4_weight_floats floats).
 * chstats | NULL: per-channel partial statistics of the output, [NI][rows][sum | sum of squares][N] fp32 with
 *   rows = nd_conv_winograd_f4_stats_rows(variant, NI, H, W) per image, every entry written by every launch (needs ldo == N);
 *   nd_groupnorm_stats_from_partials folds them.
 * splits > 1: split over K as nd_conv3x3_winograd_splitk_nhwc (same workspace size, same restrictions); chstats then come from
 *   the reduce pass, nd_conv_winograd_f4_splitk_stats_rows rows per image (one per run of 16 pixels; 0: H * W % 16 != 0 and such a
 *   launch takes no chstats). */
int nd_conv_winograd_f4_num_variants(void);
const char* nd_conv_winograd_f4_variant_name(int variant);
int nd_conv_winograd_f4_variant_info(int variant, int* bm, int* bn, int* threads);
int64_t nd_conv_winograd_f4_weight_floats(int variant, int N, int C);
int64_t nd_conv_winograd_f4_max_weight_read(int variant, int N, int C);
int nd_repack_conv_weight_winograd_f4(const float* w_oihw, float* w_out, int N, int C, int variant, nd_stream_t stream);
int nd_conv_winograd_f4_stats_rows(int variant, int NI, int H, int W);
int nd_conv_winograd_f4_splitk_stats_rows(int variant, int NI, int H, int W);
int nd_conv3x3_winograd_f4_nhwc(const float* x0, int C0, int ldx0, const float* w, const float* bias,
                                const float* rowbias, int ld_rowbias, const float* residual, int ldr, float* out, int ldo,
                                int NI, int H, int W, int N, int flags, int variant, float* chstats, int splits,
                                float* workspace, nd_stream_t stream);

/* ---- bf16 path (BASELINE configs[3], [4]): bf16 activations and weights in HBM, fp32 accumulation -------------------
 * nd_conv_bf16_nhwc: the same convolution and fused options as nd_conv_nhwc (ksize 1 | 3; two-source input, bias,
 *   rowbias, residual, ND_CONV_IN_UP2X / RES_UP2X / SILU_OUT) on v_mfma_f32_32x32x16_bf16.  x0 / x1 / residual / out
 *   are bf16 NHWC (channel counts and strides multiples of 8 = 16 bytes); bias and rowbias stay fp32; with
 *   ND_CONV_OUT_F32 `out` is float* and receives the fp32 accumulators (the UNet's last conv, feeding the fp32 sampler
 *   update).  `w` comes from nd_repack_conv_weight_bf16 (fp32 OIHW -> bf16 fragment order
 *   [c64][n tile][tap][k-step][lane][8], nd_conv_bf16_weight_elems elements).  `variant` < 0: cost model.
 *   Variants are tile shapes of three kernel families (nd_conv_bf16_variant_name); 20 and 21 are the GEMM-shaped 1x1
 *   forms on flat pixel lists (21: 128 px x 256 ch, two blocks per CU -- M % 128 == 0, N % 256 == 0, whole 64-channel
 *   chunks, plain bf16 output); a variant that cannot take a launch returns ND_E_ARG with the reason.
 *   gnA/gnB [NI][ld_gn] fp32 | NULL: GroupNorm(+AdaGN)(+SiLU with ND_CONV_GN_SILU) of the INPUT applied by the loader
 *   (coefficients from nd_groupnorm_coeffs; padding stays zero); needs one image per block (H*W >= the pixel tile).
 * nd_f32_to_bf16_rows: [rows][ldx] fp32 -> [rows][ldo] bf16, channels [C, ldo) zeroed (x_t enters the bf16 UNet).
 * nd_attention_bf16_nhwc: nd_attention_nhwc on bf16 q/k/v (both products on bf16 MFMA, softmax in fp32), bf16 out. */
int nd_conv_bf16_num_variants(void);
int nd_conv_bf16_variant_info(int variant, int* bm, int* bn, int* threads);
int64_t nd_conv_bf16_weight_elems(int N, int C, int ksize);
/* Upper bound, in elements, of what a launch of `variant` (< 0: any) with `splits` splits over K may read of the packed
 * tensor (read-ahead and split-K's shifted pointers included).  Always <= nd_conv_bf16_weight_elems(N, C, ksize). */
int64_t nd_conv_bf16_max_weight_read(int variant, int N, int C, int ksize, int splits);
/* layout 0: fragments of v_mfma_f32_32x32x16_bf16; layout 1: of v_mfma_f32_16x16x32_bf16 -- a variant takes the layout
 * nd_conv_bf16_variant_layout(variant) names (0 for variant < 0); both have nd_conv_bf16_weight_elems elements. */
int nd_conv_bf16_variant_layout(int variant);
/* 3x3 bf16 convolution that also leaves the statistics of its (bf16-rounded) output behind for the next GroupNorm
 * (model.py:190,201-207 read the tensor this conv writes): chstats rows [NI][rows][sum | sum of squares][N] in fp32, one
 * row per (pixel tile of the image, wave row), every row written by every launch (no zeroing, no atomics, fixed order).
 * rows = nd_conv_bf16_stats_rows(NI, H, W, N, variant) (0: this variant / shape cannot: LDS-DMA and 16x16x32 forms,
 * several images per block, N % 4 != 0).  nd_groupnorm_stats_from_partials folds the rows.  The variant must be named. */
int nd_conv_bf16_stats_rows(int NI, int H, int W, int N, int variant);
int nd_conv3x3_bf16_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                               const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                               const void* residual, int ldr, void* out, int ldo,
                               int NI, int H, int W, int N, int flags, int variant,
                               const float* gnA, const float* gnB, int ld_gn, float* chstats, nd_stream_t stream);
/* The same for a 1x1 convolution on the two-blocks-per-CU GEMM form (variant 21; the attention block's output projection
 * + residual feeds the next block's GroupNorm, model.py:291,190): rows = H*W / 128 per image (nd_conv_bf16_stats_rows; 0
 * unless H*W % 128 == 0 and N % 256 == 0); same argument list, rowbias / gnA / gnB must be NULL. */
int nd_conv1x1_bf16_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                               const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                               const void* residual, int ldr, void* out, int ldo,
                               int NI, int H, int W, int N, int flags, int variant,
                               const float* gnA, const float* gnB, int ld_gn, float* chstats, nd_stream_t stream);
/* The same convolution split over K for layers with too few output tiles to fill the chip (8x8 / 16x16 maps): block row s
 * of `splits` computes a range of whole input-channel chunks and leaves raw fp32 partials in `workspace` (splits * NI*H*W * N
 * floats), a second kernel adds them in split order (deterministic) with bias / rowbias / residual / SiLU and writes bf16.
 * Plain bf16 output only: no fused GroupNorm, output statistics, 2x-upsampled reads or LDS-DMA variants; the tile variant
 * must be named; fewer splits are used when the layer has fewer chunks (error if that leaves one). */
/* fp32 words of `workspace` a nd_conv_bf16_splitk_nhwc launch of this shape uses (C = C0 + C1; negative: bad arguments or
 * too few input channels for two splits). */
int64_t nd_conv_bf16_splitk_workspace_floats(int NI, int H, int W, int N, int C, int ksize, int splits);
int nd_conv_bf16_splitk_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                             const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                             const void* residual, int ldr, void* out, int ldo,
                             int NI, int H, int W, int N, int ksize, int flags, int variant, int splits,
                             float* workspace, nd_stream_t stream);
/* Kernel behind a tile variant ("nd::conv_bf16_kernel", "nd::conv_bf16s_kernel", "nd::conv_bf16w_kernel",
 * "nd::gemm_bf16_kernel"); "" for an unknown variant. */
const char* nd_conv_bf16_variant_name(int variant);
int nd_repack_conv_weight_bf16(const float* w_oihw, void* w_out, int N, int C, int ksize, int layout, nd_stream_t stream);
int nd_f32_to_bf16_rows(const float* x, int ldx, void* out, int ldo, int C, int64_t rows, nd_stream_t stream);
int nd_conv_bf16_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                      const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                      const void* residual, int ldr, void* out, int ldo,
                      int NI, int H, int W, int N, int ksize, int flags, int variant,
                      const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream);
int nd_attention_bf16_nhwc(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int T, int heads, int hd,
                           int q_off, int k_off, int v_off, int head_stride, float scale, nd_stream_t stream);

/* Direct (non-MFMA) convolution for the shapes the MFMA path does not take: 3x3 stride 2 pad 1
 * (Downsample with_conv, model.py:103-105).  Weights in PyTorch's own OIHW layout.  out is [NI, Ho, Wo, N]. */
int nd_conv_direct_nhwc(const float* x, int C, int ldx, const float* w_oihw, const float* bias,
                        float* out, int ldo, int NI, int H, int W, int N, int ksize, int stride, int pad,
                        nd_stream_t stream);

/* Weight repack, run once at load time: OIHW [N][C][k][k] (k = 1 or 3; Conv1d [N][C][1] and Linear [N][C] are the
 * k = 1 case) -> MFMA-fragment order [c32][n tile][tap][kc][lane][4], zero padded, so that every B operand of the
 * matrix instruction is one coalesced 1 KiB load.  w_out must hold nd_conv_weight_floats(N, C, ksize) floats. */
int64_t nd_conv_weight_floats(int N, int C, int ksize);
/* Upper bound, in floats, of what a launch of `variant` (< 0: any) may read of the packed tensor (read-ahead included).
 * Always <= nd_conv_weight_floats(N, C, ksize). */
int64_t nd_conv_max_weight_read(int variant, int N, int C, int ksize);
int nd_repack_conv_weight(const float* w_oihw, float* w_out, int N, int C, int ksize, nd_stream_t stream);

/* ---- K3/K4: GroupNorm(32 groups) over NHWC, input = concat(x0, x1); activations fp32 or bf16 (`dtype`) ---------
 * stats: per (img, block, group) sum and sum of squares of (x + addvec[img,c]) in float64 ->
 *   partials[NI][nblocks][G][2], nblocks = nd_groupnorm_stats_blocks(NI, HW, C, dtype); every entry is WRITTEN (nothing
 *   to zero) and bitwise reproducible: fixed-order folds inside a block, and the consumers below add the blocks'
 *   partials in block order -- no atomics at all.  addvec [NI][ld_add] | NULL is the non-adaptive
 *   timestep-embedding add that precedes out_norm (model.py:205).
 * apply: y = ((x+addvec) - mean) * rstd * gamma + beta;  if scale: y = y*(1+scale[img,c]) + shift[img,c]
 *   (model.py:201-203); optional SiLU (model.py:190,207,447); optional 2x2 average pool of the result
 *   (model.py:111 applied to h, :192).  eps as nn.GroupNorm (1e-5).  out has the input's element type.
 */
int nd_groupnorm_stats_blocks(int NI, int HW, int C, int dtype);
int nd_groupnorm_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                            const float* addvec, int ld_add, double* partials, int NI, int HW, int G,
                            int dtype, nd_stream_t stream);
/* Partial output statistics written by nd_conv3x3_winograd_stats_nhwc (p0: rows0 = mbi*4 partial rows per image over C0
 * channels; optionally concatenated with p1 over C1 channels) -> WRITES the per-group sums to stats [NI][1][G][2] (the
 * nblocks = 1 form of what nd_groupnorm_stats_nhwc fills), summed in a fixed order. */
/* Per-channel partial sums of one tensor in the layout the convolutions' epilogues leave behind: rows
 * [NI][nd_groupnorm_stats_blocks(NI, HW, C, dtype)][sum | sum of squares][C], fp32, every entry written.  A tensor's sums
 * are computed once and re-grouped by every GroupNorm that reads it (alone or concatenated: model.py:474,190). */
int nd_groupnorm_channel_partials_nhwc(const void* x, int C, int ldx, float* rows, int NI, int HW, int dtype,
                                       nd_stream_t stream);
int nd_groupnorm_stats_from_partials(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1,
                                     double* stats, int NI, int G, nd_stream_t stream);
/* Statistics + apply in ONE launch for small tensors (launch-bound plans, e.g. the EMNIST preset at batch 4: replaces
 * nd_groupnorm_channel_partials_nhwc + nd_groupnorm_stats_from_partials + nd_groupnorm_apply_nhwc; reference
 * model.py:190,201-207,264,446-447 as above): one block per (group, image), float64 sums in a fixed order, the same
 * coefficient arithmetic and flags (ND_GN_SILU, ND_GN_POOL2) as nd_groupnorm_apply_nhwc.  Channels per group <= 64. */
int nd_groupnorm_fused_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                            const float* gamma, const float* beta, const float* scale, const float* shift, int ld_ss,
                            void* out, int ldo, int NI, int H, int W, int G, float eps, int flags, int dtype,
                            nd_stream_t stream);
/* The same affine as nd_groupnorm_apply_nhwc as per-(image, channel) coefficients y = x*A + B, for convolutions
 * that apply it while loading their input (gnA/gnB of nd_conv_nhwc / nd_conv3x3_winograd_nhwc). */
int nd_groupnorm_coeffs(const double* partials, int nblocks, const float* gamma, const float* beta, const float* scale,
                        const float* shift, int ld_ss, float* coefA, float* coefB, int ld_coef,
                        int NI, int C, int HW, int G, float eps, nd_stream_t stream);
/* nd_groupnorm_stats_from_partials + nd_groupnorm_coeffs in one launch (same order of additions, same arithmetic: the
 * coefficients are bit-identical to the two-step form), for norms whose only consumer is a convolution's loader. */
int nd_groupnorm_coeffs_from_partials(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1,
                                      const float* gamma, const float* beta, const float* scale, const float* shift,
                                      int ld_ss, float* coefA, float* coefB, int ld_coef, int NI, int HW, int G,
                                      float eps, nd_stream_t stream);
/* The apply pass on READY per-(image, channel) coefficients (nd_groupnorm_coeffs / nd_groupnorm_coeffs_from_partials in front
 * of it): out = act(x * coefA[img][c] + coefB[img][c]), the same arithmetic and bits as nd_groupnorm_apply_nhwc without its
 * per-block fold.  (C0 + C1) / V <= 256 and C0 % V == 0 (V = 4 fp32, 8 bf16 elements per 16 bytes); flags: ND_GN_SILU only. */
int nd_groupnorm_apply_coeffs_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                   const float* coefA, const float* coefB, int ld_coef, void* out, int ldo, int NI, int HW,
                                   int flags, int dtype, nd_stream_t stream);
int nd_groupnorm_apply_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                            const float* addvec, int ld_add, const double* partials, int nblocks,
                            const float* gamma, const float* beta,
                            const float* scale, const float* shift, int ld_ss,
                            void* out, int ldo, int NI, int H, int W, int G, float eps, int flags, int dtype,
                            nd_stream_t stream);

/* ---- K7: attention core softmax(q k^T * scale) v over T tokens (model.py:266-287) ---------------------------
 * qkv is [B*T][ld_qkv]; head h reads q at column q_off + h*head_stride + d, k at k_off + ..., v at v_off + ...
 *   split_qkv_first=True : q_off=0, k_off=C,  v_off=2C,  head_stride=hd      (model.py:268-270)
 *   split_qkv_first=False: q_off=0, k_off=hd, v_off=2hd, head_stride=3*hd    (model.py:279-280)
 * out is [B*T][ld_out] with column h*hd + d.  hd must be a multiple of 8 and <= 256.
 */
int nd_attention_nhwc(const float* qkv, int ld_qkv, float* out, int ld_out, int B, int T, int heads, int hd,
                      int q_off, int k_off, int v_off, int head_stride, float scale, nd_stream_t stream);

/* ---- K8: standalone 2x resampling (Upsample/Downsample without conv, model.py:77,111; x path of :193) ------- */
int nd_upsample2x_nhwc(const float* x, int ldx, float* out, int ldo, int NI, int H, int W, int C, nd_stream_t stream);
/* Space-to-depth by 2 (even H, W): out[img][y][x][(p*2+q)*C + c] = x[img][2y+p][2x+q][c], out is [NI][H/2][W/2][4C].
 * The stride-2 3x3 Downsample convolution (model.py:103-108) is then the stride-1 3x3 convolution of `out` with the
 * weights w'[n][(p,q,c)][u][v] = w[n][c][dy][dx], (dy -> p,u): 0 -> (1,0), 1 -> (0,1), 2 -> (1,1) (same for dx -> q,v), zero
 * elsewhere -- i.e. it runs on nd_conv_nhwc / nd_conv3x3_winograd_nhwc. */
int nd_space_to_depth2_nhwc(const float* x, int ldx, float* out, int ldo, int NI, int H, int W, int C, nd_stream_t stream);
/* 2x2 average pool; fp32 or bf16 elements (`dtype`).  nd_upsample2x_nhwc / nd_space_to_depth2_nhwc only move 16-byte
 * groups of channels, so bf16 tensors go through them as fp32 tensors of C/2 channels and ld/2 strides. */
int nd_avgpool2x_nhwc(const void* x, int ldx, void* out, int ldo, int NI, int H, int W, int C, int dtype, nd_stream_t stream);

/* ---- layout at the API edge: NCHW [NI][C][HW] <-> NHWC [NI][HW][ld] (pad channels written as 0) ------------- */
int nd_nchw_to_nhwc(const float* src, float* dst, int NI, int C, int HW, int ld, nd_stream_t stream);
int nd_nhwc_to_nchw(const float* src, float* dst, int NI, int C, int HW, int ld, nd_stream_t stream);

/* ---- K10: sampler updates (device-side step index so the step body is hipGraph-capturable) ------------------
 * step: device int32, the current rescaled index t.  coef: fp32 [S][ND_COEF_COLS], row t =
 *   { sqrt(1/abar_t), sqrt(1/abar_t - 1), abar_t, abar_{t-1}, posterior_mean_coef_x0, posterior_mean_coef_xt,
 *     logvar_a, logvar_b }  (float64 tables of diffusion.py:115-130 cast to fp32 as extract() does, :492).
 * x, x_out: NHWC [B][HW][ldx] (C channels used); eps / eps_uncond: model output NHWC [B][HW][ld_eps], channels
 *   [0,C) = epsilon, [C,2C) = variance head.  eps_uncond != NULL applies classifier-free guidance
 *   eps = (1+w)*eps - w*eps_uncond (diffusion.py:284,347).
 * noise: NHWC like x, indexed noise + t*noise_step_stride (floats) | NULL.  If NULL and noise is needed
 *   (eta != 0 / DDPM) it is drawn in-kernel from Philox4x32-10 keyed by (seed, t, first_elem + element), element =
 *   (b*HW + p)*C + c: a rank that denoises rows [r0, r1) of a larger batch passes first_elem = r0*HW*C and draws
 *   exactly the numbers a single process would have drawn for those rows.  seed_dev | NULL: device word holding the
 *   seed; when given it replaces `seed`, so a captured graph of the step serves every seed (the host rewrites the word).
 */
int nd_fill_timestep(const int64_t* timestep_map, const int32_t* step, int64_t* t_out, int B, nd_stream_t stream);
int nd_step_advance(int32_t* step, int delta, nd_stream_t stream);
/* out[0 .. row_floats) = table[*step - lo][0 .. row_floats) for a [rows][row_floats] fp32 table: the step body picks this
 * step's precomputed embedding rows (K1/K2 of every step of a chain evaluated once, before the loop; model.py:197,348-352
 * depend on t and y only) by the device step word, so the captured graph needs no timestep MLP.  row_floats % 4 == 0,
 * 16-byte aligned pointers; a step word outside [lo, lo + rows) reads the nearest row (never outside the table). */
int nd_copy_row_by_step(const float* table, const int32_t* step, int lo, int rows, int64_t row_floats, float* out,
                        nd_stream_t stream);
/* x_dup | NULL: second destination of the updated images (classifier-free guidance: the unconditional half of the next
 * forward's input batch, diffusion.py:281,344 evaluate the model on the same x_t twice); not x, not x_out.
 * pred_x0 | NULL: receives the step's x_0 estimate (diffusion.py:287-290 / :350-353) -- the second element of the
 *   (sample, pred_x0) tuple Diffusion.denoising_step / ddim_denoising_step return -- laid out like x_out (ldx).
 * flags: ND_STEP_NO_CLIP = the reference's clip_x=False (no clamp of pred_x0 to [-1, 1]); ND_STEP_PER_IMAGE = `step` points
 *   to B words, one rescaled step index per image (the reference's `t` argument is a [B] tensor), instead of one word. */
#define ND_STEP_NO_CLIP 1
#define ND_STEP_PER_IMAGE 2
int nd_ddim_step(const float* x, float* x_out, float* x_dup, float* pred_x0, int flags, int ldx, const float* eps,
                 const float* eps_uncond, int ld_eps, float guidance_w, const float* coef, const int32_t* step, float eta,
                 const float* noise, int64_t noise_step_stride, uint64_t seed, const uint64_t* seed_dev,
                 uint64_t first_elem, int B, int HW, int C, nd_stream_t stream);
int nd_ddpm_step(const float* x, float* x_out, float* x_dup, float* pred_x0, int flags, int ldx, const float* eps,
                 const float* eps_uncond, int ld_eps, float guidance_w, const float* coef, const int32_t* step, int var_kind,
                 const float* noise, int64_t noise_step_stride, uint64_t seed, const uint64_t* seed_dev,
                 uint64_t first_elem, int B, int HW, int C, nd_stream_t stream);
/* Diffusion.get_eps_and_log_var after the model call (diffusion.py:248-264): NHWC model output [B][HW][ld_out] -> eps and
 * log_var, both NCHW [B][C][HW]; steps = B rescaled step indices (device), coef / var_kind as for nd_ddpm_step. */
int nd_eps_log_var(const float* model_out, int ld_out, const float* coef, const int32_t* steps, int var_kind, float* eps,
                   float* log_var, int B, int HW, int C, nd_stream_t stream);
/* forward diffusion q(x_t | x_0) = sqrt(abar_t) x0 + sqrt(1-abar_t) noise (diffusion.py:232-240) */
int nd_qsample(const float* x0, const float* noise, float* out, int64_t n, float sqrt_ab, float sqrt_1mab,
               nd_stream_t stream);
/* the same with one rescaled step index per image (Diffusion.diffusion_step takes a [B] tensor t): sqrt_ab / sqrt_1mab are
 * device fp32 tables [S], steps B device words, per_image = C*H*W */
int nd_qsample_steps(const float* x0, const float* noise, float* out, int B, int64_t per_image, const float* sqrt_ab,
                     const float* sqrt_1mab, const int32_t* steps, nd_stream_t stream);

/* ---- N2: image post-processing ((x+1)*127.5).clamp(0,255) -> uint8 (truncation), NHWC -> HWC bytes,
 * optional grayscale inversion 255-v (scripts/sample.py:94-100,164-171). */
int nd_to_uint8_hwc(const float* x, int ldx, uint8_t* out, int NI, int HW, int C, int invert, nd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ND_HIP_H */
