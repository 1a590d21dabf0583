/*
 * masklab_hip.h -- C ABI of libmasklab_hip.so (gfx950 / MI355X).
 *
 * The reference (craftsangjae/instance-segmentation-road-project) is pure Python on
 * tensorflow.keras and has NO FFI of its own: every arithmetic step of
 * its inference path is a TensorFlow op.  Each entry point below therefore replaces
 * the TF op(s) behind the cited reference call site (paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (the library never
 *     allocates persistent memory; scratch is a caller-provided workspace);
 *   - tensors are NHWC float32 unless stated; "cstride" = channels of the buffer a
 *     tensor view lives in, "coff" = first channel of the view (lets a kernel read or
 *     write a slice of a concat buffer without a copy);
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work;
 *   - return value: 0 = ok, negative = ML_E_* (no exceptions cross the boundary);
 *   - thread-safe per stream; the only global state is the per-thread error string, per
 *     kernel an atomic bitmask of the devices whose dynamic-LDS limit has been raised
 *     (hipFuncSetAttribute is a per-device setting; racing threads set the same value), and
 *     a per-device cache of the compute-unit count (sizes the persistent kernels' grids;
 *     never changes what is computed).
 */
#ifndef MASKLAB_HIP_H
#define MASKLAB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ML_OK 0
#define ML_E_BADARG (-1)   /* shape / alignment / range precondition violated */
#define ML_E_LAUNCH (-2)   /* hipLaunchKernel or attribute call failed        */
#define ML_E_NOGPU  (-3)   /* no gfx950 device visible                        */

enum { ML_ACT_NONE = 0, ML_ACT_RELU = 1, ML_ACT_RELU6 = 2, ML_ACT_SIGMOID = 3 };
enum { ML_MATH_F32 = 0, ML_MATH_F16 = 1, ML_MATH_F16S = 2, ML_MATH_F32X3 = 3 };

#define ML_ABI_VERSION 7              /* 2: ml_conv2d_desc gained `math` / `reserved0`
                                         3: detection gather payload, mask_distribute level_max,
                                            fp16 tensor storage
                                         4: fp16 storage in the heads: ML_MATH_F16S on the generic conv,
                                            ml_gn_desc.dtype, the *_f16 entry points of GroupNorm, resize,
                                            depthwise conv, global mean, RoI crop and the mask-head tail
                                         5: fixed-capacity RoI batches (`live`) in conv / GroupNorm / RoI crop /
                                            mask-head tail descriptors, ml_mold_levels_f32
                                         6: ml_conv2d_launch_splits, ml_conv2d_gn_min_launch_tiles (reporting /
                                            the size rule of gn_partials asked of the library, not restated by callers)
                                         7: ml_stem7x7s2_pool_f16 / _f32 / _x3; ml_gconv3x3_f16 takes groups of 32 channels;
                                            ML_MATH_F32X3 on the persistent 1x1 kernel (ml_conv2d_uses_pipe);
                                            ml_mold_levels_dev_f32                                                    */
int ml_version(void);                 /* returns ML_ABI_VERSION of the library that was built */
const char *ml_last_error(void);      /* text of the last failure on the calling thread   */
int ml_device_check(void);            /* ML_OK iff device 0.. current is gfx950           */

/* ---------------------------------------------------------------- convolution (MFMA)
 * Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32: M = B*Ho*Wo output pixels,
 * N = cout, K = taps * span_pad.  Replaces tf.keras Conv2D (+folded BatchNormalization,
 * +Add, +activation), the ResNeXt grouped 3x3 and Conv2DTranspose(2,2,s2):
 *   engine/backbone/ResNext.py:200-231,343-349   engine/backbone/base.py:294-312
 *   engine/layers/detection.py:42-48,56,64,120-128,190-200
 *   engine/layers/instance.py:188-199            engine/layers/semantic.py:66,112,126,133,199,213,219
 * Weights are pre-packed by the host as wgt[n_pad][taps*span_pad] (k contiguous,
 * k = tap*span_pad + c, zero padded; see masklab_hip/packing.py).
 */
typedef struct ml_conv2d_desc {
    const float *in;        /* [B,H,W,in_cstride] view starting at channel in_coff            */
    const float *wgt;       /* packed weights, n_pad rows of ktot floats                      */
    const float *bias;      /* [cout] or NULL                                                 */
    const float *residual;  /* optional tensor added before the activation, or NULL          */
    float *out;             /* [B,Ho,Wo,out_cstride] view starting at channel out_coff        */
    int32_t B, H, W;        /* input batch / height / width                                   */
    int32_t in_cstride, in_coff;
    int32_t span;           /* K floats per tap actually read (cin; 32 for the NHWC4 stem)    */
    int32_t span_pad;       /* span rounded up to 32                                          */
    int32_t cpp_shift;      /* stem: log2(floats per pixel) so a tap spans pixels; else 30    */
    int32_t Ho, Wo;
    int32_t KH, KW, stride, dil, pad_t, pad_l;
    int32_t cout;           /* real N                                                         */
    int32_t n_pad;          /* rows in wgt (multiple of the N tile)                           */
    int32_t out_cstride, out_coff;
    int32_t res_cstride, res_coff;
    int32_t act;            /* ML_ACT_*                                                       */
    int32_t group_cin_step; /* grouped 3x3: input-channel offset per 32-wide N block; else 0  */
    int32_t shuffle2x2;     /* 1: Conv2DTranspose epilogue, column = (a*2+b)*cout_real + o    */
    int32_t tile;           /* 0 auto, 1 = 128x128, 2 = 128x64, 3 = 128x32, 4 = pipelined 1x1 (128x128), 5 = half 1x1 on 256x256 tiles */
    int32_t math;           /* ML_MATH_F32: v_mfma_f32_32x32x2_f32 (exact fp32 products);
                               ML_MATH_F16: operands rounded to fp16 on their way into LDS,
                               v_mfma_f32_32x32x16_f16 with fp32 accumulation (BASELINE config 5);
                               tensors in HBM stay fp32 either way;
                               ML_MATH_F16S: fp16 STORAGE -- in / wgt / residual point to IEEE half data
                               (element counts and strides unchanged; span_pad = span rounded up to 64,
                               span / in_cstride / in_coff multiples of 8), fp16 MFMA, fp32 accumulation,
                               bias (fp32) + residual + activation in fp32, one rounding at the store.
                               1x1 stride-1 problems with cout % 128 == 0 and span % 64 == 0 run on the
                               persistent kernel (conv1x1_pipe.hip: half output, optional half residual);
                               every other shape on the generic kernel (no residual; output half or fp32
                               by `out_f16`);
                               ML_MATH_F32X3: fp32 tensors, fp32-grade arithmetic on the f16 matrix pipe: each
                               operand x = hi + 2^-11 lo (two halves), each product hi hi + 2^-11 (hi lo + lo hi)
                               with fp32 accumulation (operands to 2^-22 relative for 2^-14 <= |x| < 65520, 2^-36
                               absolute below; the dropped term is <= 2^-22 |a b|).
                               Activations are split inside the kernel (|x| < 65520, else Inf / NaN); `wgt` points to
                               weights split BEFOREHAND: the fp32 packing with every 32-float chunk of a row
                               replaced, in place, by 32 halves hi(w) followed by 32 halves 2^11 (w - hi(w))
                               (same bytes, same strides; masklab_hip/ops.py DeviceConv.wgt_x3).  Generic
                               kernel only (every shape, residual, split-K, gn_partials as ML_MATH_F32)   */
    int32_t out_f16;        /* 1 = `out` is IEEE half: ML_MATH_F16 (the stem feeding an fp16-storage body)
                               and ML_MATH_F16S on the generic kernel (0 there = fp32 `out`: the prediction
                               tensors); dense fast epilogue only (no residual / shuffle2x2 / out_bstride /
                               sigmoid, cout % 4 == 0).  The persistent kernel always writes half.      */
    int64_t out_bstride;    /* floats between images in `out`; 0 = Ho*Wo*out_cstride (dense).
                               Lets a level's head write straight into the concatenated
                               [B, A, classes] prediction (detection.py:210-212 Reshape+Concatenate) */
    const int32_t *live;    /* NULL, or a DEVICE int: the batch is a fixed-capacity RoI batch (the mask head run
                               without a host read of the RoI counts, instance.py:121-134 + MoldBatch misc.py:231-286)
                               in which image i exists iff i % live_period < max(1, *live); tiles that hold only
                               non-existing images compute and store nothing.  Generic kernel only.        */
    int32_t live_period;    /* RoI slots per image (B % live_period == 0); ignored when live == NULL      */
    int32_t reserved1;
    double *gn_partials;    /* NULL, or [ceil(M/128)][4][2] DEVICE doubles: the epilogue also writes (sum, sum of squares)
                               of the values each 128-row tile stores, one pair per wave of the block (4 per tile) --
                               what the GroupNormalization behind a head conv
                               needs (detection.py:120-125): ml_gn_desc.partials then replaces the statistics pass.
                               fp32, cout = n_pad = 128, M % 128 == 0, dense destination, no residual, and a launch
                               that is neither narrowed nor split along K (>= 257 tiles in all)                 */
} ml_conv2d_desc;

int ml_conv2d_f32(const ml_conv2d_desc *d, void *stream);

/* Several independent conv problems of the SAME tile shape in one launch (the un-shared head
 * towers run the same conv at every pyramid level: detection.py:109-130,179-202, instance.py:177-201).
 * With a workspace, a launch with few output tiles in total and a long K is split along K
 * (partials in the workspace, fixed-order reduction: deterministic).  workspace may be NULL.   */
#define ML_CONV_MAX_PROBLEMS 12
int64_t ml_conv2d_workspace_bytes(void);
int ml_conv2d_multi_f32(const ml_conv2d_desc *descs, int32_t n, void *workspace, int64_t workspace_bytes,
                        void *stream);
/* N-tile width the auto heuristic picks for `cout` (host packs n_pad from it). */
int ml_conv2d_ntile(int32_t cout, int32_t tile);
/* N-tile width (128 / 64 / 32) of the generic implicit-GEMM kernel that ml_conv2d_multi_f32 will run for these
 * problems: launches too small to fill the chip with 128-wide tiles run on narrower ones (bit-identical results:
 * same k-ordered chains, split-K cut at the same k).  For reporting only; 0 on bad arguments. */
int ml_conv2d_launch_ntile(const ml_conv2d_desc *descs, int32_t n, int32_t has_workspace);
/* M-tile height (128 / 256) of the same launch: ML_MATH_F32X3 launches that fill the chip with 256 x 128 tiles run the
 * 8-wave software-pipelined form of the kernel (bit-identical results).  For reporting only; 0 on bad arguments. */
int ml_conv2d_launch_mtile(const ml_conv2d_desc *descs, int32_t n, int32_t has_workspace);
/* Which persistent 1x1 kernel ml_conv2d_multi_f32 runs this single problem on: 1 = the tile-pipelined 128 x 128 kernel
 * (conv1x1_pipe.hip: the short-K bottleneck convs of engine/backbone/ResNext.py:199-231; fp32 or half tensors),
 * 2 = the 256 x 256-tile kernel for half tensors with K >= 256 (conv1x1_h256.hip: the ResNeXt-101 stage 2-4 convs of
 * BASELINE configs[4]), 0 = neither (the generic implicit-GEMM kernel).  `tile` = 4 / 5 force 1 / 2 where they apply. */
int ml_conv2d_uses_pipe(const ml_conv2d_desc *d);
/* K slices (1 = not split) of every problem of the launch ml_conv2d_multi_f32 would make for these problems with a
 * workspace of `workspace_bytes` (0 = none): launches of fewer than 192 tiles with a long K are cut along K, so a shard
 * of a batch may sum K in other pieces than the whole batch does (fp32 rounding; reference DP merge
 * engine/parallel.py:64-107).  splits: n host ints.  For reporting / tests. */
int ml_conv2d_launch_splits(const ml_conv2d_desc *descs, int32_t n, int64_t workspace_bytes, int32_t *splits);
/* Smallest launch, in 128 x 128 tiles over all its problems, that ml_conv2d_multi_f32 neither narrows to 128 x 64 / 128 x 32
 * tiles nor cuts along K on the current device: the size from which ml_conv2d_desc.gn_partials may be set. */
int64_t ml_conv2d_gn_min_launch_tiles(void);

/* ResNeXt grouped 3x3 (reference engine/backbone/ResNext.py:212-219: DepthwiseConv2D(depth_multiplier=c)
 * + SplitGroups/ReduceGroups/MergeGroups), c = channels per group in {4,8,16}, C % 64 == 0, on
 * v_mfma_f32_4x4x1 (16 independent 4x4 blocks per instruction: no block-diagonal padding waste).
 * wgt is [C][9][c]: wgt[g*c+m][tap][i] = K[tap][g*c+i][m] (BatchNorm scale folded), bias [C] or NULL. */
int ml_gconv3x3_f32(const float *in, const float *wgt, const float *bias, float *out,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t c, int32_t Ho, int32_t Wo,
                    int32_t stride, int32_t pad_t, int32_t pad_l, int32_t act, void *stream);

/* The same on fp16 tensors (in / out IEEE half, weights / bias / arithmetic fp32): the grouped conv of an
 * fp16-STORAGE ResNeXt body (BASELINE config 5).  Also c = 32 (the last stage, filters = 1024: two
 * v_mfma_f32_32x32x16_f16 per tap contract a group).                                                 */
int ml_gconv3x3_f16(const void *in, const float *wgt, const float *bias, void *out,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t c, int32_t Ho, int32_t Wo,
                    int32_t stride, int32_t pad_t, int32_t pad_l, int32_t act, void *stream);

/* ---------------------------------------------------------------- fp16-storage helpers (BASELINE config 5)
 * ZeroPadding2D(1)+MaxPooling2D(3,2) on an fp16 map (ResNext.py:351-352; resnext.py:196-197), C % 8 == 0. */
int ml_maxpool3x3s2_f16(const void *in, void *out, int32_t B, int32_t H, int32_t W, int32_t C,
                        int32_t Ho, int32_t Wo, int32_t pad_t, int32_t pad_l, void *stream);
/* out[b,oy,ox,:] = in[b,2oy,2ox,:] on fp16 [B,H,W,C] -> [B,ceil(H/2),ceil(W/2),C]: the sampling of a 1x1
 * stride-2 conv (the strided shortcuts, ResNext.py:199-203), which then runs as a stride-1 ML_MATH_F16S conv. */
int ml_subsample2_f16(const void *in, void *out, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
/* n halves -> n floats / n floats -> n halves (tensors crossing between an fp16-storage and an fp32 part), n % 8 == 0 */
int ml_cast_f16_to_f32(const void *in, float *out, int64_t n, void *stream);
int ml_cast_f32_to_f16(const float *in, void *out, int64_t n, void *stream);
/* The heads on IEEE-half tensors (fp16 storage beyond the backbone body, BASELINE configs[4]): the arithmetic of
 * ml_resize_bilinear_ac_f32 (engine/layers/misc.py:306 + the FPN Add detection.py:58-60 / concat-slice store
 * semantic.py:154,227), ml_dwconv3x3_f32 (semantic.py:63-64) and ml_global_mean_f32 (semantic.py:149) on half
 * in / add / out (weights and bias fp32), computed in fp32 with one rounding at the store; channel counts, strides
 * and offsets multiples of 8.                                                                                    */
int ml_resize_bilinear_ac_f16(const void *in, const void *add, void *out,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                              int32_t Ho, int32_t Wo, int32_t add_cstride, int32_t add_coff,
                              int32_t out_cstride, int32_t out_coff, void *stream);
int ml_dwconv3x3_f16(const void *in, const float *wgt, const float *bias, void *out,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                     int32_t out_cstride, int32_t out_coff, int32_t Ho, int32_t Wo,
                     int32_t stride, int32_t dil, int32_t pad_t, int32_t pad_l, int32_t act, void *stream);
int ml_global_mean_f16(const void *in, void *out, int32_t B, int32_t HW, int32_t C, void *stream);

/* The ResNeXt stem of the fp16-storage mode in one pass (engine/backbone/ResNext.py:343-352; thirdparty/classification_models/
 * models/resnext.py:193-197): ZeroPadding2D(3) + Conv2D(64, 7x7, stride 2, BatchNorm folded) + ReLU + ZeroPadding2D(1) +
 * MaxPooling2D(3, 2).  image: fp32 NHWC4 [B,H,W,4] (ml_preprocess_f32 with out_c = 4); wgt_h: IEEE half [64][7][8][4] =
 * the row-span packing of the stem (k = kernel row * 32 + pixel * 4 + channel, 8th pixel / 4th channel zero) rounded to
 * half; bias fp32 [64] or NULL; out: IEEE half [B,Hp,Wp,64].  Operands rounded to half, fp32 accumulation from the bias,
 * one rounding: bit-identical to ml_conv2d_f32 (ML_MATH_F16, out_f16) followed by ml_maxpool3x3s2_f16, without the
 * un-pooled map (839 MB at 16 x 1280^2) ever reaching memory.                                                     */
int ml_stem7x7s2_pool_f16(const float *image, const void *wgt_h, const float *bias, void *out, int32_t B, int32_t H,
                          int32_t W, int32_t Hp, int32_t Wp, void *stream);
/* The same stem on fp32 tensors with exact fp32 products (the default math): wgt = the fp32 row-span packing itself
 * ([64][7 x 32]), out fp32 [B,Hp,Wp,64].  Only the products with a non-zero weight are issued (12 of the generic kernel's
 * 16 MFMAs per kernel row), in the generic kernel's pairs and order: bit-identical to ml_conv2d_f32 (ML_MATH_F32, ReLU)
 * followed by ml_maxpool3x3s2_f32, without the un-pooled map (537 MB at 8 x 1024^2) ever reaching memory.            */
int ml_stem7x7s2_pool_f32(const float *image, const float *wgt, const float *bias, float *out, int32_t B, int32_t H,
                          int32_t W, int32_t Hp, int32_t Wp, void *stream);
/* ... and with ML_MATH_F32X3 products: wgt_x3 = the split row-span packing (per kernel row 32 hi halves, then 32 halves
 * 2^11 (w - hi)); the image values are split once, on their way into LDS.  Same steps, products and order as the generic
 * kernel's X3 path: bit-identical to ml_conv2d_f32 (ML_MATH_F32X3, ReLU) followed by ml_maxpool3x3s2_f32.            */
int ml_stem7x7s2_pool_x3(const float *image, const void *wgt_x3, const float *bias, float *out, int32_t B, int32_t H,
                         int32_t W, int32_t Hp, int32_t Wp, void *stream);

/* ---------------------------------------------------------------- fused mask-head tail
 * Conv2DTranspose(C_mid, (2,2), (2,2)) + bias + act_mid followed by Conv2D(ncls, (1,1)) + bias + act_out in one
 * kernel (reference engine/layers/instance.py:196-201 constructs the pair, :226-233 calls it; replaces the
 * tf.nn.conv2d_transpose -> tf.nn.conv2d pair).  Up to 4 problems (RoI levels, each with its own weights) per launch.
 *   x        [M, K] fp32, M = rois * hw input pixels (whole h x w maps, RoIs grouped per image: rois = B * rois_per_image)
 *   wd       [4][C_mid][K]: position q = dy*2+dx major, k contiguous (the ml_conv2d_desc packing of the transposed conv)
 *   bd       [C_mid] or NULL;   bo [ncls] or NULL
 *   wo_table [C_mid/32][16][2][cp]: entry (t, e, half, c) = W_out[32 t + (e & 3) + 8 (e >> 2) + 4 half][c] (0 for
 *            c >= ncls) -- the 1x1 kernel's rows in the order the matrix cores' accumulator registers hold channels;
 *            cp = power of two >= ncls
 *   out      element (img, j, 2y+dy, 2x+dx, c) of RoI j of image img at
 *            out[img * out_image_stride + out_base + j * (4 hw ncls) + ((2y+dy) * 2w + 2x+dx) * ncls + c]
 * K % 32 == 0, C_mid in {128, 256}, ncls <= 32, M < 2^24 per problem. */
typedef struct ml_deconv_out_problem {
    const float *x, *wd, *bd, *wo_table, *bo;
    float *out;
    int64_t M;
    int32_t hw, w, rois_per_image, reserved0;
    int64_t out_image_stride, out_base;
    const int32_t *live;   /* NULL, or a device int: RoI slot j of an image exists iff j < max(1, *live) (fixed-capacity
                              batch with rois_per_image = the capacity); tiles of non-existing slots are skipped */
} ml_deconv_out_problem;
int ml_deconv2x2_out1x1_f32(const ml_deconv_out_problem *probs, int32_t nprob, int32_t K, int32_t c_mid, int32_t ncls,
                            int32_t cp, int32_t act_mid, int32_t act_out, void *stream);
/* The same with `x` and `wd` in IEEE half (the fp16-storage mask head; K % 64 == 0): the transposed conv runs on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation; bias, the 1x1 conv (fp32 table), sigmoid and `out` stay fp32. */
int ml_deconv2x2_out1x1_f16(const ml_deconv_out_problem *probs, int32_t nprob, int32_t K, int32_t c_mid, int32_t ncls,
                            int32_t cp, int32_t act_mid, int32_t act_out, void *stream);

/* ---------------------------------------------------------------- depthwise / pooling
 * 3x3 DepthwiseConv2D (depth_multiplier 1), stride 1/2, dilation, explicit pads, +bias
 * (folded BN) +activation.  tf.keras.applications MobileNet body; semantic.py:63; misc.py:85.
 * wgt is [9][C] (tap major).                                                              */
int ml_dwconv3x3_f32(const float *in, const float *wgt, const float *bias, float *out,
                     int32_t B, int32_t H, int32_t W, int32_t C,
                     int32_t in_cstride, int32_t in_coff, int32_t out_cstride, int32_t out_coff,
                     int32_t Ho, int32_t Wo, int32_t stride, int32_t dil,
                     int32_t pad_t, int32_t pad_l, int32_t act, void *stream);

/* ZeroPadding2D(1)+MaxPooling2D(3,2) on a non-negative (post-ReLU) map: ResNext.py:351-352 */
int ml_maxpool3x3s2_f32(const float *in, float *out, int32_t B, int32_t H, int32_t W, int32_t C,
                        int32_t Ho, int32_t Wo, int32_t pad_t, int32_t pad_l, void *stream);

/* BackBonePreProcess (engine/backbone/base.py:57-75) fused with the NHWC->NHWC4 repack the
 * MFMA stem wants: out[...,k] = (in[..., flip?2-k:k] - mean[k]) / div[k] + shift[k], out[...,3]=0
 * (per-channel div covers normalize=3 :71-73 and the ResNeXt-101 `bn_data` input BatchNorm,
 * thirdparty/classification_models/models/resnext.py:194, which precedes the zero padding).
 * in is uint8 (is_u8=1) or float32 RGB [B,H,W,3]; out_c is 3 or 4; mean/div/shift: 3 host floats. */
int ml_preprocess_f32(const void *in, int32_t is_u8, float *out, int64_t npix, int32_t out_c,
                      int32_t flip, const float *mean, const float *div, const float *shift,
                      void *stream);

/* ---------------------------------------------------------------- GroupNormalization
 * The reference's chunk-wise GroupNormalization (engine/normalization.py:116-160):
 * per sample, the flat H*W*C vector is cut into G contiguous chunks; y=(x-mean_g)/sqrt(var_g+eps)
 * * gamma[j] + beta[j], j = g*(C/G) + (c mod C/G).  Optional fused ReLU (semantic.py:72-73,
 * 116,137,202).  workspace: >= ml_groupnorm_workspace_bytes() bytes.  In-place (y==x) allowed. */
int64_t ml_groupnorm_workspace_bytes(int32_t N, int32_t G);
int ml_groupnorm_chunk_f32(const float *x, float *y, const float *gamma, const float *beta,
                           int32_t N, int64_t HWC, int32_t C, int32_t G, float eps, int32_t relu,
                           int32_t out_cstride, int32_t out_coff, /* y view: channels of the buffer / first channel; (C,0) = dense */
                           void *workspace, void *stream);
/* The same on IEEE-half tensors (x, y half; gamma / beta float): the heads of the fp16 path (BASELINE configs[4]).
 * Statistics are fp64 sums of the stored values, the normalisation runs in fp32, one rounding at the store. */
int ml_groupnorm_chunk_f16(const void *x, void *y, const float *gamma, const float *beta,
                           int32_t N, int64_t HWC, int32_t C, int32_t G, float eps, int32_t relu,
                           int32_t out_cstride, int32_t out_coff, void *workspace, void *stream);

/* Several independent GroupNormalizations in one launch pair (the five pyramid levels of a tower depth,
 * detection.py:124,194; the three RoI levels of the mask head, instance.py:192): the few-thousand-float problems of
 * the coarse levels ride along instead of being ~10 us launches of their own.  Same arithmetic per problem.
 * Every problem: 16-byte aligned tensors, HWC/G and C multiples of 4.  workspace_bytes >= the sum over the problems of
 * ml_groupnorm_workspace_bytes(N, G).                                                                          */
typedef struct ml_gn_desc {
    const float *x;
    float *y;
    const float *gamma, *beta;     /* [C] or NULL                                   */
    int64_t HWC;                   /* floats per sample                             */
    int32_t N, C, G, relu;
    int32_t out_cstride, out_coff; /* y view, as in ml_groupnorm_chunk_f32          */
    float eps;
    int32_t dtype;                 /* 0: x / y are float; 1: x / y point to IEEE half (fp16-storage heads;
                                      HWC/G and C multiples of 8); gamma / beta are float either way */
    const int32_t *live;           /* NULL, or a device int: sample n exists iff n % live_period < max(1, *live)
                                      (fixed-capacity RoI batches, as ml_conv2d_desc.live); others are skipped   */
    int32_t live_period, reserved;
    const double *partials;        /* NULL, or per chunk (n * G + g) `n_partials` consecutive (sum, sum of squares) pairs
                                      written by the producing conv (ml_conv2d_desc.gn_partials: the chunk's 128-row
                                      tiles, HWC/G a multiple of 128 * C): added in that order, no statistics pass    */
    int32_t n_partials, reserved2;
} ml_gn_desc;
#define ML_GN_MAX_PROBLEMS 8
int ml_groupnorm_multi_f32(const ml_gn_desc *descs, int32_t n, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- resampling / reductions
 * tf.compat.v1.image.resize_bilinear(align_corners=True) (engine/layers/misc.py:306), with the
 * FPN `Add` (detection.py:58-60) or a concat-slice write (semantic.py:154,227) fused.        */
int ml_resize_bilinear_ac_f32(const float *in, const float *add, float *out,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                              int32_t Ho, int32_t Wo, int32_t add_cstride, int32_t add_coff,
                              int32_t out_cstride, int32_t out_coff, void *stream);

/* tf.reduce_mean over H,W (semantic.py:149; GlobalAveragePooling2D misc.py:43): [B,HW,C]->[B,C] */
int ml_global_mean_f32(const float *in, float *out, int32_t B, int32_t HW, int32_t C, void *stream);

/* x[b,h,w,c] *= s[b,c]  (SqueezeExcite scale, misc.py:46-47)                                 */
int ml_scale_channels_f32(float *x, const float *s, int32_t B, int32_t HW, int32_t C, void *stream);

/* ---------------------------------------------------------------- detection post-process
 * RestoreBoxes (engine/layers/detection.py:325-344): priors int32 [A,4] (cx,cy,w,h) shared by
 * the batch; loc [B,A,4] -> boxes [B,A,4].                                                    */
int ml_restore_boxes_f32(const float *loc, const int32_t *priors, float *boxes,
                         int32_t B, int32_t A, void *stream);

/* DetectionProposal (detection.py:482-567) in fixed capacity: threshold -> per-(image,class)
 * greedy NMS -> per-image cross-class NMS -> rows (cx,cy,w,h,class,conf), -1 padded.
 *   cls_pred [B,A,C], boxes [B,A,4] -> proposed [B,max_out,6], counts [B] (int32),
 *   kept [B,max_out,2] (anchor, class) int32 or NULL,
 *   gather_payload [B, max_out*6 + 1] or NULL: per image the 6*max_out floats of `proposed` followed
 *   by the count bit-cast to float -- the fixed-size record one RCCL all-gather merges across GPUs
 *   (the reference's DP merge is Concatenate(axis=0), engine/parallel.py:92-107).
 * C <= 64; C*max_out <= 2048 keeps the cross-class stage in LDS, larger values (the constructor
 * default nms_max_output_size=1000, detection.py:472) run it through the workspace.
 * workspace: >= ml_detection_workspace_bytes(B,A,C,max_out) bytes.                            */
int64_t ml_detection_workspace_bytes(int32_t B, int32_t A, int32_t C, int32_t max_out);
int ml_detection_proposal_f32(const float *cls_pred, const float *boxes, float *proposed,
                              int32_t *counts, int32_t *kept, float *gather_payload,
                              int32_t B, int32_t A, int32_t C,
                              float min_confidence, float nms_iou, float post_iou, int32_t max_out,
                              void *workspace, void *stream);

/* MaskDistribute (engine/layers/instance.py:52-66) + the per-level `tf.where` of
 * PyramidRoiAlign (instance.py:121): proposed [B,cap,6] -> level_slots [B,L,cap] (row indices,
 * ascending), level_counts [B,L], level_max [L] or NULL = max over the images of level_counts
 * (the second axis MoldBatch gives each level's crops, misc.py:235-236: the one host read of
 * the forward).  L = max_k+1.                                                                  */
int ml_mask_distribute_i32(const float *rows, int32_t row_stride, int32_t has_k,
                           float *kvals /* [B,cap] or NULL */, int32_t *level_slots,
                           int32_t *level_counts, int32_t *level_max, int32_t B, int32_t cap,
                           int32_t max_k, float base_size, void *stream);
/* rows: [B,cap,row_stride]; has_k=0: rows are (cx,cy,w,h,cls,conf) and k is computed
 * (MaskDistribute); has_k=1: rows are dist_boxes (k,cx,cy,w,h,cls,conf) and k is read.        */

/* tf.image.crop_and_resize + MoldBatch(-1) for one pyramid level (instance.py:115-134):
 * fmap [B,Hf,Wf,C]; rows [B,cap,row_stride] with (cx,cy,w,h,cls,conf) starting at column row_off;
 * writes roi_fmaps [B,n_l,ch,cw,C] and roi_boxes[b, box_off+j, 0..5]
 * (row stride box_rows*6); slots j >= level_counts[b,level] are filled with -1.              */
int ml_roi_crop_resize_f32(const float *fmap, const float *rows, int32_t row_stride, int32_t row_off,
                           const int32_t *level_slots,
                           const int32_t *level_counts, float *roi_fmaps, float *roi_boxes,
                           int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t cap, int32_t L,
                           int32_t level, int32_t n_l, int32_t ch, int32_t cw,
                           float img_h, float img_w, int32_t box_off, int32_t box_rows,
                           const int32_t *live /* NULL, or a device int: with n_l = cap (no host read of the counts) only
                                                  slots j < max(1, *live) are written (crop or -1 fill) */,
                           void *stream);
/* The same with fmap / roi_fmaps in IEEE half (C % 8 == 0): the mask head of the fp16 path; boxes and rows stay fp32. */
int ml_roi_crop_resize_f16(const void *fmap, const float *rows, int32_t row_stride, int32_t row_off,
                           const int32_t *level_slots, const int32_t *level_counts, void *roi_fmaps, float *roi_boxes,
                           int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t cap, int32_t L,
                           int32_t level, int32_t n_l, int32_t ch, int32_t cw,
                           float img_h, float img_w, int32_t box_off, int32_t box_rows, const int32_t *live,
                           void *stream);

/* MoldBatch + Concatenate(axis=1) of a fixed-capacity stage 2 (instance.py:222-225, misc.py:231-286): src [B, L*cap, E]
 * holds level l's RoIs at rows l*cap ..; dst [B, sum n_l, E] receives rows [l*cap, l*cap + n_l) of every image, the
 * levels next to each other.  n_l: L HOST ints (1 <= n_l <= cap) -- the one read of the forward, done after all of it
 * has been enqueued.  E % 4 == 0.                                                                                 */
int ml_mold_levels_f32(const float *src, float *dst, int32_t B, int32_t L, int32_t cap, int64_t E,
                       const int32_t *n_l, void *stream);
/* The same with the level sizes read ON THE DEVICE: lmax_dev = the L per-level RoI maxima ml_mask_distribute_f32 wrote
 * (n_l = min(max(1, lmax), cap), as the host computes them) -- no host value enters the launch, so it can be part of a
 * captured hipGraph; dst is a capacity buffer of B * L * cap * E floats whose FRONT receives the [B, sum n_l, E] tensor
 * (the host, once it has read lmax, takes that front as a view: no launch after the graph).  Any E.                */
int ml_mold_levels_dev_f32(const float *src, float *dst, int32_t B, int32_t L, int32_t cap, int64_t E,
                           const int32_t *lmax_dev, void *stream);

/* x += y over n floats (n % 4 == 0): the `Add` of MobileSeparableConv2D (misc.py:92,105) */
int ml_add_f32(float *x, const float *y, int64_t n, void *stream);

/* fill n floats with v */
int ml_fill_f32(float *x, float v, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Deploy wrapper either side of the forward (reference engine/retinamasklab.py:598-643) --
 * SURVEY section 8(f) ranks 1 and 2.
 * ------------------------------------------------------------------------------------------- */
#define ML_SMOOTH_MAX_CLASSES 16

/* tf.compat.v1.image.resize_bilinear(align_corners=True) for any channel count: replaces the resize
 * inside DownSampleInput.call (engine/layers/misc.py:143-154; uint8 or float images, cast to f32
 * first), ResizeLike on the 3-channel semantic map (retinamasklab.py:628) and the semantic resize of
 * UpSampleOutput.call (misc.py:190-193).  in: [B,H,W,C] u8 (in_is_u8) or f32.  Writes the f32 result
 * to out_f32 and/or `value > threshold ? 1 : 0` to out_i32 (either may be NULL), both [B,Ho,Wo,C]. */
int ml_resize_image_ac(const void *in, int32_t in_is_u8, float *out_f32, int32_t *out_i32, float threshold,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream);

/* TrimInstances.call (engine/layers/instance.py:258-277) with mold=True, fixed capacity: per image the
 * rows of roi_boxes [B,N,6] whose class (column 4) != -1 are moved to the front in order, their mask
 * is the class channel of roi_masks [B,N,mh,mw,C]; out_boxes [B,N,6] / out_masks [B,N,mh,mw] are
 * -1 padded (MoldBatch, misc.py:231-286) and counts[b] = rows kept.  The reference's dynamic second
 * axis is max(counts): the caller slices.                                                         */
int ml_trim_instances_f32(const float *roi_boxes, const float *roi_masks, float *out_boxes, float *out_masks,
                          int32_t *counts, int32_t B, int32_t N, int32_t mh, int32_t mw, int32_t C, void *stream);

/* UpSampleOutput.call box part (engine/layers/misc.py:178-187): rows (cx,cy,w,h,label,conf) f32 ->
 * int32 (cx*ratio0, cy*ratio1, w*ratio0, h*ratio1, label, conf*100), truncating like tf.cast.
 * ratio0 is the HEIGHT ratio and ratio1 the width ratio, as in the reference.                       */
int ml_upsample_boxes_i32(const float *rows, int32_t *out, int64_t n_rows, float ratio0, float ratio1, void *stream);

/* out[i] = in[i] > threshold ? 1 : 0  (tf.cast(x > 0.5, tf.int32), misc.py:189,194) */
int ml_threshold_i32(const float *in, int32_t *out, float threshold, int64_t n, void *stream);

/* SemanticSmoothing.call (engine/layers/semantic.py:270-285) for all classes at once: per channel c a
 * grey opening -- tf.nn.erosion2d then tf.nn.dilation2d with an all-zero kernel_sizes[c]^2 element,
 * stride 1, SAME (window rows y-(k-1)/2 .. +k-1, positions outside the map ignored) -- times
 * weights[c]; kernel_sizes[c] <= 0 only applies the weight.  in/out/tmp: three distinct [B,H,W,C]
 * buffers; kernel_sizes / weights are HOST arrays of C entries (C <= ML_SMOOTH_MAX_CLASSES).        */
int ml_semantic_smoothing_f32(const float *in, float *out, float *tmp, int32_t B, int32_t H, int32_t W, int32_t C,
                              const int32_t *kernel_sizes, const float *weights, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Serving post-processing (reference road_project/setup/serving.py:28-50) -- SURVEY section 8(f) rank 4.
 * Only the arithmetic layers: JPEG decode / drawing / encode are outside this library.
 * ------------------------------------------------------------------------------------------- */

/* CropAndPadMask.call (engine/layers/misc.py:358-401): det [B,n,6] int32 (cx,cy,w,h,label,conf*100 from
 * UpSampleOutput), masks [B,n,mh,mw] int32 -> out [B,n,H,W] f32: for every row with conf >= threshold
 * (threshold = 50 if max(conf) > 50 else -100) the mask resized bilinear(align_corners=True) to its box
 * (box = max(box, 1); corners ceil(c -+ size/2) clipped to the canvas) and zero padded to H x W; other
 * rows zero.  A box clipped to zero size pastes nothing (the reference's resize would raise).
 * threshold_ws: one int32 of device scratch.                                                       */
int ml_crop_pad_mask_f32(const int32_t *det, const int32_t *masks, float *out, int32_t *threshold_ws,
                         int32_t B, int32_t n, int32_t mh, int32_t mw, int32_t H, int32_t W, void *stream);

/* CrackToInstance.call bounding box (engine/layers/misc.py:533-541): min / max (y, x) of the non-zero
 * entries of channel `coff` of an int32 [B,H,W,cstride] map over the WHOLE batch.  box5 (device, caller
 * initialises to {INT32_MAX, INT32_MAX, -1, -1, 0}) receives {ymin, xmin, ymax, xmax, any}.           */
int ml_nonzero_bbox_i32(const int32_t *map, int32_t B, int32_t H, int32_t W, int32_t cstride, int32_t coff,
                        int32_t *box5, void *stream);

/* SummaryOutput's per-instance numbers (engine/layers/misc.py:574-589): CalculateInstanceSize (:632-718:
 * per image row the min / max x of the `road_channel` pixels > 0, rows with min != max, 15 % dropped at
 * both ends, least-squares lines x(y) of the left and right edge -- float32 normal equations, 2x2 LU with
 * partial pivoting like tf.linalg.inv --, unit[y] = default_road_size / clip(right - left, 1, inf)) and
 * IncludeMyRoad (:601-618).  seg [B,H,W,seg_channels] int32, masks [B,n,H,W] f32 (CropAndPadMask output)
 * -> out5 [B,n,5] = {pixel sum, instance size, horizontal size, vertical size, include_my_road}.     */
int64_t ml_instance_summary_workspace_bytes(int32_t B, int32_t H);
int ml_instance_summary_f32(const int32_t *seg, int32_t seg_channels, int32_t road_channel, const float *masks,
                            float *out5, int32_t B, int32_t n, int32_t H, int32_t W, float default_road_size,
                            float ioi_threshold, void *workspace, void *stream);
/* The same numbers WITHOUT the padded canvases: CropAndPadMask (misc.py:358-401) and SummaryOutput's arithmetic
 * (:574-589) in one pass.  det [B,n,6] int32 and roi_masks [B,n,mh,mw] int32 are CropAndPadMask's inputs; every canvas
 * value is recomputed on the fly with that layer's arithmetic and only the rows / columns a box covers are visited
 * (what is skipped is exactly zero), so out5 is bit-identical to ml_crop_pad_mask_f32 followed by
 * ml_instance_summary_f32 while the [B,n,H,W] tensor (3.4 GB at 8 x 100 x 1024^2) is never written or read.
 * Same workspace size as ml_instance_summary_f32.                                                        */
int ml_instance_summary_rois_f32(const int32_t *seg, int32_t seg_channels, int32_t road_channel, const int32_t *det,
                                 const int32_t *roi_masks, float *out5, int32_t B, int32_t n, int32_t mh, int32_t mw,
                                 int32_t H, int32_t W, float default_road_size, float ioi_threshold, void *workspace,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKLAB_HIP_H */
