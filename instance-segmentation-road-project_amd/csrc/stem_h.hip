// The ResNeXt stem of the fp16-storage mode in ONE pass: ZeroPadding2D(3) + Conv 7x7 stride 2 (64 filters, BatchNorm folded) +
// ReLU + ZeroPadding2D(1) + MaxPooling2D(3, stride 2), fp32 NHWC4 image in, IEEE-half pooled map out
// (reference engine/backbone/ResNext.py:343-352; thirdparty/classification_models/models/resnext.py:193-197).
//
// Why fused, and only on this path: at 16 x 1280^2 the stem's output is 839 MB of half written and read back by the pool
// (0.78 + 0.23 ms of the fp16 step) while the conv itself is ~90 us of fp16 MFMA -- recomputing the pool's one-pixel halo
// (9 x 33 conv pixels for a 4 x 16 pooled tile: +16 %) costs nothing beside the bytes it saves.  (fp32 tensors: stem_f32.hip /
// stem_x3.hip.)  A caller that asks for the C1 tap (the un-pooled stem output) gets the unfused pair.
//
//   * K order = (kernel row, 8 pixels x 4 channels) exactly as the generic kernel's row-span packing: 7 rows x 32, the 8th
//     pixel and the 4th channel carry zero weights; a 16-deep MFMA step is half a kernel row, lane half h two of its pixels
//     (one ds_read_b128 of the half NHWC4 tile in LDS).  Same operands (image and weights rounded to half, RNE), same
//     k order, bias in the accumulator init, ReLU, ONE rounding to half: bit-identical to conv_mfma's ML_MATH_F16 stem
//     followed by maxpool3x3s2_f16.
//   * transposed product: A = weights (rows = output channels), B = pixels, so lane (p, q) ends up with four runs of four
//     consecutive channels of ITS conv pixel -> 8-byte writes into the conv tile in LDS.
//   * a block walks a ROW of pooled tiles; wave w multiplies output channels 32 (w >> 1) .. + 31 by every other set of 32
//     conv pixels, its 14 weight fragments resident in registers, and the next tile's input pixels are fetched into
//     registers under the current tile's MFMAs (two barriers per tile).
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int PTH = 4, PTW = 16;                 // pooled tile
constexpr int CR = 2 * PTH + 1, CC = 2 * PTW + 1; // conv pixels it needs: 9 x 33
constexpr int IR = 2 * (CR - 1) + 7;             // input rows: 23
constexpr int ICP = 2 * (CC - 1) + 8;            // input pixels per row: 72 (71 used + the zero-weight 8th tap pixel)
constexpr int NCONV = CR * CC;                   // 297
constexpr int NSETS = (NCONV + 31) / 32;         // 10 sets of 32 conv pixels
constexpr int CPS = 72;                          // halves per conv pixel in LDS (64 + 8 pad: 144 B)
constexpr int IN_BYTES = IR * ICP * 8;           // 13 248
constexpr int CONV_BYTES = NSETS * 32 * CPS * 2; // 46 080
constexpr int STEM_LDS = IN_BYTES + CONV_BYTES;

__global__ void __launch_bounds__(256)
stem_pool_h_kernel(const float *__restrict__ img, const _Float16 *__restrict__ wgt, const float *__restrict__ bias,
                   _Float16 *__restrict__ out, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x) {
    extern __shared__ __align__(16) char lds[];
    _Float16 *tin = reinterpret_cast<_Float16 *>(lds);                 // [IR][ICP][4]
    _Float16 *tconv = reinterpret_cast<_Float16 *>(lds + IN_BYTES);    // [NSETS * 32][CPS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p32 = lane & 31, q = lane >> 5;
    const int ty = blockIdx.x, b = blockIdx.y;
    const int py0 = ty * PTH;
    const int cy0 = 2 * py0 - 1;                                        // conv row of tile-local row 0
    const int iy0 = 2 * cy0 - 3;                                        // input row of tile-local row 0
    // wave w multiplies the 32 output channels 32 (w >> 1) .. by the conv-pixel sets (w & 1), (w & 1) + 2, ...: 5 sets x 14
    // MFMAs each, and only 14 weight fragments (56 registers) per lane -- room for the next tile's input in registers
    const int nt = wave >> 1, s0 = wave & 1;

    // weights: A fragments.  lane (m = p32, q): output channel 32 nt + m, k = 16 s + 8 q .. + 7
    f16x8 wv[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) wv[s] = *reinterpret_cast<const f16x8 *>(wgt + (nt * 32 + p32) * 224 + s * 16 + q * 8);
    // bias of the channels this lane's accumulator registers hold: 32 nt + (e & 3) + 8 (e >> 2) + 4 q
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * q] : 0.f;

    // ---- this thread's share of an input tile: elements tid + 256 j of the [IR][ICP] pixel grid (fp32 NHWC4 -> half)
    constexpr int NIN = (IR * ICP + 255) / 256;                         // 7
    int in_r[NIN], in_c[NIN];
    long long in_off[NIN];                                              // float offset of (row, column 0 of tile 0)
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const int i = tid + 256 * j;
        in_r[j] = i / ICP;
        in_c[j] = i - in_r[j] * ICP;
        const int iy = iy0 + in_r[j];
        const bool row_ok = i < IR * ICP && (unsigned)iy < (unsigned)H;
        in_off[j] = row_ok ? ((long long)(b * H + iy) * W) * 4 : -1;
    }
    f32x4 stage[NIN];
    auto fetch = [&](int tx) __attribute__((always_inline)) {         // global -> registers (zeros outside the image)
        const int ix0 = 2 * (2 * tx * PTW - 1) - 3;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int ix = ix0 + in_c[j];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[j] >= 0 && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4 *>(img + in_off[j] + (long long)ix * 4);
            stage[j] = v;
        }
    };
    auto deposit = [&]() __attribute__((always_inline)) {              // registers -> LDS, rounded to half
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int i = tid + 256 * j;
            if (i < IR * ICP)
                *reinterpret_cast<f16x4 *>(tin + i * 4) = f16x4{(_Float16)stage[j][0], (_Float16)stage[j][1], (_Float16)stage[j][2], (_Float16)stage[j][3]};
        }
    };

    fetch(0);
    deposit();
    __syncthreads();
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int px0 = tx * PTW;
        const int cx0 = 2 * px0 - 1;
        if (tx + 1 < tiles_x) fetch(tx + 1);                            // the next tile's pixels fly under this tile's MFMAs

        // ---- conv: set s = conv pixels 32 s .. + 31 of the 9 x 33 region (row-major)
        for (int s = s0; s < NSETS; s += 2) {
            const int cp = min(s * 32 + p32, NCONV - 1);                // (the last set's spare lanes recompute pixel 296)
            const int cyl = cp / CC, cxl = cp - cyl * CC;
            const _Float16 *src = tin + ((2 * cyl) * ICP + 2 * cxl + 2 * q) * 4;      // kernel row 0, pixels 2 q, 2 q + 1
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bv[e];
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f16x8 xv = *reinterpret_cast<const f16x8 *>(src + (ky * ICP + 4 * s2) * 4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[ky * 2 + s2], xv, acc, 0, 0, 0);
                }
            // ReLU, one rounding; conv pixels outside the conv map are the pool's zero padding
            const int cy = cy0 + cyl, cx = cx0 + cxl;
            const bool inside = (unsigned)cy < (unsigned)Hc && (unsigned)cx < (unsigned)Wc;
            _Float16 *dst = tconv + (s * 32 + p32) * CPS + nt * 32 + 4 * q;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                f16x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = inside ? (_Float16)fmaxf(acc[4 * e4 + e], 0.f) : (_Float16)0.f;
                *reinterpret_cast<f16x4 *>(dst + 8 * e4) = hv;
            }
        }
        __syncthreads();                                                // conv tile complete; every wave is done reading `tin`

        // ---- 3 x 3 stride-2 max over the conv tile: 64 pooled pixels x 8 runs of 8 channels
        for (int i = tid; i < PTH * PTW * 8; i += 256) {
            const int cg = i & 7, pp = i >> 3;
            const int ppy = pp / PTW, ppx = pp - ppy * PTW;
            const int oy = py0 + ppy, ox = px0 + ppx;
            if (oy >= Hp || ox >= Wp) continue;
            f16x8 m = *reinterpret_cast<const f16x8 *>(tconv + ((2 * ppy) * CC + 2 * ppx) * CPS + cg * 8);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const f16x8 v = *reinterpret_cast<const f16x8 *>(tconv + ((2 * ppy + dy) * CC + 2 * ppx + dx) * CPS + cg * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
                }
            *reinterpret_cast<f16x8 *>(out + ((long long)(b * Hp + oy) * Wp + ox) * 64 + cg * 8) = m;
        }
        if (tx + 1 < tiles_x) deposit();                                // (`tin` is free since the barrier above)
        __syncthreads();                                                // next input tile visible; pool done with `tconv`
    }
}

}  // namespace

extern "C" int ml_stem7x7s2_pool_f16(const float *image, const void *wgt_h, const float *bias, void *out, int32_t B, int32_t H,
                                     int32_t W, int32_t Hp, int32_t Wp, void *stream) {
    ML_REQUIRE(image && wgt_h && out, "stem7x7s2_pool: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, "stem7x7s2_pool: bad dims");
    ML_REQUIRE(ml_aligned16(image) && ml_aligned16(wgt_h) && ml_aligned16(out), "stem7x7s2_pool: pointers must be 16-byte aligned");
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;       // ZeroPadding2D(3) + 7x7 stride 2 'valid'
    ML_REQUIRE(Hp == (Hc + 2 - 3) / 2 + 1 && Wp == (Wc + 2 - 3) / 2 + 1,
               "stem7x7s2_pool: output must be [B, %d, %d, 64] (ZeroPadding2D(1) + MaxPooling2D(3, 2))", (Hc + 2 - 3) / 2 + 1,
               (Wc + 2 - 3) / 2 + 1);
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "stem7x7s2_pool: too many pixels");
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(stem_pool_h_kernel), STEM_LDS, lds_ok, "stem7x7s2_pool")) return rc;
    const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
    hipLaunchKernelGGL(stem_pool_h_kernel, dim3(tiles_y, B), dim3(256), STEM_LDS, (hipStream_t)stream, image,
                       reinterpret_cast<const _Float16 *>(wgt_h), bias, reinterpret_cast<_Float16 *>(out), H, W, Hc, Wc, Hp, Wp,
                       tiles_x);
    ML_CHECK_LAUNCH("stem7x7s2_pool");
    return ML_OK;
}
