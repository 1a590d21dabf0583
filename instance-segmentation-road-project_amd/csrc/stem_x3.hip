// The ResNeXt stem of the split-operand mode (ML_MATH_F32X3) in ONE pass: ZeroPadding2D(3) + Conv 7x7 stride 2 (64 filters,
// BatchNorm folded) + ReLU + ZeroPadding2D(1) + MaxPooling2D(3, stride 2), fp32 NHWC4 image in, fp32 pooled map out
// (reference engine/backbone/ResNext.py:343-352; thirdparty/classification_models/models/resnext.py:193-197).
// Twins: stem_h.hip (fp16 storage), stem_f32.hip (exact fp32 products).
//
// Why: as two launches the stem costs 336 + 146 us of that mode's 12.0 ms step at 8 x 1024^2, nearly all of it the 537 MB
// of un-pooled output written and read back; the products themselves are 42 v_mfma_f32_32x32x16_f16 of 32 cycles per 32
// pixels x 32 channels.
//   * operands: every image value is split ONCE, when its pixel is put into LDS (hi = RNE half, lo = (x - hi) 2^11 as a
//     half: split_hi_lo_pair, the generic kernel's arithmetic), into two half NHWC4 tiles; the weights arrive split
//     (masklab_hip.ops.DeviceConv.wgt_x3: per kernel row 32 hi halves then 32 lo halves);
//   * K order and products exactly as conv_mfma's X3 path on the row-span packing: per kernel row two 16-deep steps (4 pixels
//     x 4 channels each), per step hi hi -> acc (started from the bias), then act-hi x wgt-lo and act-lo x wgt-hi -> the
//     cross-term accumulator, folded in once at the end (2^-11), ReLU -- BIT-IDENTICAL to conv2d (f32x3) + maxpool3x3s2;
//   * tile geometry, transposed product (A = weights, B = pixels), LDS conv tile, pool and the register prefetch of the next
//     tile's pixels: stem_f32.hip's.
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int PTH = 4, PTW = 16;                  // pooled tile
constexpr int NT = 256;
constexpr int CR = 2 * PTH + 1, CC = 2 * PTW + 1; // conv pixels it needs: 9 x 33
constexpr int IR = 2 * (CR - 1) + 7;              // input rows: 23
constexpr int ICP = 2 * (CC - 1) + 8;             // input pixels per row: 72 (71 used + the zero-weight 8th tap pixel)
constexpr int NCONV = CR * CC;                    // 297
constexpr int NSETS = (NCONV + 31) / 32;          // 10 sets of 32 conv pixels: 5 per wave
constexpr int CPS = 68;                           // floats per conv pixel in LDS (64 + 4 pad: 272 B)
constexpr int IN_BYTES = IR * ICP * 8;            // 13 248 per half tile (hi, lo)
constexpr int CONV_BYTES = NSETS * 32 * CPS * 4;  // 87 040
constexpr int STEM_LDS = 2 * IN_BYTES + CONV_BYTES;   // 113 536: one block per CU
static_assert(NSETS == 10, "the wave -> set map below assumes 10 sets");

__global__ void __launch_bounds__(NT, 1)
stem_pool_x3_kernel(const float *__restrict__ img, const _Float16 *__restrict__ wgt, const float *__restrict__ bias,
                    float *__restrict__ out, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x, int seg) {
    extern __shared__ __align__(16) char lds[];
    _Float16 *tin_h = reinterpret_cast<_Float16 *>(lds);                    // [IR][ICP][4] hi halves
    _Float16 *tin_l = reinterpret_cast<_Float16 *>(lds + IN_BYTES);         // [IR][ICP][4] lo halves (x 2^11)
    float *tconv = reinterpret_cast<float *>(lds + 2 * IN_BYTES);           // [NSETS * 32][CPS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p32 = lane & 31, q = lane >> 5;
    const int ty = blockIdx.x, b = blockIdx.y;
    const int py0 = ty * PTH;
    const int cy0 = 2 * py0 - 1;                                        // conv row of tile-local row 0
    const int iy0 = 2 * cy0 - 3;                                        // input row of tile-local row 0
    const int nt = wave >> 1, s0 = wave & 1;                            // sets s0, s0 + 2, .. + 8 against channels 32 nt ..

    // weights: A fragments.  lane (m = p32, q): output channel 32 nt + m, k = 16 s + 8 q .. + 7 of kernel row ky (hi), + 32 (lo)
    f16x8 wh[14], wl[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const _Float16 *row = wgt + ((nt * 32 + p32) * 7 + (s >> 1)) * 64 + (s & 1) * 16 + q * 8;
        wh[s] = *reinterpret_cast<const f16x8 *>(row);
        wl[s] = *reinterpret_cast<const f16x8 *>(row + 32);
    }
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * q] : 0.f;

    // ---- this thread's share of an input tile: pixels tid + 256 j of the [IR][ICP] grid
    constexpr int NIN = (IR * ICP + NT - 1) / NT;                       // 7
    int in_c[NIN];
    long long in_off[NIN];                                              // float offset of (row, column 0 of the image), -1: zeros
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const int i = tid + NT * j;
        const int in_r = i / ICP;
        in_c[j] = i - in_r * ICP;
        const int iy = iy0 + in_r;
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
    const float neg_scale = -2048.f;
    auto deposit = [&]() __attribute__((always_inline)) {              // registers -> LDS, split into hi / lo halves
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int i = tid + NT * j;
            if (i < IR * ICP) {
                unsigned h0, l0, h1, l1;
                split_hi_lo_pair(stage[j][0], stage[j][1], neg_scale, h0, l0);
                split_hi_lo_pair(stage[j][2], stage[j][3], neg_scale, h1, l1);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2 *>(tin_h + i * 4) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(tin_l + i * 4) = u32x2{l0, l1};
            }
        }
    };
    // one or two sets of 32 conv pixels against this wave's 32 output channels
    auto conv_sets = [&](auto nc, int sa, int cx0) __attribute__((always_inline)) {
        constexpr int NS = decltype(nc)::value;
        int soff[NS];                                                   // halves: kernel row 0, pixels 2 q, 2 q + 1
        f32x16 acc[NS], acx[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int cp = min((sa + 2 * u) * 32 + p32, NCONV - 1);     // (the last set's spare lanes recompute pixel 296)
            const int cyl = cp / CC, cxl = cp - cyl * CC;
            soff[u] = ((2 * cyl) * ICP + 2 * cxl + 2 * q) * 4;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[u][e] = bv[e]; acx[u][e] = 0.f; }
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 xh[NS], xl[NS];
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    xh[u] = *reinterpret_cast<const f16x8 *>(tin_h + soff[u] + (ky * ICP + 4 * s2) * 4);
                    xl[u] = *reinterpret_cast<const f16x8 *>(tin_l + soff[u] + (ky * ICP + 4 * s2) * 4);
                }
#pragma unroll
                for (int u = 0; u < NS; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ky * 2 + s2], xh[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NS; ++u) acx[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ky * 2 + s2], xh[u], acx[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NS; ++u) acx[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ky * 2 + s2], xl[u], acx[u], 0, 0, 0);
            }
        // cross terms folded in (units of 2^-11), ReLU; conv pixels outside the conv map are the pool's zero padding
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int s = sa + 2 * u;
            const int cp = min(s * 32 + p32, NCONV - 1);
            const int cyl = cp / CC, cxl = cp - cyl * CC;
            const int cy = cy0 + cyl, cx = cx0 + cxl;
            const bool inside = (unsigned)cy < (unsigned)Hc && (unsigned)cx < (unsigned)Wc;
            float *dst = tconv + (s * 32 + p32) * CPS + nt * 32 + 4 * q;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(fmaf(acx[u][4 * e4 + e], 0x1p-11f, acc[u][4 * e4 + e]), 0.f) : 0.f;
                *reinterpret_cast<f32x4 *>(dst + 8 * e4) = v;
            }
        }
    };

    const int tx_begin = blockIdx.z * seg, tx_end = min(tiles_x, tx_begin + seg);
    fetch(tx_begin);
    deposit();
    __syncthreads();
    for (int tx = tx_begin; tx < tx_end; ++tx) {
        const int px0 = tx * PTW;
        const int cx0 = 2 * px0 - 1;
        if (tx + 1 < tx_end) fetch(tx + 1);                             // the next tile's pixels fly under this tile's MFMAs

        conv_sets(std::integral_constant<int, 2>{}, s0, cx0);
        conv_sets(std::integral_constant<int, 2>{}, s0 + 4, cx0);
        conv_sets(std::integral_constant<int, 1>{}, s0 + 8, cx0);
        __syncthreads();                                                // conv tile complete; every wave is done reading the input tiles

        // ---- 3 x 3 stride-2 max over the conv tile: 64 pooled pixels x 16 runs of 4 channels
        for (int i = tid; i < PTH * PTW * 16; i += NT) {
            const int cg = i & 15, pp = i >> 4;
            const int ppy = pp / PTW, ppx = pp - ppy * PTW;
            const int oy = py0 + ppy, ox = px0 + ppx;
            if (oy >= Hp || ox >= Wp) continue;
            f32x4 m = *reinterpret_cast<const f32x4 *>(tconv + ((2 * ppy) * CC + 2 * ppx) * CPS + cg * 4);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(tconv + ((2 * ppy + dy) * CC + 2 * ppx + dx) * CPS + cg * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
                }
            *reinterpret_cast<f32x4 *>(out + ((long long)(b * Hp + oy) * Wp + ox) * 64 + cg * 4) = m;
        }
        if (tx + 1 < tx_end) deposit();                                 // (the input tiles are free since the barrier above)
        __syncthreads();                                                // next input tile visible; pool done with `tconv`
    }
}

}  // namespace

extern "C" int ml_stem7x7s2_pool_x3(const float *image, const void *wgt_x3, const float *bias, float *out, int32_t B, int32_t H,
                                    int32_t W, int32_t Hp, int32_t Wp, void *stream) {
    ML_REQUIRE(image && wgt_x3 && out, "stem7x7s2_pool_x3: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, "stem7x7s2_pool_x3: bad dims");
    ML_REQUIRE(ml_aligned16(image) && ml_aligned16(wgt_x3) && ml_aligned16(out), "stem7x7s2_pool_x3: pointers must be 16-byte aligned");
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;       // ZeroPadding2D(3) + 7x7 stride 2 'valid'
    ML_REQUIRE(Hp == (Hc + 2 - 3) / 2 + 1 && Wp == (Wc + 2 - 3) / 2 + 1,
               "stem7x7s2_pool_x3: output must be [B, %d, %d, 64] (ZeroPadding2D(1) + MaxPooling2D(3, 2))", (Hc + 2 - 3) / 2 + 1,
               (Wc + 2 - 3) / 2 + 1);
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "stem7x7s2_pool_x3: too many pixels");
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(stem_pool_x3_kernel), STEM_LDS, lds_ok, "stem7x7s2_pool_x3")) return rc;
    const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
    // whole rows per block when that gives two blocks per CU or more (see stem_f32.hip)
    int seg = tiles_x;
    while (seg > 1 && (long long)tiles_y * B * ((tiles_x + seg - 1) / seg) < 2ll * ml_resident_blocks(1)) seg = (seg + 1) / 2;
    hipLaunchKernelGGL(stem_pool_x3_kernel, dim3(tiles_y, B, (tiles_x + seg - 1) / seg), dim3(NT), STEM_LDS, (hipStream_t)stream,
                       image, reinterpret_cast<const _Float16 *>(wgt_x3), bias, out, H, W, Hc, Wc, Hp, Wp, tiles_x, seg);
    ML_CHECK_LAUNCH("stem7x7s2_pool_x3");
    return ML_OK;
}
