// The ResNeXt stem in ONE pass on fp32 tensors with exact fp32 products: ZeroPadding2D(3) + Conv 7x7 stride 2 (64 filters,
// BatchNorm folded) + ReLU + ZeroPadding2D(1) + MaxPooling2D(3, stride 2), fp32 NHWC4 image in, fp32 pooled map out
// (reference engine/backbone/ResNext.py:343-352; thirdparty/classification_models/models/resnext.py:193-197).
// The fp16-storage twin is stem_h.hip; this one is the headline's (round 4).
//
// Why: as two launches the stem costs 545 + 143 us of the 21.3 ms step at 8 x 1024^2 -- the generic kernel's row-span packing
// pads a kernel row of 7 pixels x 3 channels to 8 x 4 = 32 floats (K = 224 for 147 real taps: a third of its MFMAs multiply
// zeros), and the 537 MB stem output is written and read back by the pool.  Here
//   * only products with a non-zero weight are issued, in the generic kernel's own pairs: its v_mfma_f32_32x32x2_f32 number j
//     of k-step ks of a kernel row multiplies (pixel 2 ks, channel j) and (pixel 2 ks + 1, channel j); channel 3 (j = 3) has
//     zero weights -- dropped, it adds +0 -- and pixel 7 too (its partner's products stay).  12 MFMAs per kernel row instead
//     of 16, the same chain per output: bias first, kernel rows in order, ReLU -- BIT-IDENTICAL to conv_mfma + maxpool3x3s2;
//   * the conv output lives in LDS only: a block computes the 9 x 33 conv pixels a 4 x 16 pooled tile needs (+16 % for the
//     pool's one-pixel halo), pools them from LDS and stores 64 pooled pixels x 64 channels;
//   * transposed product as in stem_h.hip: A = weights (rows = output channels), B = pixels, so a lane ends up with runs of
//     four consecutive channels of ITS pixel -> 16-byte writes into the conv tile; the B operand of an MFMA is one float per
//     lane, read from the 3-channel input tile in LDS at an immediate offset (lane half q = the odd pixel of the pair: + 3
//     floats; 32 lanes read conv pixels 24 bytes apart: no bank conflict);
//   * a block walks a ROW of pooled tiles with its 84 weight registers resident, the next tile's image pixels fetched into
//     registers under the current tile's MFMAs; wave w multiplies output channels 32 (w >> 1) .. + 31 by every other set of
//     32 conv pixels, two sets at a time (two independent accumulator chains).
// Measured at 8 x 1024^2: 507 us against 545 + 143 for the two launches -- 71 % of the MFMA time of the chains it issues
// (8 192 tiles x 20 x 84 MFMAs of 64 cycles = 0.36 ms on 1 024 SIMDs), the same fraction the generic kernel reaches on this
// K = 7-chunk problem; pool, deposit and the two barriers per tile are not covered by matrix work with one block per CU.
// Measured and not kept: eight waves (two per SIMD, 5 x 16 pooled tile, one chain per wave, runs of two tiles per block):
// 548 us; the B operands read from LDS one kernel row ahead of their MFMAs (pinned with sched_barrier): 524 us -- the LDS
// latency is not what the chains wait for; runs of two tiles per block instead of whole rows: 582 us (the prologue).
// HBM traffic 100 MB in + 134 MB out.
#include <type_traits>
#include "common.h"

namespace {

constexpr int PTH = 4, PTW = 16;                  // pooled tile
constexpr int NT = 256;                           // threads: 4 waves, one per SIMD
constexpr int CR = 2 * PTH + 1, CC = 2 * PTW + 1; // conv pixels it needs: 9 x 33
constexpr int IR = 2 * (CR - 1) + 7;              // input rows: 23
constexpr int ICP = 2 * (CC - 1) + 8;             // input pixels per row: 72 (71 used + the zero-weight 8th tap pixel)
constexpr int NCONV = CR * CC;                    // 297
constexpr int NSETS = (NCONV + 31) / 32;          // 10 sets of 32 conv pixels: 5 per wave
constexpr int CPS = 68;                           // floats per conv pixel in LDS (64 + 4 pad: 272 B)
constexpr int IN_BYTES = IR * ICP * 3 * 4;        // 19 872: the input tile as 3-channel pixels
constexpr int CONV_BYTES = NSETS * 32 * CPS * 4;  // 87 040
constexpr int STEM_LDS = IN_BYTES + CONV_BYTES;   // 106 912: one block per CU

static_assert(NSETS == 10, "the wave -> set map below assumes 10 sets");

__global__ void __launch_bounds__(NT, 1)
stem_pool_f32_kernel(const float *__restrict__ img, const float *__restrict__ wgt, const float *__restrict__ bias,
                     float *__restrict__ out, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_x, int seg) {
    extern __shared__ __align__(16) char lds[];
    float *tin = reinterpret_cast<float *>(lds);                        // [IR][ICP][3]
    float *tconv = reinterpret_cast<float *>(lds + IN_BYTES);           // [NSETS * 32][CPS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p32 = lane & 31, q = lane >> 5;
    const int ty = blockIdx.x, b = blockIdx.y;
    const int py0 = ty * PTH;
    const int cy0 = 2 * py0 - 1;                                        // conv row of tile-local row 0
    const int iy0 = 2 * cy0 - 3;                                        // input row of tile-local row 0
    const int nt = wave >> 1, s0 = wave & 1;                            // sets s0, s0 + 2, .. + 8 against channels 32 nt ..

    // weights: A operand of MFMA (ky, ks, j) = W[output channel 32 nt + p32][kernel row ky][pixel 2 ks + q][channel j], straight
    // from the generic kernel's row-span packing ([64][7 x 32]: 8 pixels x 4 channels per kernel row)
    float wv[7][4][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j) wv[ky][ks][j] = wgt[(nt * 32 + p32) * 224 + ky * 32 + (2 * ks + q) * 4 + j];
    // bias of the channels this lane's accumulator registers hold: 32 nt + (e & 3) + 8 (e >> 2) + 4 q
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * q] : 0.f;

    // ---- this thread's share of an input tile: pixels tid + 256 j of the [IR][ICP] grid (fp32 NHWC4 -> 3 floats in LDS)
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
    auto deposit = [&]() __attribute__((always_inline)) {              // registers -> LDS
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int i = tid + NT * j;
            if (i < IR * ICP) {
                tin[i * 3 + 0] = stage[j][0];
                tin[i * 3 + 1] = stage[j][1];
                tin[i * 3 + 2] = stage[j][2];
            }
        }
    };
    // one or two sets of 32 conv pixels against this wave's 32 output channels
    auto conv_sets = [&](auto nc, int sa, int cx0) __attribute__((always_inline)) {
        constexpr int NS = decltype(nc)::value;
        const float *src[NS];
        f32x16 acc[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int cp = min((sa + 2 * u) * 32 + p32, NCONV - 1);     // (the last set's spare lanes recompute pixel 296)
            const int cyl = cp / CC, cxl = cp - cyl * CC;
            src[u] = tin + ((2 * cyl) * ICP + 2 * cxl + q) * 3;         // kernel row 0, pixel q of the pair, channel 0
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[u][e] = bv[e];
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int u = 0; u < NS; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ky][ks][j], src[u][(ky * ICP + 2 * ks) * 3 + j], acc[u], 0, 0, 0);
        // ReLU; conv pixels outside the conv map are the pool's zero padding
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
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(acc[u][4 * e4 + e], 0.f) : 0.f;
                *reinterpret_cast<f32x4 *>(dst + 8 * e4) = v;
            }
        }
    };

    // this block's run of pooled tiles in its row: seg of them (the launcher cuts rows so that blocks are many and short)
    const int tx_begin = blockIdx.z * seg, tx_end = min(tiles_x, tx_begin + seg);
    fetch(tx_begin);
    deposit();
    __syncthreads();
    for (int tx = tx_begin; tx < tx_end; ++tx) {
        const int px0 = tx * PTW;
        const int cx0 = 2 * px0 - 1;
        if (tx + 1 < tx_end) fetch(tx + 1);                             // the next tile's pixels fly under this tile's MFMAs

        // ---- conv: set s = conv pixels 32 s .. + 31 of the 9 x 33 region (row-major); this wave: s0, s0 + 2, ... (5 sets, two at a time)
        conv_sets(std::integral_constant<int, 2>{}, s0, cx0);
        conv_sets(std::integral_constant<int, 2>{}, s0 + 4, cx0);
        conv_sets(std::integral_constant<int, 1>{}, s0 + 8, cx0);
        __syncthreads();                                                // conv tile complete; every wave is done reading `tin`

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
        if (tx + 1 < tx_end) deposit();                                 // (`tin` is free since the barrier above)
        __syncthreads();                                                // next input tile visible; pool done with `tconv`
    }
}

}  // namespace

extern "C" int ml_stem7x7s2_pool_f32(const float *image, const float *wgt, const float *bias, float *out, int32_t B, int32_t H,
                                     int32_t W, int32_t Hp, int32_t Wp, void *stream) {
    ML_REQUIRE(image && wgt && out, "stem7x7s2_pool_f32: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, "stem7x7s2_pool_f32: bad dims");
    ML_REQUIRE(ml_aligned16(image) && ml_aligned16(out), "stem7x7s2_pool_f32: image and output must be 16-byte aligned");
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;       // ZeroPadding2D(3) + 7x7 stride 2 'valid'
    ML_REQUIRE(Hp == (Hc + 2 - 3) / 2 + 1 && Wp == (Wc + 2 - 3) / 2 + 1,
               "stem7x7s2_pool_f32: output must be [B, %d, %d, 64] (ZeroPadding2D(1) + MaxPooling2D(3, 2))", (Hc + 2 - 3) / 2 + 1,
               (Wc + 2 - 3) / 2 + 1);
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "stem7x7s2_pool_f32: too many pixels");
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(stem_pool_f32_kernel), STEM_LDS, lds_ok, "stem7x7s2_pool_f32")) return rc;
    const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
    // a block keeps its weights in registers over `seg` tiles of a row: whole rows when that gives two blocks per CU or more
    // (its prologue -- 84 strided weight loads per lane -- wants many tiles behind it: runs of 2 tiles measured 582 us against
    // 507 for whole rows at 8 x 1024^2), shorter runs only for launches that would otherwise leave CUs empty
    int seg = tiles_x;
    while (seg > 1 && (long long)tiles_y * B * ((tiles_x + seg - 1) / seg) < 2ll * ml_resident_blocks(1)) seg = (seg + 1) / 2;
    hipLaunchKernelGGL(stem_pool_f32_kernel, dim3(tiles_y, B, (tiles_x + seg - 1) / seg), dim3(NT), STEM_LDS, (hipStream_t)stream,
                       image, wgt, bias, out, H, W, Hc, Wc, Hp, Wp, tiles_x, seg);
    ML_CHECK_LAUNCH("stem7x7s2_pool_f32");
    return ML_OK;
}
