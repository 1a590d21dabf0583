// ResNeXt grouped 3x3 convolution (c = 4/8/16 channels per group) on the 16-block
// matrix instruction v_mfma_f32_4x4x1_16B_f32.
//
// The reference spells this op as DepthwiseConv2D(depth_multiplier=c) + reshape + reduce_sum
// (engine/backbone/ResNext.py:212-219): out[g*c+m] = sum_{tap,i} x[tap][g*c+i] * K[tap][g*c+i][m].
// A dense 32x32 MFMA tile wastes 8x (c=4) / 4x (c=8) of its work on the zero blocks of the
// block-diagonal weight.  The 4x4x1 form runs 16 INDEPENDENT 4x4 outer products per instruction:
//   block b  <-> 4 consecutive output channels (a quarter/eighth/... of one group)
//   A[i]     = weight of output channel 4b+i for this (tap, input channel) k-step   (lane 4b+i)
//   B[j]     = input pixel j of a 4-pixel quad, same k-step, the block's group       (lane 4b+j)
//   D[i][j]  -> lane 4b+j holds the 4 output channels of pixel j in 4 registers => float4 store.
// So one instruction yields 64 output channels x 4 pixels with no padding waste for any c = 4q.
//
// Block = 256 threads: one spatial tile (TH x TW output pixels) x one 64-channel slab.  The
// input halo tile is staged in LDS once (pixel stride 80 floats: the four pixels of a quad land
// on disjoint bank quarters for ds_read_b128) and read 9 times; weights stream from L1/L2 as one
// float4 per lane per (tap, 4 input channels).  HBM-bound for c = 4, 8 (AI 7-9 flop/B).
#include "common.h"

namespace {

// tensor element type TIO: float, or _Float16 for the fp16-STORAGE mode (BASELINE config 5): 4 channels are then an
// 8-byte load / store, the halo tile in LDS, the weights and all arithmetic stay fp32, the result is rounded once.
typedef _Float16 f16x4g __attribute__((ext_vector_type(4)));
template <class TIO>
__device__ __forceinline__ f32x4 load4(const TIO *p) {
    if constexpr (sizeof(TIO) == 4) {
        return *reinterpret_cast<const f32x4 *>(p);
    } else {
        const f16x4g h = *reinterpret_cast<const f16x4g *>(p);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
}
template <class TIO>
__device__ __forceinline__ void store4(TIO *p, f32x4 v) {
    if constexpr (sizeof(TIO) == 4) {
        *reinterpret_cast<f32x4 *>(p) = v;
    } else {
        *reinterpret_cast<f16x4g *>(p) = f16x4g{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
}

// staging: every thread moves 16 bytes per load -- 4 fp32 or 8 fp16 channels of one halo pixel
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
template <class TIO> struct Stage16 { typedef f32x4 type; };
template <> struct Stage16<_Float16> { typedef f16x8g type; };
template <class TIO>
__device__ __forceinline__ void stage_to_lds(float *dst, const typename Stage16<TIO>::type &v) {
    if constexpr (sizeof(TIO) == 4) {
        *reinterpret_cast<f32x4 *>(dst) = v;
    } else {
        *reinterpret_cast<f32x4 *>(dst) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    }
}

#ifndef GCONV_TH1_HALF
#define GCONV_TH1_HALF 16          // (experiment knob: rows of a stride-1 tile on half tensors)
#endif
constexpr int PS = 80;   // LDS floats per input pixel (64 channels + pad, = 16 mod 64)
constexpr int CS = 64;   // channels per block

template <int STRIDE, int TH, int TW, int CPG, class TIO>
__global__ void __launch_bounds__(256)
gconv_mfma4_kernel(const TIO *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
                   TIO *__restrict__ out, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l, int act,
                   int tiles_x, int tiles_y, int seg) {
    constexpr int c = CPG;
    constexpr int THIN = (TH - 1) * STRIDE + 3;
    constexpr int TWIN = (TW - 1) * STRIDE + 3;
    constexpr int NPIX = THIN * TWIN;
    constexpr int CPT = 16 / (int)sizeof(TIO);  // channels per staging load: 4 (fp32) / 8 (fp16)
    constexpr int TPP = CS / CPT;               // threads per halo pixel
    constexpr int PPP = 256 / TPP;              // pixels per pass of the block
    constexpr int NLD = (NPIX + PPP - 1) / PPP; // staging 16-byte loads per thread
    constexpr int QUADS = TH * TW / 4;          // 4-pixel quads (along x) per tile
    constexpr int QPW = QUADS / 4;              // quads per wave
    constexpr int WV = 9 * c / 4;               // weight float4 per lane
    static_assert(TW % 4 == 0 && QUADS % 4 == 0, "tile must split into quads over 4 waves");
    extern __shared__ __align__(16) float tile[];   // [NPIX][PS]

    // A block walks a run of `seg` tiles DOWN one tile column of its image and 64-channel slab (round 4): its weights are
    // read once, and the next tile's halo is fetched into registers while the current one is multiplied -- a one-tile
    // block spent its life in the chain load -> LDS -> compute -> store with only the co-resident blocks to cover it, and
    // re-read 9 c x 64 weights (as many bytes as the tile itself at c = 16) from L2 for every tile.  Per launch
    // (profiles/r04_gconv_walker_ab.txt): fp32 c = 4 143 -> 124 us, c = 8 stride 2 177 -> 149; half c = 8 179 -> 127, c = 4
    // 219 -> 196; the c = 16 forms +-3 %.  (The first form kept every halo pixel's coordinates and address in registers
    // across the MFMA loop -- 76 -> 150 VGPRs, half the blocks per CU, fp32 c = 4 SLOWER than one-tile blocks; `fetch`
    // now recomputes them per tile from an opaque copy of the thread index.)
    const int tx = blockIdx.x % tiles_x, sg = blockIdx.x / tiles_x;
    const int ty_begin = sg * seg, ty_end = min(tiles_y, ty_begin + seg);
    const int cs0 = blockIdx.y * CS;
    const int b = blockIdx.z;
    const int ox0 = tx * TW;
    const int ix0 = ox0 * STRIDE - pad_l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2;            // 0..15: 4 output channels cs0 + 4*blk + (0..3)
    const int sub = lane & 3;             // A: output channel within the block; B/D: pixel within the quad

    // ---- the halo tile of a tile row (16-byte pieces, zero outside the image) goes to registers first
    typename Stage16<TIO>::type stage[NLD];
    const int cN = (tid % TPP) * CPT;
    auto fetch = [&](int ty) __attribute__((always_inline)) {
        const int iy0 = ty * TH * STRIDE - pad_t;
        int tl = tid;                                     // (opaque copy: the pixel coordinates below are recomputed per tile
        asm volatile("" : "+v"(tl));                     //  instead of living in ~30 registers across the MFMA loop)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tl / TPP + PPP * i;
            const int py = p / TWIN, px = p - py * TWIN;
            const int iy = iy0 + py, ix = ix0 + px;
            typename Stage16<TIO>::type v = {};
            if (p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const typename Stage16<TIO>::type *>(in + ((long long)(b * H + iy) * W + ix) * C + cs0 + (tl % TPP) * CPT);
            stage[i] = v;
        }
    };
    fetch(ty_begin);
    f32x4 wv[WV];
    {
        const float *wrow = wgt + (long long)(cs0 + blk * 4 + sub) * 9 * c;
#pragma unroll
        for (int i = 0; i < WV; ++i) wv[i] = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
    }
    const int gch = ((cs0 + blk * 4) / c) * c - cs0;      // first input channel of the block's group, slab-relative
    int qbase[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        const int quad = wave * QPW + q;
        const int qy = quad / (TW / 4), qx = (quad % (TW / 4)) * 4 + sub;
        qbase[q] = ((qy * STRIDE) * TWIN + qx * STRIDE) * PS + gch;
    }
    const int oc = cs0 + blk * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + oc);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        if (ty > ty_begin) __syncthreads();               // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            if (p < NPIX) stage_to_lds<TIO>(tile + p * PS + cN, stage[i]);
        }
        __syncthreads();
        if (ty + 1 < ty_end) fetch(ty + 1);               // flies under this tile's MFMAs

        f32x4 acc[QPW];
#pragma unroll
        for (int q = 0; q < QPW; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * TWIN + (t % 3)) * PS;
#pragma unroll
            for (int i0 = 0; i0 < c; i0 += 4) {
                const f32x4 w4 = wv[(t * c + i0) / 4];
                f32x4 xv[QPW];
#pragma unroll
                for (int q = 0; q < QPW; ++q) xv[q] = *reinterpret_cast<const f32x4 *>(tile + qbase[q] + toff + i0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < QPW; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4[e], xv[q][e], acc[q], 0, 0, 0);
            }
        }

        // ---- epilogue: lane (blk, sub) owns pixel `sub` of each quad, output channels cs0+4*blk .. +3
        const int oy0 = ty * TH;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const int quad = wave * QPW + q;
            const int oy = oy0 + quad / (TW / 4), ox = ox0 + (quad % (TW / 4)) * 4 + sub;
            if (oy >= Ho || ox >= Wo) continue;
            f32x4 v = acc[q] + bv;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(v[e], act);
            store4<TIO>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + oc, r);
        }
    }
}

// ---- fp16 storage (BASELINE config 5): the same decomposition on v_mfma_f32_4x4x4_16B_f16 -- one instruction contracts
// the 4 input channels that took four 4x4x1 fp32 instructions, operands are the tensor's own halves (the halo tile
// stays fp16 in LDS, staged with plain 16-byte copies) and the weights are rounded to fp16 once per block: the "fp16
// MFMA path" proper.  fp32 accumulation, bias and activation; ONE rounding at the store.
//   A: lane 4b+i holds W[out 4b+i][in i0 .. i0+3]; B: lane 4b+j holds X[pixel j][in i0 .. i0+3]; D as above.
// Pixel stride in LDS: 192 B (stride 1) / 160 B (stride 2) -- the four pixels of a quad, 64 B per half-wave each,
// land on disjoint bank quarters for ds_read_b64.
template <int STRIDE> struct PixH { static constexpr int value = STRIDE == 1 ? 96 : 80; };   // halves per halo pixel

template <int STRIDE, int TH, int TW, int CPG>
__global__ void __launch_bounds__(256)
gconv_mfma4h_kernel(const _Float16 *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
                    _Float16 *__restrict__ out, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l, int act,
                    int tiles_x, int tiles_y, int seg) {
    constexpr int c = CPG;
    constexpr int PSH = PixH<STRIDE>::value;
    constexpr int THIN = (TH - 1) * STRIDE + 3;
    constexpr int TWIN = (TW - 1) * STRIDE + 3;
    constexpr int NPIX = THIN * TWIN;
    constexpr int TPP = CS / 8;                 // threads per halo pixel (8 halves each)
    constexpr int PPP = 256 / TPP;
    constexpr int NLD = (NPIX + PPP - 1) / PPP;
    constexpr int QUADS = TH * TW / 4;
    constexpr int QPW = QUADS / 4;
    constexpr int WV = 9 * c / 4;
    static_assert(TW % 4 == 0 && QUADS % 4 == 0, "tile must split into quads over 4 waves");
    extern __shared__ __align__(16) _Float16 tileh[];   // [NPIX][PSH]

    // (a block walks `seg` tiles down a tile column, weights resident, next halo prefetched: see gconv_mfma4_kernel)
    const int tx = blockIdx.x % tiles_x, sg = blockIdx.x / tiles_x;
    const int ty_begin = sg * seg, ty_end = min(tiles_y, ty_begin + seg);
    const int cs0 = blockIdx.y * CS;
    const int b = blockIdx.z;
    const int ox0 = tx * TW;
    const int ix0 = ox0 * STRIDE - pad_l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, sub = lane & 3;

    f16x8g stage[NLD];
    const int cN = (tid % TPP) * 8;
    auto fetch = [&](int ty) __attribute__((always_inline)) {
        const int iy0 = ty * TH * STRIDE - pad_t;
        int tl = tid;                                     // (opaque copy: the pixel coordinates below are recomputed per tile
        asm volatile("" : "+v"(tl));                     //  instead of living in ~30 registers across the MFMA loop)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tl / TPP + PPP * i;
            const int py = p / TWIN, px = p - py * TWIN;
            const int iy = iy0 + py, ix = ix0 + px;
            f16x8g v = {};
            if (p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const f16x8g *>(in + ((long long)(b * H + iy) * W + ix) * C + cs0 + (tl % TPP) * 8);
            stage[i] = v;
        }
    };
    fetch(ty_begin);
    f16x4g wv[WV];
    {
        const float *wrow = wgt + (long long)(cs0 + blk * 4 + sub) * 9 * c;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
            wv[i] = f16x4g{(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
        }
    }
    const int gch = ((cs0 + blk * 4) / c) * c - cs0;
    int qbase[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        const int quad = wave * QPW + q;
        const int qy = quad / (TW / 4), qx = (quad % (TW / 4)) * 4 + sub;
        qbase[q] = ((qy * STRIDE) * TWIN + qx * STRIDE) * PSH + gch;
    }
    const int oc = cs0 + blk * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + oc);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        if (ty > ty_begin) __syncthreads();               // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            if (p < NPIX) *reinterpret_cast<f16x8g *>(tileh + p * PSH + cN) = stage[i];
        }
        __syncthreads();
        if (ty + 1 < ty_end) fetch(ty + 1);               // flies under this tile's MFMAs

        f32x4 acc[QPW];
#pragma unroll
        for (int q = 0; q < QPW; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * TWIN + (t % 3)) * PSH;
#pragma unroll
            for (int i0 = 0; i0 < c; i0 += 4) {
                const f16x4g w4 = wv[(t * c + i0) / 4];
#pragma unroll
                for (int q = 0; q < QPW; ++q) {
                    const f16x4g xv = *reinterpret_cast<const f16x4g *>(tileh + qbase[q] + toff + i0);
                    acc[q] = __builtin_amdgcn_mfma_f32_4x4x4f16(w4, xv, acc[q], 0, 0, 0);
                }
            }
        }

        const int oy0 = ty * TH;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const int quad = wave * QPW + q;
            const int oy = oy0 + quad / (TW / 4), ox = ox0 + (quad % (TW / 4)) * 4 + sub;
            if (oy >= Ho || ox >= Wo) continue;
            f32x4 v = acc[q] + bv;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(v[e], act);
            store4<_Float16>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + oc, r);
        }
    }
}

// ---- c = 16: one group is a full 16 x 16 block, so v_mfma_f32_16x16x4_f32 wastes nothing either and needs a
// quarter of the operand traffic per flop: D[out 16][pixel 16] += W[out][in 4] * X[in 4][pixel].
//   A: lane (i = l & 15, q = l >> 4) holds W[out i][in 4q + s] for k-step s   (9 float4 per lane, in registers)
//   B: lane (j = l & 15, q)          holds X[pixel j][in 4q + s]              (one ds_read_b128 per tap)
//   D: lane (j, q) holds out channels 4q .. 4q+3 of pixel j                   (one float4 store)
// (the MFMA's k index is permuted -- lane quarter q, step s <-> channel 4q + s -- identically for A and B.)
// Wave w of the block owns group w of the 64-channel slab and walks the tile's 16-pixel sets.
constexpr int PS16 = 68;   // LDS floats per pixel: 16 consecutive pixels x 16 B land on 64 distinct banks

template <int STRIDE, int TH, int TW, class TIO>
__global__ void __launch_bounds__(256)
gconv16_kernel(const TIO *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
               TIO *__restrict__ out, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l, int act, int tiles_x,
               int tiles_y, int seg) {
    constexpr int THIN = (TH - 1) * STRIDE + 3;
    constexpr int TWIN = (TW - 1) * STRIDE + 3;
    constexpr int NPIX = THIN * TWIN;
    constexpr int CPT = 16 / (int)sizeof(TIO);
    constexpr int TPP = CS / CPT;
    constexpr int PPP = 256 / TPP;
    constexpr int NLD = (NPIX + PPP - 1) / PPP;
    constexpr int SETS = TH * TW / 16;          // 16-pixel sets per tile: two output rows of 8
    static_assert(TW == 8 && TH % 2 == 0, "a 16-pixel set is two rows of 8");
    extern __shared__ __align__(16) float tile[];   // [NPIX][PS16]

    // (a block walks `seg` tiles down a tile column, weights resident, next halo prefetched: see gconv_mfma4_kernel)
    const int tx = blockIdx.x % tiles_x, sg = blockIdx.x / tiles_x;
    const int ty_begin = sg * seg, ty_end = min(tiles_y, ty_begin + seg);
    const int cs0 = blockIdx.y * CS;
    const int b = blockIdx.z;
    const int ox0 = tx * TW;
    const int ix0 = ox0 * STRIDE - pad_l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;

    typename Stage16<TIO>::type stage[NLD];
    const int cN = (tid % TPP) * CPT;
    auto fetch = [&](int ty) __attribute__((always_inline)) {
        const int iy0 = ty * TH * STRIDE - pad_t;
        int tl = tid;                                     // (opaque copy: the pixel coordinates below are recomputed per tile
        asm volatile("" : "+v"(tl));                     //  instead of living in ~30 registers across the MFMA loop)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tl / TPP + PPP * i;
            const int py = p / TWIN, px = p - py * TWIN;
            const int iy = iy0 + py, ix = ix0 + px;
            typename Stage16<TIO>::type v = {};
            if (p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const typename Stage16<TIO>::type *>(in + ((long long)(b * H + iy) * W + ix) * C + cs0 + (tl % TPP) * CPT);
            stage[i] = v;
        }
    };
    fetch(ty_begin);
    // weights of group `wave`: out channel cs0 + 16 wave + j, taps 0..8, in channels 4q..4q+3
    f32x4 wv[9];
    {
        const float *wrow = wgt + (long long)(cs0 + wave * 16 + j) * 9 * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4 *>(wrow + t * 16);
    }
    int pbase[SETS];
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        const int py = 2 * s + (j >> 3), px = j & 7;            // output pixel of this lane in set s
        pbase[s] = ((py * STRIDE) * TWIN + px * STRIDE) * PS16 + wave * 16 + 4 * q;
    }
    const int oc = cs0 + wave * 16 + 4 * q;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + oc);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        if (ty > ty_begin) __syncthreads();               // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            if (p < NPIX) stage_to_lds<TIO>(tile + p * PS16 + cN, stage[i]);
        }
        __syncthreads();
        if (ty + 1 < ty_end) fetch(ty + 1);               // flies under this tile's MFMAs

        f32x4 acc[SETS];
#pragma unroll
        for (int s = 0; s < SETS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * TWIN + (t % 3)) * PS16;
            f32x4 xv[SETS];
#pragma unroll
            for (int s = 0; s < SETS; ++s) xv[s] = *reinterpret_cast<const f32x4 *>(tile + pbase[s] + toff);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int s = 0; s < SETS; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][e], xv[s][e], acc[s], 0, 0, 0);
        }

        const int oy0 = ty * TH;
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            const int oy = oy0 + 2 * s + (j >> 3), ox = ox0 + (j & 7);
            if (oy >= Ho || ox >= Wo) continue;
            const f32x4 v = acc[s] + bv;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(v[e], act);
            store4<TIO>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + oc, r);
        }
    }
}

// fp16 storage: c = 16 on v_mfma_f32_16x16x16_f16 -- ONE instruction per tap and 16-pixel set contracts the group's 16
// input channels (four 16x16x4 fp32 instructions before); lane (j, q) supplies the same 4 channels 4q .. 4q+3, as halves.
constexpr int PS16H = 72;  // LDS halves per pixel (144 B)

template <int STRIDE, int TH, int TW>
__global__ void __launch_bounds__(256)
gconv16h_kernel(const _Float16 *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
                _Float16 *__restrict__ out, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l, int act, int tiles_x,
                int tiles_y, int seg) {
    constexpr int THIN = (TH - 1) * STRIDE + 3;
    constexpr int TWIN = (TW - 1) * STRIDE + 3;
    constexpr int NPIX = THIN * TWIN;
    constexpr int TPP = CS / 8;
    constexpr int PPP = 256 / TPP;
    constexpr int NLD = (NPIX + PPP - 1) / PPP;
    constexpr int SETS = TH * TW / 16;
    static_assert(TW == 8 && TH % 2 == 0, "a 16-pixel set is two rows of 8");
    extern __shared__ __align__(16) _Float16 tileh[];   // [NPIX][PS16H]

    // (a block walks `seg` tiles down a tile column, weights resident, next halo prefetched: see gconv_mfma4_kernel)
    const int tx = blockIdx.x % tiles_x, sg = blockIdx.x / tiles_x;
    const int ty_begin = sg * seg, ty_end = min(tiles_y, ty_begin + seg);
    const int cs0 = blockIdx.y * CS;
    const int b = blockIdx.z;
    const int ox0 = tx * TW;
    const int ix0 = ox0 * STRIDE - pad_l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;

    f16x8g stage[NLD];
    const int cN = (tid % TPP) * 8;
    auto fetch = [&](int ty) __attribute__((always_inline)) {
        const int iy0 = ty * TH * STRIDE - pad_t;
        int tl = tid;                                     // (opaque copy: the pixel coordinates below are recomputed per tile
        asm volatile("" : "+v"(tl));                     //  instead of living in ~30 registers across the MFMA loop)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tl / TPP + PPP * i;
            const int py = p / TWIN, px = p - py * TWIN;
            const int iy = iy0 + py, ix = ix0 + px;
            f16x8g v = {};
            if (p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const f16x8g *>(in + ((long long)(b * H + iy) * W + ix) * C + cs0 + (tl % TPP) * 8);
            stage[i] = v;
        }
    };
    fetch(ty_begin);
    f16x4g wv[9];
    {
        const float *wrow = wgt + (long long)(cs0 + wave * 16 + j) * 9 * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + t * 16);
            wv[t] = f16x4g{(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
        }
    }
    int pbase[SETS];
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        const int py = 2 * s + (j >> 3), px = j & 7;
        pbase[s] = ((py * STRIDE) * TWIN + px * STRIDE) * PS16H + wave * 16 + 4 * q;
    }
    const int oc = cs0 + wave * 16 + 4 * q;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + oc);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        if (ty > ty_begin) __syncthreads();               // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            if (p < NPIX) *reinterpret_cast<f16x8g *>(tileh + p * PS16H + cN) = stage[i];
        }
        __syncthreads();
        if (ty + 1 < ty_end) fetch(ty + 1);               // flies under this tile's MFMAs

        f32x4 acc[SETS];
#pragma unroll
        for (int s = 0; s < SETS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * TWIN + (t % 3)) * PS16H;
#pragma unroll
            for (int s = 0; s < SETS; ++s) {
                const f16x4g xv = *reinterpret_cast<const f16x4g *>(tileh + pbase[s] + toff);
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x16f16(wv[t], xv, acc[s], 0, 0, 0);
            }
        }

        const int oy0 = ty * TH;
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            const int oy = oy0 + 2 * s + (j >> 3), ox = ox0 + (j & 7);
            if (oy >= Ho || ox >= Wo) continue;
            const f32x4 v = acc[s] + bv;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(v[e], act);
            store4<_Float16>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + oc, r);
        }
    }
}

// fp16 storage, c = 32 (the last ResNeXt stage: 32 groups of 32 channels, ResNext.py:212-219 with filters = 1024): a group is a
// full 32 x 32 block of v_mfma_f32_32x32x16_f16 -- two instructions per tap contract its 32 input channels.
//   A: lane (m = l & 31, q = l >> 5) holds W[out m][in 16 kh + 8 q .. + 7]     (9 taps x 2 k-halves x 8 halves, in registers)
//   B: lane (p = l & 31, q)          holds X[pixel p][in 16 kh + 8 q .. + 7]    (one ds_read_b128 per tap and k-half)
//   D: lane (p, q) holds out channels (e & 3) + 8 (e >> 2) + 4 q of pixel p     (four 8-byte stores)
// Block = a COLUMN of 8 x 8-pixel tiles x one 64-channel slab (two groups): wave w multiplies group w & 1 by the 32 pixels of
// tile rows 4 (w >> 1) .. + 3, tile after tile, its weights staying in registers.  (Round 3 ran this stage through an fp32 copy of the input on the dense kernel's 32-wide
// block-diagonal tiles: a cast launch + a conv on operands converted in registers.)
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));

template <int STRIDE>
__global__ void __launch_bounds__(256)
gconv32h_kernel(const _Float16 *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
                _Float16 *__restrict__ out, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l, int act, int tiles_y) {
    constexpr int TH = 8, TW = 8;
    constexpr int THIN = (TH - 1) * STRIDE + 3;
    constexpr int TWIN = (TW - 1) * STRIDE + 3;
    constexpr int NPIX = THIN * TWIN;
    constexpr int TPP = CS / 8;                 // threads per halo pixel (8 halves each)
    constexpr int PPP = 256 / TPP;
    constexpr int NLD = (NPIX + PPP - 1) / PPP;
    extern __shared__ __align__(16) _Float16 tileh[];   // [NPIX][PS16H]  (144 B per pixel: 32 consecutive pixels x 16 B hit distinct bank groups)

    // one block = one COLUMN of 8 x 8 tiles of one image and slab: the group's 9 x 32 x 32 weights (36 KB per wave, more than
    // a tile's 12.8 KB of input) are fetched once and stay in registers while the block walks down the column
    const int tx = blockIdx.x;
    const int cs0 = blockIdx.y * CS;
    const int b = blockIdx.z;
    const int ox0 = tx * TW;
    const int ix0 = ox0 * STRIDE - pad_l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p32 = lane & 31, q = lane >> 5;
    const int grp = wave & 1, half = wave >> 1;

    // weights of group `grp` of the slab: out channel cs0 + 32 grp + p32, tap t, in channels 16 kh + 8 q .. + 7
    f16x8h wv[9][2];
    {
        const float *wrow = wgt + (long long)(cs0 + grp * 32 + p32) * 9 * 32 + 8 * q;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wrow + t * 32 + kh * 16);
                const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wrow + t * 32 + kh * 16 + 4);
                wv[t][kh] = f16x8h{(_Float16)w0[0], (_Float16)w0[1], (_Float16)w0[2], (_Float16)w0[3],
                                   (_Float16)w1[0], (_Float16)w1[1], (_Float16)w1[2], (_Float16)w1[3]};
            }
    }
    f32x4 bv[4];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4)
        bv[e4] = bias ? *reinterpret_cast<const f32x4 *>(bias + cs0 + grp * 32 + 8 * e4 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int py = 4 * half + (p32 >> 3), px = p32 & 7;             // this lane's output pixel inside a tile
    const int pbase = ((py * STRIDE) * TWIN + px * STRIDE) * PS16H + grp * 32 + 8 * q;
    const int cN = (tid % TPP) * 8;

    for (int ty = 0; ty < tiles_y; ++ty) {
        const int oy0 = ty * TH;
        const int iy0 = oy0 * STRIDE - pad_t;
        f16x8g stage[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            const int ppy = p / TWIN, ppx = p - ppy * TWIN;
            const int iy = iy0 + ppy, ix = ix0 + ppx;
            f16x8g v = {};
            if (p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const f16x8g *>(in + ((long long)(b * H + iy) * W + ix) * C + cs0 + cN);
            stage[i] = v;
        }
        if (ty > 0) __syncthreads();                                // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid / TPP + PPP * i;
            if (p < NPIX) *reinterpret_cast<f16x8g *>(tileh + p * PS16H + cN) = stage[i];
        }
        __syncthreads();

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * TWIN + (t % 3)) * PS16H;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const f16x8h xv = *reinterpret_cast<const f16x8h *>(tileh + pbase + toff + kh * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[t][kh], xv, acc, 0, 0, 0);
            }
        }
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < Ho && ox < Wo) {
            _Float16 *orow = out + ((long long)(b * Ho + oy) * Wo + ox) * C + cs0 + grp * 32;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {                        // channels 8 e4 + 4 q .. + 3 of the group: registers 4 e4 .. + 3
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(acc[4 * e4 + e] + bv[e4][e], act);
                store4<_Float16>(orow + 8 * e4 + 4 * q, r);
            }
        }
    }
}

template <int STRIDE>
int launch32h(const _Float16 *in, const float *wgt, const float *bias, _Float16 *out, int B, int H, int W, int C, int Ho, int Wo,
              int pad_t, int pad_l, int act, hipStream_t s) {
    constexpr int THIN = 7 * STRIDE + 3, TWIN = 7 * STRIDE + 3;
    constexpr int LDS_BYTES = THIN * TWIN * PS16H * 2;
    auto kern = gconv32h_kernel<STRIDE>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, lds_ok, "gconv3x3")) return rc;
    const int tiles_x = (Wo + 7) / 8, tiles_y = (Ho + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(tiles_x, C / CS, B), dim3(256), LDS_BYTES, s, in, wgt, bias, out, H, W, C, Ho, Wo,
                       pad_t, pad_l, act, tiles_y);
    ML_CHECK_LAUNCH("gconv3x3");
    return ML_OK;
}

// Tiles a block walks down its tile column: the longest run that still leaves GCONV_BLOCKS_PER_CU blocks per CU (the blocks
// of a CU cover each other's store / barrier phases; a launch of few long blocks would leave CUs idle at its end)
#ifndef GCONV_BLOCKS_PER_CU
#define GCONV_BLOCKS_PER_CU 6
#endif
static int column_run(int tiles_x, int tiles_y, int slabs, int B) {
    const long long want = (long long)GCONV_BLOCKS_PER_CU * ml_resident_blocks(1);
    int seg = tiles_y;
    while (seg > 1 && (long long)tiles_x * ((tiles_y + seg - 1) / seg) * slabs * B < want) seg = (seg + 1) / 2;
    return seg;
}

// kernel of a tensor type: fp32 tensors -> the fp32 MFMA forms, fp16 tensors -> the fp16 MFMA forms
template <int STRIDE, int TH, int TW, class TIO>
auto pick16() {
    if constexpr (sizeof(TIO) == 2) return gconv16h_kernel<STRIDE, TH, TW>;
    else return gconv16_kernel<STRIDE, TH, TW, TIO>;
}
template <int STRIDE, int TH, int TW, int CPG, class TIO>
auto pick4() {
    if constexpr (sizeof(TIO) == 2) return gconv_mfma4h_kernel<STRIDE, TH, TW, CPG>;
    else return gconv_mfma4_kernel<STRIDE, TH, TW, CPG, TIO>;
}

template <int STRIDE, int TH, int TW, class TIO>
int launch16(const TIO *in, const float *wgt, const float *bias, TIO *out, int B, int H, int W, int C, int Ho, int Wo,
             int pad_t, int pad_l, int act, hipStream_t s) {
    constexpr int THIN = (TH - 1) * STRIDE + 3, TWIN = (TW - 1) * STRIDE + 3;
    constexpr bool HALF = sizeof(TIO) == 2;
    constexpr int LDS_BYTES = HALF ? THIN * TWIN * PS16H * 2 : THIN * TWIN * PS16 * 4;
    auto kern = pick16<STRIDE, TH, TW, TIO>();
    static std::atomic<unsigned long long> lds_ok{0};      // per kernel instantiation, one bit per device
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, lds_ok, "gconv3x3")) return rc;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int seg = column_run(tiles_x, tiles_y, C / CS, B);
    hipLaunchKernelGGL(kern, dim3(tiles_x * ((tiles_y + seg - 1) / seg), C / CS, B), dim3(256), LDS_BYTES, s, in, wgt, bias, out, H,
                       W, C, Ho, Wo, pad_t, pad_l, act, tiles_x, tiles_y, seg);
    ML_CHECK_LAUNCH("gconv3x3");
    return ML_OK;
}

template <int STRIDE, int TH, int TW, int CPG, class TIO>
int launch(const TIO *in, const float *wgt, const float *bias, TIO *out, int B, int H, int W, int C, int Ho,
           int Wo, int pad_t, int pad_l, int act, hipStream_t s) {
    constexpr int THIN = (TH - 1) * STRIDE + 3, TWIN = (TW - 1) * STRIDE + 3;
    constexpr bool HALF = sizeof(TIO) == 2;
    constexpr int LDS_BYTES = HALF ? THIN * TWIN * PixH<STRIDE>::value * 2 : THIN * TWIN * PS * 4;
    auto kern = pick4<STRIDE, TH, TW, CPG, TIO>();
    static std::atomic<unsigned long long> lds_ok{0};      // per kernel instantiation, one bit per device
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, lds_ok, "gconv3x3")) return rc;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int seg = column_run(tiles_x, tiles_y, C / CS, B);
    hipLaunchKernelGGL(kern, dim3(tiles_x * ((tiles_y + seg - 1) / seg), C / CS, B), dim3(256), LDS_BYTES, s, in, wgt, bias, out, H,
                       W, C, Ho, Wo, pad_t, pad_l, act, tiles_x, tiles_y, seg);
    ML_CHECK_LAUNCH("gconv3x3");
    return ML_OK;
}

}  // namespace

template <class TIO>
static int gconv3x3_any(const TIO *in, const float *wgt, const float *bias, TIO *out, int32_t B, int32_t H, int32_t W,
                        int32_t C, int32_t c, int32_t Ho, int32_t Wo, int32_t stride, int32_t pad_t, int32_t pad_l,
                        int32_t act, void *stream) {
    ML_REQUIRE(in && wgt && out, "gconv3x3: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "gconv3x3: bad dims");
    ML_REQUIRE(C > 0 && C % CS == 0 && C / CS < 65536, "gconv3x3: channels must be a multiple of %d", CS);
    ML_REQUIRE(c == 4 || c == 8 || c == 16 || (c == 32 && sizeof(TIO) == 2),
               "gconv3x3: channels per group %d must be 4, 8 or 16 (half tensors: also 32; fp32 groups of 32: ml_conv2d_f32 "
               "with group_cin_step)", c);
    ML_REQUIRE(C % c == 0, "gconv3x3: C must be a multiple of c");
    ML_REQUIRE(stride == 1 || stride == 2, "gconv3x3: stride must be 1 or 2");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(wgt) && ml_aligned16(out) && (!bias || ml_aligned16(bias)),
               "gconv3x3: pointers must be 16-byte aligned");
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "gconv3x3: too many pixels");
    hipStream_t s = (hipStream_t)stream;
#define GC_ARGS in, wgt, bias, out, B, H, W, C, Ho, Wo, pad_t, pad_l, act, s
    // Tile heights: half tensors take tiles twice as tall as fp32 ones.  A halo pixel of a 64-channel slab is ONE 128-byte
    // line in half (two in fp32), so a half block of the fp32 tile shape moves half the bytes per request and per block
    // prologue; 16 x 8 (stride 2: 8 x 8) tiles cut the halo share from 56 % to 41 % of a patch and measured 9-29 % faster
    // on the 16 x 1280^2 ResNeXt-101 shapes (gpurun_out/r04g_gconv_*.txt), while fp32 tensors lose 2-14 % with them
    // (22-46 KB of LDS per block: fewer blocks per CU to hide the load -> LDS -> compute chain of a block).
    constexpr bool HALF = sizeof(TIO) == 2;
    constexpr int TH1 = HALF ? GCONV_TH1_HALF : 8, TH2 = HALF ? 8 : 4;
    if constexpr (HALF) {
        if (c == 32) return stride == 1 ? launch32h<1>(GC_ARGS) : launch32h<2>(GC_ARGS);
    }
    if (stride == 1) {
        if (c == 4) return launch<1, TH1, 8, 4, TIO>(GC_ARGS);
        if (c == 8) return launch<1, TH1, 8, 8, TIO>(GC_ARGS);
        return launch16<1, TH1, 8, TIO>(GC_ARGS);
    }
    if (c == 4) return launch<2, TH2, 8, 4, TIO>(GC_ARGS);
    if (c == 8) return launch<2, TH2, 8, 8, TIO>(GC_ARGS);
    return launch16<2, TH2, 8, TIO>(GC_ARGS);
#undef GC_ARGS
}

extern "C" int ml_gconv3x3_f32(const float *in, const float *wgt, const float *bias, float *out, int32_t B, int32_t H,
                               int32_t W, int32_t C, int32_t c, int32_t Ho, int32_t Wo, int32_t stride, int32_t pad_t,
                               int32_t pad_l, int32_t act, void *stream) {
    return gconv3x3_any<float>(in, wgt, bias, out, B, H, W, C, c, Ho, Wo, stride, pad_t, pad_l, act, stream);
}

extern "C" int ml_gconv3x3_f16(const void *in, const float *wgt, const float *bias, void *out, int32_t B, int32_t H,
                               int32_t W, int32_t C, int32_t c, int32_t Ho, int32_t Wo, int32_t stride, int32_t pad_t,
                               int32_t pad_l, int32_t act, void *stream) {
    return gconv3x3_any<_Float16>(reinterpret_cast<const _Float16 *>(in), wgt, bias, reinterpret_cast<_Float16 *>(out), B, H,
                                  W, C, c, Ho, Wo, stride, pad_t, pad_l, act, stream);
}
