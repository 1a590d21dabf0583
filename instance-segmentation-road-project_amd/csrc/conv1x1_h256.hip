// 1x1 convolution on IEEE-half tensors as a 256 x 256-tile GEMM:  out[M, N] = act(in[M, K] * wgt[N, K]^T + bias + residual)
// for the K >= 256 bottleneck convs of an fp16-STORAGE ResNeXt body (BASELINE configs[4]: ResNeXt-101 stage 3 is 46
// launches of M = 102 400, K / N = 1024 / 512 and 512 / 1024; thirdparty/classification_models/models/resnext.py:62-135).
//
// Why a second kernel beside conv1x1_pipe.hip: on half tensors a 128 x 128 tile with 64 x 64 per wave stages 32 KB for
// 16 MFMAs per wave and chunk.  Here a block is 8 waves (2 per SIMD) on a 256 x 256 tile, 128 x 64 per wave: the same 8
// LDS-direct requests per wave and chunk feed 32 MFMAs, every staged byte is used for twice the MACs.
//   * K chunk = 64 halves = 128 bytes per row (full cache lines; XOR-swizzled 16-byte groups as in the other kernels);
//   * ROUND 4 -- LDS is a RING: three 32 KB slots for the activation (A) chunks and two for the weight (B) chunks, all
//     160 KB of a CU.  While chunk c is multiplied the block requests B of chunk c + 1 and A of chunk c + 2; every wait is
//     a counted `s_waitcnt vmcnt(N)` (requests retire in issue order).  Round 3 had two 64 KB buffers, requests one chunk
//     ahead and `vmcnt(0)` per chunk;
//   * a block is PERSISTENT over ONE contiguous range of rows (cut at multiples of 32: all ranges of a launch are equal
//     +- 32 rows) and one N tile; the chunk stream crosses tile boundaries, so the next tile's first chunks are in flight
//     during the epilogue.  The last tile of a range may be short; up to 128 rows it takes a LIGHT path -- the same chunk
//     stream, but wave w multiplies columns 32 w .. + 31 by the Q <= 4 row groups that exist (Q MFMAs per 16-deep step
//     instead of 8).  Round 3 dealt whole 256-row panels round-robin: 3.125 tile rounds cost 4;
//   * all N tiles of a row range run at the same time on ONE XCD (blocks with equal blockIdx & 7): the rows come from
//     beyond L2 once (TCC_EA0_RDREQ = the activation bytes, profiles/r04_h256_pmc.md);
//   * epilogue: accumulators -> wave-private LDS scratch (16 rows x 32 columns at a time; it lives in the A slot the tile's
//     last chunk freed) -> rows of 8 halves per lane, + bias + half residual, clamp, one rounding, 16-byte stores; B of the
//     next tile's chunk 1 is requested BEFORE the stores, so two chunks of the next tile overlap the store drain.
// Measured (16 x 1280^2 ResNeXt-101 shapes, outputs bit-identical to conv1x1_pipe_kernel<_Float16> and to round 3's kernel,
// 0 mismatches in 30 repeated launches per shape; gpurun_out/r04e_h256_tail.txt, profiles/r04_h256_ab.txt), round 3 ->
// ring -> ring + light tail:  1024->512  133.0 -> 124.7 -> 116.5 us;  512->1024 + residual  179.7 -> 160.2 -> 154.5;
// 512->256  163.6 -> 150.1 -> 146.8;  256->512 + residual  245.2 -> 233.5 -> 234.0;  2048->1024  120.9 -> 114.9 -> 114.0;
// 1024->2048 + residual  141.4 -> 139.5 -> 125.8;  the mode's 62 launches 10.3 -> 8.6 ms (667 -> 797 TF).
// What bounds it now (profiles/r04_h256_pmc.md, profiles/r04_h256_request_ablations.txt): the matrix pipe is busy 45 % of the
// launch (SQ_VALU_MFMA_BUSY_CYCLES = exactly the 32-cycle MFMAs the FLOPs need), the texture-address / L1 path 50-62 %
// (TA_TA_BUSY: ~26 cycles per 1 KB LDS-direct request), waves wait 39 % of their cycles; request latencies are SHORT --
// 430-500 cycles L1 -> L2 on average, 950-1 260 from L2 to the fabric -- so more requests in flight buy nothing more, and
// with both operands through empty extents (same instructions, no bytes) the launch still takes 77 % of its time: barrier
// + fragment-read + epilogue structure.  A 256 x 256 half tile moves 64 KB per chunk through a per-CU L1 path that
// delivers ~30-39 B/clk while it is busy, i.e. ~1 700-2 200 cycles beside 2 048 cycles of MFMA: the two are of equal
// size and only partly overlap.  Tried and measured no better in round 4: the 8 requests spread over every 2nd / 4th MFMA
// (+-1 %); the two waves of a SIMD requesting in opposite halves of the chunk (+-2 %); `s_setprio 1` for waves 4-7
// (+-1 %); a guard around every MFMA to skip the sub-tiles of a short tile (216 vs 134 us: branches in the matrix stream).
// Round 3's notes that still hold: blocks started a quarter tile apart (s_sleep) no better; a ring of four 32-deep
// chunks slower (64-byte row pieces = half cache lines, twice the requests); padding the row pitch off a power of two
// changes nothing (not an L2-channel effect).
// Numerics: fp16 products are exact in fp32, fp32 accumulation in k order per output, bias / residual / clamp in fp32 --
// the same arithmetic as conv1x1_pipe_kernel<_Float16> (which chains 64-deep chunks the same way).
#include <type_traits>
#include "common.h"

namespace {

// experiment knobs (scripts/experiments): spacing of the LDS-direct requests in the MFMA stream, and ablations that send the
// A / B requests through an empty extent (same instructions, no traffic; results are then wrong)
#ifndef H256_DMA_STRIDE
#define H256_DMA_STRIDE 1
#endif
#ifndef H256_ABL_A
#define H256_ABL_A 0
#endif
#ifndef H256_ABL_B
#define H256_ABL_B 0
#endif
constexpr int TM = 256, TN = 256;                  // block tile
constexpr int ROWB = 128;                          // bytes of K per staged row and chunk (64 halves)
constexpr int ABUF = TM * ROWB, BBUF = TN * ROWB;  // one chunk of the A tile / of the B tile: 32 KB each
constexpr int SCR_LD = 36;                         // floats per row of a wave's transposition scratch
constexpr int SCRB = 16 * SCR_LD * 4;              // (8 waves x 2 304 B: lives in the A slot the tile's last chunk freed)
constexpr int H256_LDS = 3 * ABUF + 2 * BBUF;      // 163 840 B = all of a CU's LDS: a ring of 3 A slots + 2 B slots
static_assert(8 * SCRB <= ABUF, "the epilogue scratch must fit an A slot");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct H256Args {
    const void *in, *wgt, *res;
    const float *bias;
    void *out;
    long long M;
    int K, N;
    int in_cs, in_coff, out_cs, out_coff;    // elements; the residual shares out_cs / out_coff
    int G, gshift;                           // N tiles = G = 1 << gshift
    int nb;                                  // row ranges = blocks per N tile = grid >> gshift
    int g_base, g_rem;                       // 32-row groups per range: g_base, the first g_rem ranges one more
    int xcd_map;
    float lo, hi;
    int clamp;
};

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ unsigned records_left(long long off, long long total) {
    const unsigned long long left = (unsigned long long)(total - off);
    const unsigned hi = (unsigned)(left >> 32), lo = (unsigned)left;
    return (hi & 0x80000000u) ? 0u : (hi ? 0xffffffffu : lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void *ptr, long long off, long long total) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)ptr + off), 0, (int)records_left(off, total), 0x00020000);
}
// `nbytes` bytes from ptr + off (the rows of ONE tile: anything past them reads zeros / is not stored, no traffic)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_span(const void *ptr, long long off, int nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)ptr + off), 0, nbytes, 0x00020000);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// the same descriptor as four plain dwords, for inline-asm operands
__device__ __forceinline__ i32x4 rsrc_words(const void *ptr, long long off, long long total) {
    const unsigned long long base = (unsigned long long)((const char *)ptr + off);
    const i32x4 w = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)records_left(off, total), 0x00020000};
    return w;
}
// Register-destination load hipcc must not see: beside LDS-direct loads in flight it would wait vmcnt(0) before every
// use of an ordinary load's result -- i.e. for the previous piece's STORES too.  Completion is waited for by hand.
__device__ __forceinline__ f32x4 buf_load16_asm(i32x4 rsrc, int voff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_loaded4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

template <bool HAS_RES>
__global__ void __launch_bounds__(512, 2)
conv1x1_h256_kernel(const H256Args A) {
    extern __shared__ __align__(16) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                   // wave tile: rows 128 wr .., columns 64 wc ..
    const int r = lane & 31, h = lane >> 5;
    const int swz = (r >> 1) & 7;

    // ---- staging: wave w, piece i (0..3) covers rows 64 i + 8 w .. + 7 of the A tile and of the B tile;
    // lane -> row lane >> 3, 16-byte group (lane & 7) ^ swizzle key of that row
    const int ld_r = 8 * wave + (lane >> 3);                   // + 64 i
    const int ld_g = (lane & 7) ^ ((ld_r >> 1) & 7);           // (64 i does not change the key)
    int a_voff[4], b_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_voff[i] = ((ld_r + 64 * i) * A.in_cs + A.in_coff) * 2 + ld_g * 16;
        b_voff[i] = (ld_r + 64 * i) * A.K * 2 + ld_g * 16;
    }
    const int a_off = (wr * 128 + r) * ROWB;                   // fragment rows of this lane, inside an A slot / a B slot
    const int b_off = (wc * 64 + r) * ROWB;
    char *const lds_b = lds + 3 * ABUF;                        // [A slot 0][A slot 1][A slot 2][B slot 0][B slot 1]

    // ---- epilogue ownership inside a 16-row x 32-column piece: lane -> row lane >> 2, columns 8 (lane & 3) .. + 7
    const int t_row = lane >> 2, t_col = (lane & 3) * 8;
    const int scr_w = (4 * h * SCR_LD + r) * 4;
    const int scr_r = (t_row * SCR_LD + t_col) * 4;

    const int nk = A.K / 64;
    const long long w_total = (long long)A.N * A.K * 2;

    // ---- which rows: one N tile for the block's life and ONE contiguous range of rows, cut at multiples of 32 so that
    // all ranges of a launch differ by at most 32 rows (a 1x1 conv's rows are independent: where the cut falls changes no
    // result).  The range is walked in 256-row tiles; the last may be short (rows past it read zeros and are not stored).
    const int bid = (int)blockIdx.x;
    const int gmask = A.G - 1;
    const int slot = bid >> 3, ppx = A.nb >> 3;
    const int nt = A.xcd_map ? (slot & gmask) : (bid & gmask);
    const int ri = A.xcd_map ? (bid & 7) * ppx + (slot >> A.gshift) : (bid >> A.gshift);
    const int g0 = ri * A.g_base + min(ri, A.g_rem);
    const int gcnt = A.g_base + (ri < A.g_rem ? 1 : 0);
    long long row = (long long)g0 * 32;
    const long long row_end = min((long long)(g0 + gcnt) * 32, A.M);
    if (row >= row_end) return;
    int rows = (int)min((long long)TM, row_end - row);

    const __amdgpu_buffer_rsrc_t rb = rsrc_at(A.wgt, (long long)nt * TN * A.K * 2, w_total);
    const __amdgpu_buffer_rsrc_t r_none = rsrc_at(A.wgt, 0, 0);         // empty extent: the request is counted, nothing moves
    auto res_a = [&](long long r0, int nrows) {              // rows r0 .. r0 + nrows - 1 of the input (nrows = 0: empty)
        return rsrc_span(A.in, nrows > 0 ? r0 * A.in_cs * 2 : 0, nrows * A.in_cs * 2);
    };
    auto stage_a = [&](__amdgpu_buffer_rsrc_t ra, char *dst, int soff) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ra, dst + (64 * i + 8 * wave) * ROWB, a_voff[i], soff);
    };
    auto stage_b = [&](__amdgpu_buffer_rsrc_t rbb, char *dst, int soff) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rbb, dst + (64 * i + 8 * wave) * ROWB, b_voff[i], soff);
    };

    // bias of this wave's 64 columns, in the epilogue's column ownership: 2 sub-tiles x 8 floats
    f32x4 bias[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = nt * TN + wc * 64 + ni * 32 + t_col + 4 * k;
            bias[ni][k] = A.bias ? *reinterpret_cast<const f32x4 *>(A.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // ---- the chunk stream.  Chunk c of the block (tiles back to back, nk chunks each) reads A slot c % 3 and B slot c % 2.
    // While it is computed the block requests B of chunk c + 1 and A of chunk c + 2: the activations -- which come from
    // beyond L2 and need ~2 us under load -- fly for two chunk times, the weights (L2 hits) for one.  With two 64 KB
    // buffers (round 3) at most 64 KB per CU were in flight, and 64 KB per ~2.2 us is exactly the ~29 GB/s per CU that
    // kernel stood at; here it is 64 KB of A + 32 KB of B.  Vector-memory operations retire in issue order, so every wait is
    // a COUNT of what may stay in flight behind the requests that are needed.
    struct Chunk {
        const char *rd_a, *rd_b;                 // this chunk's A / B slot
        char *wr_a, *wr_b;                       // where A(c + 2) / B(c + 1) land
        __amdgpu_buffer_rsrc_t ra_nx, rb_nx;
        int soff_a, soff_b;
        bool send_b;
    };
    __amdgpu_buffer_rsrc_t ra = res_a(row, rows);
    stage_b(rb, lds_b, 0);                                             // B(0)
    stage_a(ra, lds, 0);                                               // A(0)
    stage_a(ra, lds + ABUF, ROWB);                                     // A(1)     (nk >= 4: always this tile's)
    int sa = 0, sb = 0;                                                // slots of the current chunk
    bool first_tile = true;
    __amdgpu_buffer_rsrc_t ra_next = r_none, rb_next = r_none;         // the tile after the current one (empty: none)

    auto begin_chunk = [&](int kc) __attribute__((always_inline)) {
        // what must have landed: A(c) and B(c).  In flight behind them, oldest first --
        //   ordinary chunk:                  A(c+1)                                   -> 4 may stay
        //   chunk 0 of a later tile:         A(c+1), B(c+1) [sent before the stores], the 16 stores  -> 24
        //   chunk 1 of a later tile:         the 16 stores, A(c+1)  [chunk 0 sent A only]            -> 20
        //   chunk 2 of a later tile needs B(c), which is YOUNGER than the stores: the store drain of a tile overlaps
        //   the next tile's first two chunks and no more.
        // (with a residual the epilogue's own waits have retired everything older than its stores: the counts hold)
        if (!first_tile && kc == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (!first_tile && kc == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // ... everyone's; and chunk c - 1's slots are free
        Chunk C;
        C.rd_a = lds + sa * ABUF;
        C.rd_b = lds_b + sb * BBUF;
        const int sa2 = sa == 0 ? 2 : sa - 1;                       // (c + 2) % 3 == (c - 1) % 3
        C.wr_a = lds + sa2 * ABUF;
        C.wr_b = lds_b + (sb ^ 1) * BBUF;
        // sources of B(c+1) and A(c+2): this tile, or the head of the next one (empty extents past the block's end)
        const bool b_here = kc + 1 < nk, a_here = kc + 2 < nk;
        C.rb_nx = b_here ? rb : rb_next;
        C.ra_nx = a_here ? ra : ra_next;
        C.soff_b = b_here ? (kc + 1) * ROWB : 0;
        C.soff_a = a_here ? (kc + 2) * ROWB : (kc + 2 - nk) * ROWB;
        C.send_b = first_tile || kc > 0;                            // (B of a later tile's chunk 1 went out before the stores)
        return C;
    };
    auto send_piece = [&](const Chunk &C, auto pc) __attribute__((always_inline)) {     // piece 0..7: B first (needed first)
        constexpr int p = decltype(pc)::value;
        if constexpr (p < 4) {
            if (C.send_b) lds_dma16(H256_ABL_B ? r_none : C.rb_nx, C.wr_b + (64 * p + 8 * wave) * ROWB, b_voff[p], C.soff_b);
        } else {
            lds_dma16(H256_ABL_A ? r_none : C.ra_nx, C.wr_a + (64 * (p - 4) + 8 * wave) * ROWB, a_voff[p - 4], C.soff_a);
        }
    };
    auto end_chunk = [&]() __attribute__((always_inline)) {
        sa = sa == 2 ? 0 : sa + 1;
        sb ^= 1;
    };
    // between tiles: every wave has read the last chunk's slots (A slot `sl`, B slot sb ^ 1 -- `sa` / `sb` already name the
    // next chunk's).  B of the next tile's chunk 1 goes into that B slot NOW, ahead of the stores, and the A slot serves as
    // the waves' transposition scratch until the next chunk's barrier (its next request comes after that barrier).
    auto between_tiles = [&]() __attribute__((always_inline)) {
        const int sl = sa == 0 ? 2 : sa - 1;
        __builtin_amdgcn_s_barrier();
        stage_b(rb_next, lds_b + (sb ^ 1) * BBUF, ROWB);
        return lds + sl * ABUF + wave * SCRB;
    };
    // one 16-row x 32-column piece of the epilogue: 8 accumulator registers of a 32 x 32 tile -> the wave's scratch ->
    // rows of 8 consecutive columns per lane, + residual (8 halves) + bias, clamp, one rounding, one 16-byte store
    auto finish_piece = [&](char *scratch, const f32x16 &acc, int hs, const f32x4 &res8, const f32x4 &b0, const f32x4 &b1,
                            __amdgpu_buffer_rsrc_t ro, int voff) __attribute__((always_inline)) {
        float *sw = reinterpret_cast<float *>(scratch + scr_w);
        const f32x4 *sr = reinterpret_cast<const f32x4 *>(scratch + scr_r);
        // C/D layout (col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) -> scratch[row][col]
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const int e = hs * 8 + e8;
            sw[((e & 3) + 8 * ((e >> 2) & 1)) * SCR_LD] = acc[e];
        }
        f32x4 v0 = sr[0], v1 = sr[1];
        if constexpr (HAS_RES) {
            const f16x8 rh = __builtin_bit_cast(f16x8, res8);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v0[c] += (float)rh[c]; v1[c] += (float)rh[4 + c]; }
        }
        v0 += b0;
        v1 += b1;
        if (A.clamp) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v0[c] = __builtin_amdgcn_fmed3f(v0[c], A.lo, A.hi);
                v1[c] = __builtin_amdgcn_fmed3f(v1[c], A.lo, A.hi);
            }
        }
        const f16x8 o = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                         (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, voff, 0, 0);
    };
    auto out_rsrc = [&](long long r0, int nrows, __amdgpu_buffer_rsrc_t &ro, i32x4 &rr) __attribute__((always_inline)) {
        const long long tile_off = (r0 * A.out_cs + A.out_coff + (long long)nt * TN) * 2;
        const int tile_bytes = nrows * A.out_cs * 2 - (A.out_coff + nt * TN) * 2;
        ro = rsrc_span(A.out, tile_off, tile_bytes);
        const unsigned long long rbase = (unsigned long long)((const char *)A.res + (HAS_RES ? tile_off : 0));
        rr = i32x4{(int)(unsigned)rbase, (int)((unsigned)(rbase >> 32) & 0xffffu), HAS_RES ? tile_bytes : 0, 0x00020000};
    };

    // ---- whole tiles (and short last tiles of more than 128 rows, which run the same stream on zero-filled rows)
    int tail_q = 0;                                       // 32-row groups of a short last tile that takes the light path below
    for (;;) {
        const long long next_row = row + TM;
        const int next_rows = next_row < row_end ? (int)min((long long)TM, row_end - next_row) : 0;
        const bool has_next = next_rows > 0;
        ra_next = res_a(next_row, next_rows);                          // (empty past the end: no traffic)
        rb_next = has_next ? rb : r_none;
        f32x16 acc[4][2];
        for (int kc = 0; kc < nk; ++kc) {
            const Chunk C = begin_chunk(kc);
            const char *rd_a = C.rd_a + a_off, *rd_b = C.rd_b + b_off;
            // 4 k-steps of 16: 6 fragment reads + 8 MFMAs each; the 8 requests ride behind the first MFMAs
            f32x4 fa[2][4], fb[2][2];
            auto read_frags = [&](int ks, f32x4 (&a)[4], f32x4 (&b)[2]) __attribute__((always_inline)) {
                const int slot16 = ((ks * 2 + h) ^ swz) * 16;
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const f32x4 *>(rd_a + m * 32 * ROWB + slot16);
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const f32x4 *>(rd_b + n * 32 * ROWB + slot16);
            };
            read_frags(0, fa[0], fb[0]);
            if (kc == 0) {                                             // a tile's chains start from 0 (once per tile)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
            }
            static_for<0, 32>([&](auto ic) {
                constexpr int idx = decltype(ic)::value;               // ks * 8 + mi * 2 + ni
                constexpr int ks = idx >> 3, mi = (idx >> 1) & 3, ni = idx & 1;
                if constexpr ((idx & 7) == 2 && ks < 3) read_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                const f16x8 a = __builtin_bit_cast(f16x8, fa[ks & 1][mi]), b = __builtin_bit_cast(f16x8, fb[ks & 1][ni]);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[mi][ni], 0, 0, 0);
                if constexpr (idx % H256_DMA_STRIDE == 0 && idx / H256_DMA_STRIDE < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    send_piece(C, std::integral_constant<int, idx / H256_DMA_STRIDE>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            end_chunk();
        }
        char *scratch = between_tiles();

        // ---- epilogue of this tile.  Four groups (mi) of four 16-row x 32-column pieces; the residual of group g + 1 is
        // fetched (inline asm, counted waits) before group g's stores are issued, so no load ever waits for a store.  Rows
        // past the tile's own are neither read nor stored (resource extents).
        __amdgpu_buffer_rsrc_t ro;
        i32x4 rr;
        out_rsrc(row, rows, ro, rr);
        auto voff_of = [&](int mi, int ni, int hs) {
            return ((wr * 128 + mi * 32 + hs * 16 + t_row) * A.out_cs + wc * 64 + ni * 32 + t_col) * 2;
        };
        f32x4 rq[2][4];
        auto load_group = [&](int mi, f32x4 (&q)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = buf_load16_asm(rr, voff_of(mi, k >> 1, k & 1));
        };
        if constexpr (HAS_RES) load_group(0, rq[0]);
        static_for<0, 4>([&](auto mc) {
            constexpr int mi = decltype(mc)::value;
            if constexpr (HAS_RES) {
                if constexpr (mi < 3) load_group(mi + 1, rq[(mi + 1) & 1]);
                // younger than this group's loads: the next group's 4 loads and the previous group's 4 stores
                constexpr int younger = (mi < 3 ? 4 : 0) + (mi > 0 ? 4 : 0);
                wait_loaded4<younger>(rq[mi & 1][0], rq[mi & 1][1], rq[mi & 1][2], rq[mi & 1][3]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ni = k >> 1, hs = k & 1;
                finish_piece(scratch, acc[mi][ni], hs, rq[mi & 1][k], bias[ni][0], bias[ni][1], ro, voff_of(mi, ni, hs));
            }
        });
        if (!has_next) break;
        row = next_row;
        rows = next_rows;
        ra = ra_next;
        first_tile = false;
        if (rows <= 128) {                                // a short last tile: the light path
            tail_q = (rows + 31) >> 5;
            break;
        }
    }

    // ---- the short last tile of the range (<= 128 rows = Q <= 4 groups of 32): the SAME chunk stream -- its requests are
    // already in flight -- but another split of the work: wave w takes columns 32 w .. 32 w + 31 of the N tile and all Q row
    // groups, Q MFMAs per 16-deep step instead of 8 of which 8 - 2Q.. would multiply zero rows.  A range of 3.125 tiles
    // then costs 3 tiles and a fraction instead of 4.  Own accumulators, own (unpipelined) epilogue; nothing follows it.
    auto tail = [&](auto qc) __attribute__((always_inline)) {
        constexpr int Q = decltype(qc)::value;
        ra_next = r_none;
        rb_next = r_none;
        f32x16 acc_t[Q];
        const int ta_off = r * ROWB, tb_off = (wave * 32 + r) * ROWB;
        for (int kc = 0; kc < nk; ++kc) {
            const Chunk C = begin_chunk(kc);
            const char *rd_a = C.rd_a + ta_off, *rd_b = C.rd_b + tb_off;
            f32x4 fa[2][Q], fb[2];
            auto read_frags = [&](int ks, f32x4 (&a)[Q], f32x4 &b) __attribute__((always_inline)) {
                const int slot16 = ((ks * 2 + h) ^ swz) * 16;
#pragma unroll
                for (int m = 0; m < Q; ++m) a[m] = *reinterpret_cast<const f32x4 *>(rd_a + m * 32 * ROWB + slot16);
                b = *reinterpret_cast<const f32x4 *>(rd_b + slot16);
            };
            read_frags(0, fa[0], fb[0]);
            if (kc == 0) {
#pragma unroll
                for (int m = 0; m < Q; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc_t[m][e] = 0.f;
            }
            static_for<0, 8>([&](auto pc) { send_piece(C, pc); });
            static_for<0, 4>([&](auto kc4) {
                constexpr int ks = decltype(kc4)::value;
                if constexpr (ks < 3) read_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
                for (int m = 0; m < Q; ++m)
                    acc_t[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ks & 1][m]),
                                                                      __builtin_bit_cast(f16x8, fb[ks & 1]), acc_t[m], 0, 0, 0);
            });
            end_chunk();
        }
        char *scratch = between_tiles();
        __amdgpu_buffer_rsrc_t ro;
        i32x4 rr;
        out_rsrc(row, rows, ro, rr);
        f32x4 bt[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = nt * TN + wave * 32 + t_col + 4 * k;
            bt[k] = A.bias ? *reinterpret_cast<const f32x4 *>(A.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 0; m < Q; ++m)
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int voff = ((m * 32 + hs * 16 + t_row) * A.out_cs + wave * 32 + t_col) * 2;
                f32x4 res8 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (HAS_RES) {
                    res8 = buf_load16_asm(rr, voff);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(res8));
                }
                finish_piece(scratch, acc_t[m], hs, res8, bt[0], bt[1], ro, voff);
            }
    };
    switch (tail_q) {
        case 1: tail(std::integral_constant<int, 1>{}); break;
        case 2: tail(std::integral_constant<int, 2>{}); break;
        case 3: tail(std::integral_constant<int, 3>{}); break;
        case 4: tail(std::integral_constant<int, 4>{}); break;
        default: break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool HAS_RES>
int launch_h256(const H256Args &A, int grid, hipStream_t s) {
    auto kern = conv1x1_h256_kernel<HAS_RES>;
    static std::atomic<unsigned long long> ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), H256_LDS, ok, "conv1x1_h256")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), H256_LDS, s, A);
    ML_CHECK_LAUNCH("conv1x1_h256");
    return ML_OK;
}

}  // namespace

// 1 when this kernel handles the problem AND is expected to beat conv1x1_pipe_kernel<_Float16> on it
int ml_conv1x1_h256_eligible(const ml_conv2d_desc &d) {
    const bool shape_ok = d.math == ML_MATH_F16S && d.out_f16 == 1 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.dil == 1 &&
                          d.pad_t == 0 && d.pad_l == 0 && d.cpp_shift == 30 && d.group_cin_step == 0 && d.shuffle2x2 == 0 &&
                          d.out_bstride == 0 && d.Ho == d.H && d.Wo == d.W && !d.live && !d.gn_partials && d.act != ML_ACT_SIGMOID;
    if (!shape_ok) return 0;
    if (d.span % 64 != 0 || d.span < 256 || d.cout % 256 != 0 || d.n_pad != d.cout) return 0;
    const int G = d.cout / 256;
    if (G > 8 || (G & (G - 1))) return 0;
    if (d.in_cstride % 8 || d.in_coff % 8 || d.out_cstride % 8 || d.out_coff % 8 || !ml_aligned16(d.in) || !ml_aligned16(d.wgt) ||
        !ml_aligned16(d.out) || (d.residual && !ml_aligned16(d.residual)) || (d.bias && !ml_aligned16(d.bias)))
        return 0;
    if (d.residual && (d.res_cstride != d.out_cstride || d.res_coff != d.out_coff)) return 0;
    if ((long long)(TM + 8) * d.in_cstride * 2 >= (1ll << 31) || (long long)(TM + 8) * d.out_cstride * 2 >= (1ll << 31)) return 0;
    const long long M = (long long)d.B * d.H * d.W;
    const long long tiles = ((M + TM - 1) / TM) * G;
    return tiles >= ml_resident_blocks(1) ? 1 : 0;                // at least one full round of one block per CU
}

int ml_conv1x1_h256_try(const ml_conv2d_desc &d, hipStream_t s, int *took) {
    *took = 0;
    if (!ml_conv1x1_h256_eligible(d)) return ML_OK;
    const long long M = (long long)d.B * d.H * d.W;
    H256Args A;
    A.in = d.in; A.wgt = d.wgt; A.bias = d.bias; A.res = d.residual; A.out = d.out;
    A.M = M; A.K = d.span; A.N = d.cout;
    A.in_cs = d.in_cstride; A.in_coff = d.in_coff; A.out_cs = d.out_cstride; A.out_coff = d.out_coff;
    const int panels = (int)((M + TM - 1) / TM);
    A.G = d.cout / 256;
    A.gshift = 0;
    while ((1 << A.gshift) < A.G) ++A.gshift;
    const int resident = ml_resident_blocks(1);                  // one 8-wave block per CU (256 on MI355X; a multiple of 32)
    const long long units = (long long)panels * A.G;
    const int grid = units < resident ? (int)units : resident / A.G * A.G;
    A.xcd_map = (A.G > 1 && grid % (8 * A.G) == 0) ? 1 : 0;
    A.nb = grid >> A.gshift;                                      // row ranges: every block gets the same rows +- 32
    const long long groups32 = (M + 31) / 32;
    A.g_base = (int)(groups32 / A.nb);
    A.g_rem = (int)(groups32 % A.nb);
    A.clamp = d.act != ML_ACT_NONE;
    A.lo = 0.f;
    A.hi = d.act == ML_ACT_RELU6 ? 6.f : 3.402823466e38f;
    const int rc = d.residual ? launch_h256<true>(A, grid, s) : launch_h256<false>(A, grid, s);
    if (rc != ML_OK) return rc;
    *took = 1;
    return ML_OK;
}
