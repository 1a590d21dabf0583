// Persistent, software-pipelined 1x1 convolution (= GEMM  out[M,N] = act(in[M,K] * wgt[N,K]^T + bias + residual))
// on the gfx950 matrix cores, for the ResNeXt bottleneck 1x1 convs (reference engine/backbone/ResNext.py:199-231;
// thirdparty/classification_models/models/resnext.py:62-135 for ResNeXt-101) whose K is short (64 .. 512): there the
// generic implicit-GEMM kernel (conv_mfma.hip) spends as long in per-tile set-up, first-chunk latency and the store
// drain of its epilogue as in its K loop (44-65 % of the MFMA peak).
//
// Two storage types T:
//   float     tensors fp32 in HBM, v_mfma_f32_32x32x2_f32: exact fp32 products (configs 1-4, the default);
//   _Float16  tensors (activations, residual, output, weights) fp16 in HBM, v_mfma_f32_32x32x16_f16 with fp32
//             accumulation, bias / residual / clamp in fp32, ONE rounding to fp16 at the store: the "fp16 MFMA path"
//             of BASELINE config 5 with fp16 STORAGE -- the 1x1 convs are HBM-bound there, so bytes are what counts.
//   f32x3_t   ML_MATH_F32X3: fp32 tensors like `float`, products on the f16 matrix pipe from split operands (three
//             v_mfma_f32_32x32x16_f16 per 16-deep step, two accumulator sets; the weights arrive split, see x3_chunk).  The
//             short-K 1x1 convs of ResNeXt stages 1-2 are HBM-bound in that mode: 24 MFMAs of 32 cycles per chunk against the
//             64 x 64 cycles of `float`, so nothing needs to ride between MFMAs -- the epilogue of a tile runs in one piece at
//             the start of the next tile's first chunk, behind that chunk's staging requests (round 4).
// A K chunk is 128 bytes per row for all three (32 floats / 64 halves): staging, swizzle and fragment addressing are the
// same code; a chunk is 64 MFMAs of 64 cycles (f32) or 16 MFMAs of 32 cycles (f16).
//
// What is different from the generic kernel
//   * a block is PERSISTENT: it walks M panels (128 pixels) round-robin and, inside a panel, N tiles (128 channels),
//     so the panel's activations are fetched from HBM once and re-read from L2;
//   * the work of a tile that is not matrix math rides between the MFMAs of the NEIGHBOURING tiles, in the wave's
//     own instruction stream (on this chip a second wave cannot issue beside a wave that streams fp32 MFMAs, and
//     the wave's own non-MFMA instructions cost their issue slots 1:1 -- measured in round 1 -- so the only things
//     that can be removed are the exposed LATENCIES):
//       - the next chunk's LDS-direct loads cross tile boundaries (no first-chunk bubble);
//       - tile t's result stays in its accumulator registers while tile t+1 accumulates into a second set; during
//         tile t+1's first chunk it is transposed through a small WAVE-PRIVATE LDS scratch (16 rows x 32 columns at
//         a time: LDS operations of one wave execute in order, so no barrier), the residual -- fetched 4 pieces
//         ahead by hand-counted inline-asm loads -- and the bias are added, and it leaves as 16-byte-per-lane
//         stores (1 KB per instruction: a wave may have 63 memory operations in flight, so wide ones matter);
//   * all tile addressing is scalar: the buffer resources' base addresses are advanced per tile (64-bit SALU), lane
//     offsets never change, out-of-range rows (M tail) are dropped / zero-filled by the buffer range check -- which
//     also means no 2 GiB tensor limit on this path.
// Numerics (float): exact fp32 products, k-ordered fma chain per output like conv_mfma.hip (which starts the chain
// from the bias; here bias and residual are added after it: one rounding placed differently, ~1e-7 relative).
#include <type_traits>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                          // bytes of K per staged row and chunk
constexpr int BUFB = (BM + BN) * ROWB;             // bytes per staging buffer (32 KB)
constexpr int SCR_LD = 36;                         // floats per row of the transposition scratch (16-byte aligned rows)
constexpr int SCRB = 16 * SCR_LD * 4;              // bytes of scratch per wave
// LDS per block = NSTAGE staging buffers + 4 scratches + 4 B per bias channel of the block's N tiles.  NSTAGE = 2
// (78 KB, two blocks per CU) is what runs: for fp16 tensors a 4-deep ring with one block per CU was measured SLOWER
// (gpurun_out/r02i_half.log vs r02g_ab.log: 267 vs 232 us at 512 -> 1024 + residual) -- the fp16 chunk is 16 MFMAs of 32
// cycles and the 8 LDS-direct loads beside it cost more issue time than that, which a second resident block hides and a
// deeper ring does not.
constexpr int pipe_lds_bytes(int nstage) { return nstage * BUFB + 4 * SCRB; }
constexpr int PIPE_MAX_NBG = 8;                    // N tiles one block walks (its slice of the bias lives in LDS)

struct f32x3_t { float v; };                       // storage type tag of ML_MATH_F32X3 (fp32 tensors, split-operand products)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PipeArgs {
    const void *in, *wgt, *res;
    const float *bias;
    void *out;
    long long M;
    int K, N;                                // elements
    int in_cs, in_coff, out_cs, out_coff;    // elements; res shares out_cs / out_coff (checked by the launcher)
    int panels, NB, grid;
    int units, gshift, NBG;                  // work unit = (panel, group of NBG consecutive N tiles); 2^gshift groups per panel
    int xcd_map;                             // 1: the groups of one panel run at the same time on ONE XCD (see the kernel)
    float lo, hi;                            // activation as a clamp
};

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ void buf_store16(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, soff, 0);
#endif
}

// resource over [ptr + off, ptr + total): num_records saturates at 2^32 - 1 (a tile only ever reaches 128 rows in).
// (32-bit halves: 64-bit signed compares have no scalar form and would be evaluated on the VALU)
__device__ __forceinline__ unsigned records_left(long long off, long long total) {
    const unsigned long long left = (unsigned long long)(total - off);
    const unsigned hi = (unsigned)(left >> 32), lo = (unsigned)left;
    return (hi & 0x80000000u) ? 0u : (hi ? 0xffffffffu : lo);      // negative -> empty, >= 4 GiB -> saturate
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void *ptr, long long off, long long total) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)ptr + off), 0, (int)records_left(off, total), 0x00020000);
}
// the same descriptor as four plain dwords, for inline-asm operands (V# of a raw buffer: base, stride 0, num_records, flags)
__device__ __forceinline__ i32x4 rsrc_words(const void *ptr, long long off, long long total) {
    const unsigned long long base = (unsigned long long)((const char *)ptr + off);
    const i32x4 w = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)records_left(off, total), 0x00020000};
    return w;
}

// which tile a cursor points at; advanced with scalar adds only
struct Cursor {
    int u, nt;           // M panel and N tile; u >= panels: past the end (all its resources are empty)
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// Register-destination buffer load as inline asm: hipcc must not see it -- beside LDS-direct loads in flight it waits
// vmcnt(0) before every use of an ordinary load's result (cdna_hip_programming.md, "Pipelining across barriers"),
// which would drain the epilogue's stores again and again.  Its completion is waited for by hand (wait_loaded).
__device__ __forceinline__ f32x4 buf_load16_asm(i32x4 rsrc, int voff, int soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}
// wait until at most N vector-memory operations younger than the loads of a, b are outstanding; ties the values to the wait
template <int N>
__device__ __forceinline__ void wait_loaded(f32x4 &a, f32x4 &b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_loaded(f32x4 &a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
}

struct PipeState {
    // lane-constant offsets (bytes)
    int a_voff[4], b_voff[4];     // staging loads, relative to the panel / N-tile base
    int t_voff[2];                // transposed-layout element of this lane in N sub-tile ni, relative to the tile base
    int a_off, b_off, swz, h;     // fragment read offsets (bytes) / swizzle key
    int scr_w, scr_r;             // this lane's write / read byte offsets inside its wave's transposition scratch
    int wave_u;
    // scalars
    int rowoff[8];                // byte offset of the rows of half-sub-tile piece (mi, hs[, i]): see pipe_chunk
    float lo, hi;
};

// epilogue pieces per tile: (mi, ni, half) = 8 pieces of 16 rows x 32 columns; residual loads run LEAD pieces ahead
constexpr int NPC = 8;
constexpr int RES_LEAD = 4;

// One piece of the epilogue: sub-tile (qm, qn), rows 16 hs .. 16 hs + 15 of it.  r0 / r1: residual already loaded.
template <class T, int Q, bool HAS_RES, int NB4, bool CLAMP>
__device__ __forceinline__ void finish_piece(const f32x16 (&acc)[2][2], const PipeState &S, char *scratch,
                                             __amdgpu_buffer_rsrc_t ro, const f32x4 (&bias)[2][NB4], f32x4 r0, f32x4 r1) {
    constexpr bool F32 = !std::is_same<T, _Float16>::value;       // fp32 tensors (float, f32x3_t)
    constexpr int hs = Q & 1, qn = (Q >> 1) & 1, qm = Q >> 2;
    // C/D layout (col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) -> scratch[row][col]
    float *sw = reinterpret_cast<float *>(scratch + S.scr_w);
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) {
        const int e = hs * 8 + e8;
        sw[((e & 3) + 8 * ((e >> 2) & 1)) * SCR_LD] = acc[qm][qn][e];
    }
    // ... and back as 4 (f32) / 8 (f16) consecutive channels of one pixel per lane
    const f32x4 *sr = reinterpret_cast<const f32x4 *>(scratch + S.scr_r);
    f32x4 v0 = sr[0];
    f32x4 v1 = F32 ? sr[8 * SCR_LD / 4] : sr[1];       // f32: same columns 8 rows down; f16: the next 4 columns
    if constexpr (HAS_RES) {
        if constexpr (F32) {
            v0 += r0;
            v1 += r1;
        } else {
            const f16x8 rh = __builtin_bit_cast(f16x8, r0);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v0[c] += (float)rh[c]; v1[c] += (float)rh[4 + c]; }
        }
    }
    v0 += bias[qn][0];
    v1 += bias[qn][NB4 - 1];
    if constexpr (CLAMP) {               // ML_ACT_NONE stores the sum as it is (NaN / Inf stay what they are)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v0[c] = __builtin_amdgcn_fmed3f(v0[c], S.lo, S.hi);
            v1[c] = __builtin_amdgcn_fmed3f(v1[c], S.lo, S.hi);
        }
    }
    if constexpr (F32) {
        buf_store16(v0, ro, S.t_voff[qn], S.rowoff[(qm * 2 + hs) * 2 + 0]);
        buf_store16(v1, ro, S.t_voff[qn], S.rowoff[(qm * 2 + hs) * 2 + 1]);
    } else {
        const f16x8 o = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                         (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
        buf_store16(__builtin_bit_cast(f32x4, o), ro, S.t_voff[qn], S.rowoff[(qm * 2 + hs) * 2]);
    }
}

// One K chunk (128 bytes deep): the MFMAs of the wave's 64x64 sub-tile with the non-matrix work pinned between them.
//   always     : the 8 LDS-direct loads of the NEXT chunk (may belong to the next tile), one after each of MFMA 0..7
//   FIRST chunk: the PREVIOUS tile's accumulators (`oth`) in NPC + LEAD pieces, spread over the remaining MFMAs:
//                piece p: [p >= LEAD] finish piece q = p - LEAD: 8 ds_write_b32 (C/D layout -> scratch rows),
//                         16-byte ds_reads (a lane then owns consecutive channels of one pixel), + residual + bias,
//                         clamp, [cvt,] 16-byte store(s);
//                         [p < NPC, residual] issue the residual load(s) of piece p (consumed LEAD pieces later).
// Vector-memory issue order of a FIRST chunk, on which the hand-counted waits rely (sched_barrier pins it):
//   8 staging loads, then per piece: stores(q) before loads(p).
template <class T, bool FIRST, bool HAS_RES, int NB4, bool CLAMP>   // (the staging depth only shows in which buffer `wr` is)
__device__ __forceinline__ void pipe_chunk(f32x16 (&cur)[2][2], f32x16 (&oth)[2][2], const PipeState &S, const char *rd,
                                           char *wr, char *scratch, __amdgpu_buffer_rsrc_t ra_nx,
                                           __amdgpu_buffer_rsrc_t rb_nx, int soff_nx, __amdgpu_buffer_rsrc_t ro_prev,
                                           const f32x4 (&bias_prev)[2][NB4], i32x4 rr_prev) {
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int NSLOT = F32 ? 64 : 16;                 // MFMAs per chunk and wave
    constexpr int UNITS = F32 ? 2 : 1;                   // 16-byte memory operations per lane and piece
    constexpr int LEAD = HAS_RES ? RES_LEAD : 0;
    constexpr int NREG = FIRST ? NPC + LEAD : 0;
    f32x4 ring[RES_LEAD][2];
    // fragments are double buffered: the ds_reads of k-step ks+1 are issued in the middle of k-step ks's MFMAs
    f32x4 fa[2][2], fb[2][2];
    auto read_frags = [&](int ks, f32x4 (&a)[2], f32x4 (&b)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
            a[m] = *reinterpret_cast<const f32x4 *>(rd + S.a_off + m * 32 * ROWB + (((ks * 2 + S.h) ^ S.swz) * 16));
#pragma unroll
        for (int n = 0; n < 2; ++n)
            b[n] = *reinterpret_cast<const f32x4 *>(rd + S.b_off + n * 32 * ROWB + (((ks * 2 + S.h) ^ S.swz) * 16));
    };
    read_frags(0, fa[0], fb[0]);
    static_for<0, NSLOT>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;          // MFMA number: ((ks*J + j)*2 + mi)*2 + ni, J = 4 (f32) / 1 (f16)
        constexpr int ni = idx & 1, mi = (idx >> 1) & 1;
        constexpr int j = F32 ? (idx >> 2) & 3 : 0, ks = F32 ? idx >> 4 : idx >> 2;
        constexpr int in_ks = F32 ? idx & 15 : idx & 3;  // position inside the k-step
        f32x4 (&a)[2] = fa[ks & 1];
        f32x4 (&b)[2] = fb[ks & 1];
        if constexpr (in_ks == (F32 ? 6 : 1) && ks < 3) read_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        constexpr bool start = FIRST && ks == 0 && j == 0;   // a tile's chain starts from C = 0 (inline constant)
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (F32) {
            if constexpr (start) cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], zero, 0, 0, 0);
            else cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], cur[mi][ni], 0, 0, 0);
        } else {
            const f16x8 ha = __builtin_bit_cast(f16x8, a[mi]), hb = __builtin_bit_cast(f16x8, b[ni]);
            if constexpr (start) cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, zero, 0, 0, 0);
            else cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, cur[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (idx < 4) {
            lds_dma16(ra_nx, wr + (32 * idx + 8 * S.wave_u) * ROWB, S.a_voff[idx], soff_nx);
        } else if constexpr (idx < 8) {
            lds_dma16(rb_nx, wr + (BM + 32 * (idx - 4) + 8 * S.wave_u) * ROWB, S.b_voff[idx - 4], soff_nx);
        } else if constexpr (FIRST) {
            // pieces [pb, pe) ride after this MFMA: spread evenly over MFMAs 8 .. NSLOT-1
            constexpr int pb = ((idx - 8) * NREG) / (NSLOT - 8), pe = ((idx - 7) * NREG) / (NSLOT - 8);
            static_for<pb, pe>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if constexpr (p >= LEAD) {
                    constexpr int q = p - LEAD;
                    if constexpr (HAS_RES) {
                        // vector-memory operations younger than the loads of piece q at this point: the loads of
                        // pieces q+1 .. min(p, NPC) - 1 and the stores of pieces max(0, q - LEAD + 1) .. q - 1
                        constexpr int loads_after = ((p < NPC ? p : NPC) - q - 1) * UNITS;
                        constexpr int stores_after = (q - (q >= LEAD ? q - LEAD + 1 : 0)) * UNITS;
                        constexpr int younger = loads_after + stores_after;
                        static_assert(younger >= 0 && younger <= 63, "vmcnt immediate out of range");
                        if constexpr (F32) wait_loaded<younger>(ring[q % RES_LEAD][0], ring[q % RES_LEAD][1]);
                        else wait_loaded<younger>(ring[q % RES_LEAD][0]);
                    }
                    finish_piece<T, q, HAS_RES, NB4, CLAMP>(oth, S, scratch, ro_prev, bias_prev, ring[q % RES_LEAD][0],
                                                     ring[q % RES_LEAD][F32 ? 1 : 0]);
                }
                if constexpr (HAS_RES && p < NPC) {
                    constexpr int hs = p & 1, pn = (p >> 1) & 1, pm = p >> 2;
                    ring[p % RES_LEAD][0] = buf_load16_asm(rr_prev, S.t_voff[pn], S.rowoff[(pm * 2 + hs) * 2 + 0]);
                    if constexpr (F32)
                        ring[p % RES_LEAD][1] = buf_load16_asm(rr_prev, S.t_voff[pn], S.rowoff[(pm * 2 + hs) * 2 + 1]);
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- ML_MATH_F32X3 ------------------------------------------------------------------------------------------------------
// Epilogue of one tile in one piece: the cross terms are folded in (units of 2^-11, as conv_mfma.hip does), ALL residual
// pieces are requested at once (their 64 registers are the cross-term accumulators', dead after the fold), then the eight
// pieces are transposed, finished and stored in order.  Vector-memory order: 16 residual loads, then 2 stores per piece --
// younger than piece q's loads are the loads of pieces q+1 .. 7 and the stores of pieces 0 .. q-1: always 14.
template <bool HAS_RES, bool CLAMP>
__device__ __forceinline__ void x3_finish(f32x16 (&cur)[2][2], const f32x16 (&crx)[2][2], const PipeState &S, char *scratch,
                                          __amdgpu_buffer_rsrc_t ro, const float *bias_lds, i32x4 rr) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[mi][ni][e] = fmaf(crx[mi][ni][e], 0x1p-11f, cur[mi][ni][e]);
    __builtin_amdgcn_sched_barrier(0);                     // (the loads below take the registers the cross terms leave)
    f32x4 ring[NPC][2];
    static_for<0, NPC>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int hs = p & 1, pn = (p >> 1) & 1, pm = p >> 2;
        if constexpr (HAS_RES) {
            ring[p][0] = buf_load16_asm(rr, S.t_voff[pn], S.rowoff[(pm * 2 + hs) * 2 + 0]);
            ring[p][1] = buf_load16_asm(rr, S.t_voff[pn], S.rowoff[(pm * 2 + hs) * 2 + 1]);
        } else {
            ring[p][0] = ring[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    // the tile's bias comes from LDS here (bias_lds: this lane's four channels of N sub-tile 0; sub-tile 1 is 32 floats on):
    // registers are what this form is short of at two blocks per CU
    f32x4 bias[2][1];
    bias[0][0] = *reinterpret_cast<const f32x4 *>(bias_lds);
    bias[1][0] = *reinterpret_cast<const f32x4 *>(bias_lds + 32);
    static_for<0, NPC>([&](auto pc) {
        constexpr int q = decltype(pc)::value;
        if constexpr (HAS_RES) wait_loaded<2 * (NPC - 1)>(ring[q][0], ring[q][1]);
        finish_piece<float, q, HAS_RES, 1, CLAMP>(cur, S, scratch, ro, bias, ring[q][0], ring[q][1]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One K chunk (32 deep) of the split-operand form.  Staged rows: activations 32 floats; weights 32 halves hi(w) then 32 halves
// 2^11 (w - hi(w)) (masklab_hip.h, ML_MATH_F32X3) -- 128 bytes either way, so staging and swizzle are the `float` code.  Lane
// (r, h) feeds k = 16 ks + 8 h .. + 7 of its row to v_mfma_f32_32x32x16_f16: eight floats (k-groups 4 ks + 2 h and + 1) split
// here into hi / lo halves, weight k-groups 2 ks + h (hi) and 4 + 2 ks + h (lo).  Per 16-deep step and 32 x 32 sub-tile:
// hi hi -> `cur`, hi lo and lo hi -> `crx` (same three products and the same k order as the generic kernel's X3 path).
//   FIRST: the PREVIOUS tile, still in cur / crx, is finished and stored first (x3_finish) -- behind this chunk's staging
//   requests, so those fly under it -- and the chains of the new tile start from zero.
// Vector-memory order of a FIRST chunk, on which the caller's counted wait relies: 8 staging loads, 16 residual loads
// (waited for inside), 16 stores.
template <bool FIRST, bool HAS_RES, bool CLAMP>
__device__ __forceinline__ void x3_chunk(f32x16 (&cur)[2][2], f32x16 (&crx)[2][2], const PipeState &S, const char *rd, char *wr,
                                         char *scratch, __amdgpu_buffer_rsrc_t ra_nx, __amdgpu_buffer_rsrc_t rb_nx, int soff_nx,
                                         __amdgpu_buffer_rsrc_t ro_prev, const float *bias_prev, i32x4 rr_prev) {
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(ra_nx, wr + (32 * i + 8 * S.wave_u) * ROWB, S.a_voff[i], soff_nx);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rb_nx, wr + (BM + 32 * i + 8 * S.wave_u) * ROWB, S.b_voff[i], soff_nx);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST) x3_finish<HAS_RES, CLAMP>(cur, crx, S, scratch, ro_prev, bias_prev, rr_prev);
    const float neg_scale = -2048.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const char *row = rd + S.a_off + mi * 32 * ROWB;
            split_hi_lo(*reinterpret_cast<const f32x4 *>(row + (((4 * ks + 2 * S.h) ^ S.swz) * 16)),
                        *reinterpret_cast<const f32x4 *>(row + (((4 * ks + 2 * S.h + 1) ^ S.swz) * 16)), neg_scale, ah[mi], al[mi]);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const char *row = rd + S.b_off + ni * 32 * ROWB;
            bh[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(row + (((2 * ks + S.h) ^ S.swz) * 16)));
            bl[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(row + (((4 + 2 * ks + S.h) ^ S.swz) * 16)));
        }
        if (FIRST && ks == 0) {                            // (compile time after unrolling) the chains start from C = 0
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], zero, 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) crx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], zero, 0, 0, 0);
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], cur[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) crx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], crx[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                crx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], crx[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // (keeps the second step's fragments out of the first step's registers)
    }
}

template <class T, bool HAS_RES, int NSTAGE, bool CLAMP>
__global__ void __launch_bounds__(256, NSTAGE == 2 ? 2 : 1)
conv1x1_pipe_kernel(const PipeArgs A) {
    constexpr bool X3 = std::is_same<T, f32x3_t>::value;
    constexpr bool F32 = !std::is_same<T, _Float16>::value;   // fp32 tensors
    constexpr int ES = sizeof(T);
    constexpr int KC = ROWB / ES;                              // K elements per chunk
    constexpr int NB4 = F32 ? 1 : 2;                           // float4 of bias per lane and N sub-tile
    extern __shared__ __align__(16) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int ld_row = tid >> 3;
    const int ld_g = (tid & 7) ^ ((ld_row >> 1) & 7);          // fetch the 16-byte k-group whose (swizzled) slot this lane fills

    PipeState S;
    S.h = h;
    S.swz = (r >> 1) & 7;
    S.a_off = (wm * 64 + r) * ROWB;
    S.b_off = BM * ROWB + (wn * 64 + r) * ROWB;
    S.wave_u = __builtin_amdgcn_readfirstlane(wave);
    S.lo = A.lo;
    S.hi = A.hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        S.a_voff[i] = ((ld_row + 32 * i) * A.in_cs + A.in_coff) * ES + ld_g * 16;
        S.b_voff[i] = (ld_row + 32 * i) * A.K * ES + ld_g * 16;
    }
    // transposed ownership inside a 16-row x 32-column piece: f32 -- lane -> row lane >> 3 (and + 8), columns
    // 4 (lane & 7) .. + 3; f16 -- lane -> row lane >> 2, columns 8 (lane & 3) .. + 7
    const int t_row = F32 ? (lane >> 3) : (lane >> 2);
    const int t_col = F32 ? (lane & 7) * 4 : (lane & 3) * 8;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
        S.t_voff[ni] = ((wm * 64 + t_row) * A.out_cs + A.out_coff + wn * 64 + ni * 32 + t_col) * ES;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs)
#pragma unroll
            for (int i = 0; i < 2; ++i) S.rowoff[(mi * 2 + hs) * 2 + i] = (mi * 32 + hs * 16 + 8 * i) * A.out_cs * ES;
    S.scr_w = (4 * h * SCR_LD + r) * 4;
    S.scr_r = (t_row * SCR_LD + t_col) * 4;

    const int nk = A.K / KC;
    const long long in_total = A.M * (long long)A.in_cs * ES;
    const long long out_total = A.M * (long long)A.out_cs * ES;
    const long long w_total = (long long)A.N * A.K * ES;
    // Which (panel, N-tile group) this block works on.  A block keeps ONE group for its whole life and walks panels
    // panel0, panel0 + step, ...  Two maps, the same work either way:
    //   plain   : block b -> group b & gmask, panel0 = b >> gshift (adjacent blocks = adjacent groups of one panel, which
    //             the dispatcher deals to DIFFERENT XCDs);
    //   xcd_map : blocks with equal b & 7 share an XCD and its L2 -- slot j = b >> 3 of XCD x = b & 7 takes group j & gmask
    //             of panel x * ppx + (j >> gshift): the 2^gshift groups of a panel run side by side on ONE XCD, so the
    //             panel's activations come from beyond L2 once and from L2 for the other groups.  (Measured before: one
    //             block walking a panel's N tiles one after the other re-fetched the panel for EVERY tile -- 64 blocks per
    //             XCD stream 12+ MB through its 4 MB L2 between two tiles; gpurun_out/shape_pmc)
    const int gmask = (1 << A.gshift) - 1;
    const int step = A.grid >> A.gshift;                       // panels in flight per round
    const int bid = (int)blockIdx.x;
    const int slot = bid >> 3, ppx = step >> 3;
    const int group = A.xcd_map ? (slot & gmask) : (bid & gmask);
    const int panel0 = A.xcd_map ? (bid & 7) * ppx + (slot >> A.gshift) : (bid >> A.gshift);
    auto panel_of = [&](Cursor c) { return c.u; };
    auto res_a = [&](Cursor c) { return rsrc_at(A.in, (long long)panel_of(c) * BM * A.in_cs * ES, c.u < A.panels ? in_total : 0); };
    auto res_b = [&](Cursor c) { return rsrc_at(A.wgt, (long long)c.nt * BN * A.K * ES, c.u < A.panels ? w_total : 0); };
    auto tile_off = [&](Cursor c) { return ((long long)panel_of(c) * BM * A.out_cs + (long long)c.nt * BN) * ES; };
    // rows past M must fall outside: the extent is counted from the tile's first element
    auto res_o = [&](Cursor c) { return rsrc_at(A.out, tile_off(c), c.u < A.panels ? out_total : 0); };
    auto res_words = [&](Cursor c) { return rsrc_words(A.res, tile_off(c), (HAS_RES && c.u < A.panels) ? out_total : 0); };
    const int nt0 = group * A.NBG;
    auto advance = [&](Cursor c) {
        Cursor n = c;
        if (n.nt + 1 < nt0 + A.NBG) { ++n.nt; } else { n.u += step; n.nt = nt0; }
        return n;
    };
    // LDS: [2 staging buffers][4 wave-private transposition scratches][bias of this block's N tiles].  A block only
    // ever works on ONE group of NBG N tiles (see the maps above: the grid size is a multiple of the group count, the
    // of the group count).  The bias is read with ds_read at a tile switch: a global load there would be waited for
    // with vmcnt(0) while LDS-direct loads are in flight.
    char *scratch = lds + NSTAGE * BUFB + S.wave_u * SCRB;
    float *lds_bias = reinterpret_cast<float *>(lds + NSTAGE * BUFB + 4 * SCRB);
    for (int i = tid; i < A.NBG * BN; i += 256) lds_bias[i] = A.bias ? A.bias[nt0 * BN + i] : 0.f;
    __syncthreads();
    // X3: the address of this lane's bias slice of tile c (read in x3_finish); before the first tile: any valid slice
    auto bias_at = [&](Cursor c) { return lds_bias + (c.u < A.panels ? c.nt - nt0 : 0) * BN + wn * 64 + t_col; };
    auto load_bias = [&](Cursor c, f32x4 (&bv)[2][NB4]) {
        if constexpr (X3) return;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int k = 0; k < NB4; ++k)
                bv[ni][k] = *reinterpret_cast<const f32x4 *>(lds_bias + (c.nt - nt0) * BN + wn * 64 + ni * 32 + t_col + 4 * k);
    };

    Cursor cur = {panel0, nt0};
    if (cur.u >= A.panels) return;
    Cursor nx = cur;                      // tile of the NEXT chunk to stage
    int kc_nx = 0;

    f32x16 accA[2][2], accB[2][2];
    f32x4 bias_cur[2][NB4], bias_prev[2][NB4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int k = 0; k < NB4; ++k) bias_prev[ni][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_bias(cur, bias_cur);
    // ---- prologue: the first NSTAGE - 1 chunks of the block's stream (they may already cross into the next tile)
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) {
        const __amdgpu_buffer_rsrc_t ra = res_a(nx), rb = res_b(nx);
        const int soff = kc_nx * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ra, lds + st * BUFB + (32 * i + 8 * S.wave_u) * ROWB, S.a_voff[i], soff);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rb, lds + st * BUFB + (BM + 32 * i + 8 * S.wave_u) * ROWB, S.b_voff[i], soff);
        ++kc_nx;
        if (kc_nx == nk) { kc_nx = 0; nx = advance(nx); }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NSTAGE - 2)) : "memory");      // chunk 0 has landed
    __builtin_amdgcn_s_barrier();

    Cursor prev = {A.panels, 0};          // nothing to store yet: empty resource
    int buf = 0;
    // One tile: `acc` accumulates it, `oth` holds the previous tile's result (finished and stored during the first
    // chunk).  Returns false after the block's last tile.
    // X3: acc = the tile's hi x hi sums, crx = its cross terms; the first chunk (peeled: it stores the previous tile) and the rest
    auto run_tile_x3 = [&](f32x16 (&acc)[2][2], f32x16 (&crx)[2][2]) -> bool {
        const Cursor next = advance(cur);
        const bool has_next = next.u < A.panels;
        auto chunk = [&](auto firstc) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value;
            const __amdgpu_buffer_rsrc_t ra_nx = res_a(nx), rb_nx = res_b(nx);
            const int soff_nx = kc_nx * ROWB;
            const char *rd = lds + buf * BUFB;
            char *wr = lds + (buf == 0 ? NSTAGE - 1 : buf - 1) * BUFB;
            if constexpr (first)
                x3_chunk<true, HAS_RES, CLAMP>(acc, crx, S, rd, wr, scratch, ra_nx, rb_nx, soff_nx, res_o(prev), bias_at(prev), res_words(prev));
            else
                x3_chunk<false, HAS_RES, CLAMP>(acc, crx, S, rd, wr, scratch, ra_nx, rb_nx, soff_nx, ra_nx, nullptr, i32x4{0, 0, 0, 0});
            ++kc_nx;
            if (kc_nx == nk) { kc_nx = 0; nx = advance(nx); }
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
            // the next chunk has landed: younger than its requests are only the 16 stores of a first chunk (see run_tile)
            if constexpr (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NSTAGE - 2) + 16) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NSTAGE - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
        };
        chunk(std::true_type{});
        for (int kc = 1; kc < nk; ++kc) chunk(std::false_type{});
        prev = cur;
        cur = next;
        return has_next;
    };
    auto run_tile = [&](f32x16 (&acc)[2][2], f32x16 (&oth)[2][2]) -> bool {
        const Cursor next = advance(cur);
        const bool has_next = next.u < A.panels;
        const __amdgpu_buffer_rsrc_t ro_prev = res_o(prev);
        const i32x4 rr_prev = res_words(prev);
        for (int kc = 0; kc < nk; ++kc) {
            const __amdgpu_buffer_rsrc_t ra_nx = res_a(nx), rb_nx = res_b(nx);
            const int soff_nx = kc_nx * ROWB;
            const char *rd = lds + buf * BUFB;
            char *wr = lds + (buf == 0 ? NSTAGE - 1 : buf - 1) * BUFB;     // the buffer the previous chunk was read from
            const bool first = kc == 0;
            // (wave-uniform branch: two instantiations of the chunk per accumulator role)
            if (first) pipe_chunk<T, true, HAS_RES, NB4, CLAMP>(acc, oth, S, rd, wr, scratch, ra_nx, rb_nx, soff_nx, ro_prev, bias_prev, rr_prev);
            else pipe_chunk<T, false, HAS_RES, NB4, CLAMP>(acc, oth, S, rd, wr, scratch, ra_nx, rb_nx, soff_nx, ro_prev, bias_prev, rr_prev);
            // advance the staging cursor
            ++kc_nx;
            if (kc_nx == nk) { kc_nx = 0; nx = advance(nx); }
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
            // The NEXT chunk's staging loads must have landed.  Younger than them are the loads of the NSTAGE - 2
            // chunks after it and the piece stores (16 / 8) of a first chunk among the last NSTAGE - 1 chunks (its
            // residual loads were waited for inside it): vector-memory operations complete in order, so allowing
            // exactly that many to stay outstanding waits for nothing else.  (Short tiles with more than one first
            // chunk in that window only make the wait stricter.)
            if (kc <= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NSTAGE - 2) + (F32 ? 16 : 8)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NSTAGE - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        prev = cur;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int k = 0; k < NB4; ++k) bias_prev[ni][k] = bias_cur[ni][k];
        cur = next;
        if (has_next) load_bias(cur, bias_cur);
        return has_next;
    };
    // ---- drain: the last tile's registers have not been stored yet (same piece code, nothing to overlap with)
    auto drain = [&](f32x16 (&acc)[2][2]) {
        const __amdgpu_buffer_rsrc_t ro = res_o(prev);
        const i32x4 rr = res_words(prev);
        static_for<0, NPC>([&](auto pc) {
            constexpr int q = decltype(pc)::value;
            constexpr int hs = q & 1, qn = (q >> 1) & 1, qm = q >> 2;
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
            if constexpr (HAS_RES) {
                r0 = buf_load16_asm(rr, S.t_voff[qn], S.rowoff[(qm * 2 + hs) * 2 + 0]);
                if constexpr (F32) {
                    r1 = buf_load16_asm(rr, S.t_voff[qn], S.rowoff[(qm * 2 + hs) * 2 + 1]);
                    wait_loaded<0>(r0, r1);
                } else {
                    wait_loaded<0>(r0);
                }
            }
            finish_piece<T, q, HAS_RES, NB4, CLAMP>(acc, S, scratch, ro, bias_prev, r0, r1);
        });
    };
    if constexpr (X3) {
        // accA / accB keep their roles (sums / cross terms); the last tile is finished by the same one-piece epilogue
        while (run_tile_x3(accA, accB)) {}
        x3_finish<HAS_RES, CLAMP>(accA, accB, S, scratch, res_o(prev), bias_at(prev), res_words(prev));
    } else {
        // the tile loop, unrolled by two so that the accumulator sets swap roles without register copies
        for (;;) {
            if (!run_tile(accA, accB)) { drain(accA); break; }
            if (!run_tile(accB, accA)) { drain(accB); break; }
        }
    }
}

template <class T, bool HAS_RES, int NSTAGE, bool CLAMP>
int launch_pipe(const PipeArgs &A, hipStream_t s) {
    auto kern = conv1x1_pipe_kernel<T, HAS_RES, NSTAGE, CLAMP>;
    static std::atomic<unsigned long long> ok{0};          // per kernel instantiation, one bit per device
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), pipe_lds_bytes(NSTAGE) + 4 * PIPE_MAX_NBG * BN, ok,
                                       "conv1x1_pipe"))
        return rc;
    hipLaunchKernelGGL(kern, dim3(A.grid), dim3(256), pipe_lds_bytes(NSTAGE) + 4 * A.NBG * BN, s, A);
    ML_CHECK_LAUNCH("conv1x1_pipe");
    return ML_OK;
}

}  // namespace

// 1 when the problem is one this kernel handles (the caller otherwise uses the generic implicit-GEMM kernel)
int ml_conv1x1_pipe_eligible(const ml_conv2d_desc &d) {
    const bool f16s = d.math == ML_MATH_F16S, x3 = d.math == ML_MATH_F32X3;
    const int kc = f16s ? 64 : 32;
    const bool shape_ok = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.dil == 1 && d.pad_t == 0 && d.pad_l == 0 &&
                          d.cpp_shift == 30 && d.group_cin_step == 0 && d.shuffle2x2 == 0 && d.out_bstride == 0 &&
                          (d.math == ML_MATH_F32 || f16s || x3) && d.Ho == d.H && d.Wo == d.W;
    if (!shape_ok || d.live || d.gn_partials) return 0;
    if (f16s && !d.out_f16) return 0;                     // this kernel stores the tensor type it reads (fp32 predictions: generic kernel)
    if (!f16s && d.out_f16) return 0;
    if (d.span % kc != 0 || d.cout % 128 != 0 || d.n_pad != d.cout) return 0;
    if (!f16s && d.span != d.span_pad) return 0;
    if (d.act == ML_ACT_SIGMOID) return 0;
    const int q = f16s ? 8 : 4;                           // elements per 16 bytes
    if (d.in_cstride % q || d.in_coff % q || d.out_cstride % q || d.out_coff % q || !ml_aligned16(d.in) ||
        !ml_aligned16(d.wgt) || !ml_aligned16(d.out) || (d.residual && !ml_aligned16(d.residual)))
        return 0;
    if (d.residual && (d.res_cstride != d.out_cstride || d.res_coff != d.out_coff)) return 0;
    // lane offsets are 32-bit and cover one 128-row tile only
    if ((long long)BM * d.in_cstride * 4 >= (1ll << 31) || (long long)(BM + 32) * d.out_cstride * 4 >= (1ll << 31)) return 0;
    return 1;
}

// -> ML_OK and *eligible = 1 when the problem was launched on the pipelined kernel; *eligible = 0 when it is not a
// problem this kernel handles.
int ml_conv1x1_pipe_try(const ml_conv2d_desc &d, hipStream_t s, int *eligible) {
    *eligible = 0;
    if (!ml_conv1x1_pipe_eligible(d)) return ML_OK;
    const long long M = (long long)d.B * d.H * d.W;
    PipeArgs A;
    A.in = d.in; A.wgt = d.wgt; A.bias = d.bias; A.res = d.residual; A.out = d.out;
    A.M = M; A.K = d.span; A.N = d.cout;
    A.in_cs = d.in_cstride; A.in_coff = d.in_coff; A.out_cs = d.out_cstride; A.out_coff = d.out_coff;
    A.panels = (int)((M + BM - 1) / BM);
    A.NB = d.cout / BN;
    // enough work units to fill every resident block (2 per CU), and at most PIPE_MAX_NBG N tiles per block: split a
    // panel's N tiles into 2^gshift groups
    const int resident = ml_resident_blocks(2);          // two 4-wave blocks per CU (512 on MI355X)
    // Split a panel's N tiles into 2^gshift groups (a unit = one group of one panel).  Blocks are persistent and walk
    // panels in rounds of (grid >> gshift), so the launch lasts ceil(units / grid) units: take the splits whose last
    // round is fullest (800 panels: 1 group -> 2 rounds for 1.56 rounds of work, 4 groups -> 7 for 6.25) and among those
    // within 2 % the FINEST one the XCD map can place (every N tile of a panel then runs at the same time on one XCD and
    // the panel is fetched from beyond L2 once; a block that walks several N tiles re-fetches it for each of them).
    double eff_of[8] = {0, 0, 0, 0, 0, 0, 0, 0}, best_eff = -1.0;
    int n_g = 0;
    for (int g = 0; g < 8 && (A.NB >> g) >= 1 && (A.NB % (1 << g)) == 0; ++g) {
        n_g = g + 1;
        if ((A.NB >> g) > PIPE_MAX_NBG || (1 << g) > resident) continue;
        const long long units = (long long)A.panels << g;
        const long long grid = units < resident ? units : resident / (1 << g) * (1 << g);
        const long long rounds = (units + grid - 1) / grid;
        eff_of[g] = (double)units / (double)(rounds * resident);
        if (eff_of[g] > best_eff) best_eff = eff_of[g];
    }
    if (best_eff <= 0.0) return ML_OK;                    // (an odd tile count above the limit: the generic kernel)
    int best = -1, best_map = 0;
    for (int g = n_g - 1; g > 0 && best < 0; --g) {       // finest first, where the XCD map applies
        if (eff_of[g] <= 0.0 || eff_of[g] < best_eff - 0.02) continue;
        const long long units = (long long)A.panels << g;
        const long long grid = units < resident ? units : resident / (1 << g) * (1 << g);
        if (grid % (8ll << g) == 0) { best = g; best_map = 1; }
    }
    for (int g = 0; g < n_g && best < 0; ++g)             // else the coarsest (plain map)
        if (eff_of[g] > 0.0 && eff_of[g] >= best_eff - 0.02) best = g;
    A.gshift = best;
    A.xcd_map = best_map;
    A.NBG = A.NB >> A.gshift;
    A.units = A.panels << A.gshift;
    // a block keeps ONE group's bias in LDS and advances by the grid size: the grid must be a multiple of the group count
    A.grid = A.units < resident ? A.units : resident / (1 << A.gshift) * (1 << A.gshift);
    const bool clamp = d.act != ML_ACT_NONE;
    A.lo = 0.f;
    A.hi = d.act == ML_ACT_RELU6 ? 6.f : 3.402823466e38f;
    int rc;
    if (d.math == ML_MATH_F16S) {
        if (clamp) rc = d.residual ? launch_pipe<_Float16, true, 2, true>(A, s) : launch_pipe<_Float16, false, 2, true>(A, s);
        else rc = d.residual ? launch_pipe<_Float16, true, 2, false>(A, s) : launch_pipe<_Float16, false, 2, false>(A, s);
    } else if (d.math == ML_MATH_F32X3) {        // d.wgt = the split packing (same bytes per row as fp32)
        if (clamp) rc = d.residual ? launch_pipe<f32x3_t, true, 2, true>(A, s) : launch_pipe<f32x3_t, false, 2, true>(A, s);
        else rc = d.residual ? launch_pipe<f32x3_t, true, 2, false>(A, s) : launch_pipe<f32x3_t, false, 2, false>(A, s);
    } else {
        if (clamp) rc = d.residual ? launch_pipe<float, true, 2, true>(A, s) : launch_pipe<float, false, 2, true>(A, s);
        else rc = d.residual ? launch_pipe<float, true, 2, false>(A, s) : launch_pipe<float, false, 2, false>(A, s);
    }
    if (rc != ML_OK) return rc;
    *eligible = 1;
    return ML_OK;
}
