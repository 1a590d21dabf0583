// Persistent, software-pipelined 1x1 convolution (= GEMM  out[M,N] = act(in[M,K] * wgt[N,K]^T + bias + residual))
// on the gfx950 f32 matrix cores, for the ResNeXt bottleneck 1x1 convs (reference engine/backbone/ResNext.py:199-231)
// whose K is short (64 .. 512): there the generic implicit-GEMM kernel (conv_mfma.hip) spends as long in per-tile
// set-up, first-chunk latency and the store drain of its epilogue as in its K loop (44-65 % of the MFMA peak).
//
// What is different here
//   * a block is PERSISTENT: it walks M panels (128 pixels) round-robin and, inside a panel, all N tiles (128
//     channels), so the panel's activations are fetched from HBM once and re-read from L2;
//   * the work of a tile that is not matrix math rides between the MFMAs of the NEIGHBOURING tiles, in the wave's
//     own instruction stream (on this chip a second wave cannot issue beside a wave that streams fp32 MFMAs, and
//     the wave's own non-MFMA instructions cost their issue slots 1:1 -- measured in round 1 -- so the only things
//     that can be removed are the exposed LATENCIES):
//       - the next chunk's LDS-direct loads cross tile boundaries (no first-chunk bubble);
//       - tile t's result stays in its accumulator registers while tile t+1 accumulates into a second set; its
//         64 registers are biased, clamped and stored straight from the C/D layout (one register = 2 rows x 128 B)
//         between the MFMAs of tile t+1's first chunk -- no LDS transpose, no barrier, no store drain;
//       - the residual of tile t is fetched in the same gaps, 32 registers ahead of its use (a 32-register ring
//         covers the HBM latency), and added there;
//   * all tile addressing is scalar: the buffer resources' base addresses are advanced per tile (64-bit SALU), lane
//     offsets never change, out-of-range rows (M tail) are dropped / zero-filled by the buffer range check -- which
//     also means no 2 GiB tensor limit on this path.
// Numerics: exact fp32 products, k-ordered fma chain per output like conv_mfma.hip (which starts the chain from the
// bias; here bias and residual are added after it: one rounding placed differently, ~1e-7 relative).
#include <type_traits>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int BUF = (BM + BN) * BK;          // floats per staging buffer (32 KB)
constexpr int PIPE_LDS_BYTES = 2 * BUF * 4;  // 64 KB of staging (+ 4 B per output channel for the bias): two blocks per CU

struct PipeArgs {
    const float *in, *wgt, *bias, *res;
    float *out;
    long long M;
    int K, N;
    int in_cs, in_coff, out_cs, out_coff;    // res shares out_cs / out_coff (checked by the launcher)
    int panels, NB, grid;
    int units, gshift, NBG;                  // work unit = (panel, group of NBG consecutive N tiles): unit u -> panel u >> gshift
    float lo, hi;                            // activation as a clamp
};

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, float *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
#else
    return 0.f;
#endif
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, soff, 0);
#endif
}

// resource over [ptr + off, ptr + total): num_records saturates at 2^32 - 1 (a tile only ever reaches 128 rows in)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void *ptr, long long off, long long total) {
    // (32-bit halves: 64-bit signed compares have no scalar form and would be evaluated on the VALU)
    const unsigned long long left = (unsigned long long)(total - off);
    const unsigned hi = (unsigned)(left >> 32), lo = (unsigned)left;
    const unsigned rec = (hi & 0x80000000u) ? 0u : (hi ? 0xffffffffu : lo);      // negative -> empty, >= 4 GiB -> saturate
    return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)ptr + off), 0, (int)rec, 0x00020000);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// the same descriptor as four plain dwords, for inline-asm operands (V# of a raw buffer: base, stride 0, num_records, flags)
__device__ __forceinline__ i32x4 rsrc_words(const void *ptr, long long off, long long total) {
    const unsigned long long left = (unsigned long long)(total - off);
    const unsigned hi = (unsigned)(left >> 32), lo = (unsigned)left;
    const unsigned rec = (hi & 0x80000000u) ? 0u : (hi ? 0xffffffffu : lo);
    const unsigned long long base = (unsigned long long)((const char *)ptr + off);
    const i32x4 w = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)rec, 0x00020000};
    return w;
}

// which tile a cursor points at; advanced with scalar adds only
struct Cursor {
    int u, nt;           // work unit and N tile; u >= units: past the end (all its resources are empty)
};

struct PipeState {
    // lane-constant offsets
    int a_voff[4], b_voff[4];     // staging loads, bytes relative to the panel / N-tile base
    int o_voff[2][2];             // C/D-layout element (e = 0) of sub-tile (mi, ni), bytes relative to the tile base
    int a_off, b_off, swz, h;     // fragment read offsets (floats)
    int wave_u;
    // scalars
    int eoff[16];                 // byte offset of accumulator register e inside a 32x32 sub-tile: ((e&3)+8(e>>2)) rows
    float lo, hi;
    float lo_v, hi_v;             // the same in vector registers (asm operands)
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
// which would drain the 64 stores of the epilogue 64 times.  Its completion is waited for by hand (below).
__device__ __forceinline__ float buf_load_asm(i32x4 rsrc, int voff, int soff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}
// One register's epilogue arithmetic as ONE asm block (pinned where it is written, no hazard padding, no copies):
//   wait until the residual load with N younger vector-memory operations has landed; clamp(x + r + bias, lo, hi).
// lo / hi / bias are VGPRs (v_med3 takes one scalar operand at most on gfx9).
template <int N>
__device__ __forceinline__ float finish_res(float x, float r, float bias, float lo, float hi) {
    float o;
    asm volatile("s_waitcnt vmcnt(%6)\n\tv_add_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3\n\tv_med3_f32 %0, %0, %4, %5"
                 : "=&v"(o) : "v"(x), "v"(r), "v"(bias), "v"(lo), "v"(hi), "n"(N));
    return o;
}
__device__ __forceinline__ float finish_nores(float x, float bias, float lo, float hi) {
    float o;
    asm volatile("v_add_f32 %0, %1, %2\n\tv_med3_f32 %0, %0, %3, %4" : "=&v"(o) : "v"(x), "v"(bias), "v"(lo), "v"(hi));
    return o;
}

// One K chunk (32 deep): 64 MFMAs of the wave's 64x64 sub-tile, with the non-matrix work pinned between them.
//   always     : the 8 LDS-direct loads of the NEXT chunk (may belong to the next tile), one after each of MFMA 0..7
//   FIRST chunk: the PREVIOUS tile's 64 accumulator registers (`oth`) in 64 + LEAD "register pieces":
//                piece p: [p >= LEAD] register q = p - LEAD: (+ residual) + bias, clamp, store;
//                         [p < 64, residual] load the residual of register p (consumed LEAD pieces later: a LEAD-deep
//                         register ring covers the HBM latency).
// Vector-memory issue order of a FIRST chunk, on which the hand-counted waits rely (sched_barrier pins it):
//   8 staging loads, then per piece: store(q) before load(p).
constexpr int RES_LEAD = 32;
template <bool FIRST, bool HAS_RES>
__device__ __forceinline__ void pipe_chunk(f32x16 (&cur)[2][2], f32x16 (&oth)[2][2], const PipeState &S, const float *rd,
                                           float *wr, __amdgpu_buffer_rsrc_t ra_nx, __amdgpu_buffer_rsrc_t rb_nx, int soff_nx,
                                           __amdgpu_buffer_rsrc_t ro_prev, const float (&bias_prev)[2], i32x4 rr_prev) {
    constexpr int LEAD = HAS_RES ? RES_LEAD : 0;
    constexpr int NREG = FIRST ? 64 + LEAD : 0;          // register pieces
    float ring[RES_LEAD];
    // fragments are double buffered: the ds_reads of k-step ks+1 are issued in the middle of k-step ks's 16 MFMAs
    f32x4 fa[2][2], fb[2][2];
    auto read_frags = [&](int ks, f32x4 (&a)[2], f32x4 (&b)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
            a[m] = *reinterpret_cast<const f32x4 *>(rd + S.a_off + m * 32 * BK + (((ks * 2 + S.h) ^ S.swz) * 4));
#pragma unroll
        for (int n = 0; n < 2; ++n)
            b[n] = *reinterpret_cast<const f32x4 *>(rd + S.b_off + n * 32 * BK + (((ks * 2 + S.h) ^ S.swz) * 4));
    };
    read_frags(0, fa[0], fb[0]);
    static_for<0, 64>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;          // MFMA number: ((ks*4 + j)*2 + mi)*2 + ni
        constexpr int ni = idx & 1, mi = (idx >> 1) & 1, j = (idx >> 2) & 3, ks = idx >> 4;
        f32x4 (&a)[2] = fa[ks & 1];
        f32x4 (&b)[2] = fb[ks & 1];
        if constexpr ((idx & 15) == 6 && ks < 3) read_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        if constexpr (FIRST && ks == 0 && j == 0) {       // a tile's chain starts from C = 0 (inline constant)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], zero, 0, 0, 0);
        } else {
            cur[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], cur[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef PIPE_ABL_NODMA
        if constexpr (false) {
#else
        if constexpr (idx < 4) {
#endif
            lds_dma16(ra_nx, wr + (32 * idx + 8 * S.wave_u) * BK, S.a_voff[idx], soff_nx);
#ifdef PIPE_ABL_NODMA
        } else if constexpr (false) {
#else
        } else if constexpr (idx < 8) {
#endif
            lds_dma16(rb_nx, wr + (BM + 32 * (idx - 4) + 8 * S.wave_u) * BK, S.b_voff[idx - 4], soff_nx);
#ifdef PIPE_ABL_NOEPI
        } else if constexpr (false) {
#else
        } else if constexpr (FIRST) {
#endif
            // register pieces [pb, pe) ride after this MFMA: spread evenly over MFMAs 8..63
            constexpr int pb = ((idx - 8) * NREG) / 56, pe = ((idx - 7) * NREG) / 56;
            static_for<pb, pe>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if constexpr (p >= LEAD) {
                    constexpr int q = p - LEAD;
                    constexpr int e = q & 15, qn = (q >> 4) & 1, qm = q >> 5;
                    const float x = oth[qm][qn][e];
                    float v;
                    if constexpr (HAS_RES) {
                        // vector-memory operations younger than load(q) at this point: loads q+1 .. min(p, 64) - 1
                        // and stores max(0, q - LEAD + 1) .. q - 1
                        constexpr int loads_issued = p < 64 ? p : 64;
                        constexpr int younger = (loads_issued - q - 1) + (q - (q >= LEAD ? q - LEAD + 1 : 0));
                        static_assert(younger >= 0 && younger <= 63, "vmcnt immediate out of range");
                        v = finish_res<younger>(x, ring[q % RES_LEAD], bias_prev[qn], S.lo_v, S.hi_v);
                    } else {
                        v = finish_nores(x, bias_prev[qn], S.lo_v, S.hi_v);
                    }
                    buf_store(v, ro_prev, S.o_voff[qm][qn], S.eoff[e]);
                }
                if constexpr (HAS_RES && p < 64) {
                    constexpr int e = p & 15, pn = (p >> 4) & 1, pm = p >> 5;
                    ring[p % RES_LEAD] = buf_load_asm(rr_prev, S.o_voff[pm][pn], S.eoff[e]);
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <bool HAS_RES>
__global__ void __launch_bounds__(256, 2)
conv1x1_pipe_kernel(const PipeArgs A) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int ld_row = tid >> 3;
    const int ld_c = ((tid & 7) ^ ((ld_row >> 1) & 7)) * 4;       // fetch the k-group whose (swizzled) slot this lane fills

    PipeState S;
    S.h = h;
    S.swz = (r >> 1) & 7;
    S.a_off = (wm * 64 + r) * BK;
    S.b_off = BM * BK + (wn * 64 + r) * BK;
    S.wave_u = __builtin_amdgcn_readfirstlane(wave);
    S.lo = A.lo;
    S.hi = A.hi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(S.lo_v) : "s"(A.lo));      // keep them in VGPRs (not re-materialised per use)
    asm volatile("v_mov_b32 %0, %1" : "=v"(S.hi_v) : "s"(A.hi));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        S.a_voff[i] = ((ld_row + 32 * i) * A.in_cs + A.in_coff + ld_c) * 4;
        S.b_voff[i] = ((ld_row + 32 * i) * A.K + ld_c) * 4;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            S.o_voff[mi][ni] = ((wm * 64 + mi * 32 + 4 * h) * A.out_cs + A.out_coff + wn * 64 + ni * 32 + r) * 4;
#pragma unroll
    for (int e = 0; e < 16; ++e) S.eoff[e] = ((e & 3) + 8 * (e >> 2)) * A.out_cs * 4;

    const int nk = A.K / BK;
    const long long in_total = A.M * (long long)A.in_cs * 4;
    const long long out_total = A.M * (long long)A.out_cs * 4;
    const long long w_total = (long long)A.N * A.K * 4;
    const int gmask = (1 << A.gshift) - 1;
    auto panel_of = [&](Cursor c) { return c.u >> A.gshift; };
    auto res_a = [&](Cursor c) { return rsrc_at(A.in, (long long)panel_of(c) * BM * A.in_cs * 4, c.u < A.units ? in_total : 0); };
    auto res_b = [&](Cursor c) { return rsrc_at(A.wgt, (long long)c.nt * BN * A.K * 4, c.u < A.units ? w_total : 0); };
    auto res_o = [&](const float *base, Cursor c) {
        // rows past M must fall outside: the extent is counted from the tile's first element
        return rsrc_at(base, ((long long)panel_of(c) * BM * A.out_cs + (long long)c.nt * BN) * 4,
                       c.u < A.units ? out_total : 0);
    };
    auto res_words = [&](Cursor c) {                       // the residual tile as asm operand words
        return rsrc_words(A.res, ((long long)panel_of(c) * BM * A.out_cs + (long long)c.nt * BN) * 4,
                          (HAS_RES && c.u < A.units) ? out_total : 0);
    };
    auto first_nt = [&](int u) { return (u & gmask) * A.NBG; };
    auto advance = [&](Cursor c) {
        Cursor n = c;
        if (n.nt + 1 < first_nt(n.u) + A.NBG) { ++n.nt; } else { n.u += A.grid; n.nt = first_nt(n.u); }
        return n;
    };
    // the bias vector lives in LDS behind the staging buffers (read with ds_read at a tile switch: a global load
    // there would be waited for with vmcnt(0) while LDS-direct loads are in flight)
    float *lds_bias = lds + 2 * BUF;
    for (int i = tid; i < A.N; i += 256) lds_bias[i] = A.bias ? A.bias[i] : 0.f;
    __syncthreads();
    auto load_bias = [&](Cursor c, float (&bv)[2]) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bv[ni] = lds_bias[c.nt * BN + wn * 64 + ni * 32 + r];
    };

    Cursor cur = {(int)blockIdx.x, first_nt((int)blockIdx.x)};
    if (cur.u >= A.units) return;
    Cursor nx = cur;                      // tile of the NEXT chunk to stage
    int kc_nx = 0;

    f32x16 accA[2][2], accB[2][2];
    float bias_cur[2], bias_prev[2] = {0.f, 0.f};
    load_bias(cur, bias_cur);
    // ---- prologue: first chunk of the first tile
    {
        const __amdgpu_buffer_rsrc_t ra = res_a(nx), rb = res_b(nx);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ra, lds + (32 * i + 8 * S.wave_u) * BK, S.a_voff[i], 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rb, lds + (BM + 32 * i + 8 * S.wave_u) * BK, S.b_voff[i], 0);
        kc_nx = 1;
        if (kc_nx == nk) { kc_nx = 0; nx = advance(nx); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    Cursor prev = {A.units, 0};           // nothing to store yet: empty resource
    int buf = 0;
    // One tile: `acc` accumulates it, `oth` holds the previous tile's result (finished and stored during the first
    // chunk).  Returns false after the block's last tile.
    auto run_tile = [&](f32x16 (&acc)[2][2], f32x16 (&oth)[2][2]) -> bool {
        const Cursor next = advance(cur);
        const bool has_next = next.u < A.units;
        const __amdgpu_buffer_rsrc_t ro_prev = res_o(A.out, prev);
        const i32x4 rr_prev = res_words(prev);
        for (int kc = 0; kc < nk; ++kc) {
            const __amdgpu_buffer_rsrc_t ra_nx = res_a(nx), rb_nx = res_b(nx);
            const int soff_nx = kc_nx * (BK * 4);
            const float *rd = lds + buf * BUF;
            float *wr = lds + (buf ^ 1) * BUF;
            const bool first = kc == 0;
            // (wave-uniform branch: two instantiations of the chunk per accumulator role)
            if (first) pipe_chunk<true, HAS_RES>(acc, oth, S, rd, wr, ra_nx, rb_nx, soff_nx, ro_prev, bias_prev, rr_prev);
            else pipe_chunk<false, HAS_RES>(acc, oth, S, rd, wr, ra_nx, rb_nx, soff_nx, ro_prev, bias_prev, rr_prev);
            // advance the staging cursor
            ++kc_nx;
            if (kc_nx == nk) { kc_nx = 0; nx = advance(nx); }
            buf ^= 1;
            // FIRST chunks end with 64 stores younger than the 8 staging loads: wait for the loads only
            if (first) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        prev = cur;
        bias_prev[0] = bias_cur[0];
        bias_prev[1] = bias_cur[1];
        cur = next;
        if (has_next) load_bias(cur, bias_cur);
        return has_next;
    };
    // ---- drain: the last tile's registers have not been stored yet
    auto drain = [&](f32x16 (&acc)[2][2]) {
        const __amdgpu_buffer_rsrc_t ro = res_o(A.out, prev);
        float resv[2][2][16];
        if constexpr (HAS_RES) {
            const i32x4 rr = res_words(prev);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) resv[mi][ni][e] = buf_load_asm(rr, S.o_voff[mi][ni], S.eoff[e]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(resv[mi][ni][e]));     // uses stay below the wait
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[mi][ni][e];
                    if constexpr (HAS_RES) v += resv[mi][ni][e];
                    v += bias_prev[ni];
                    v = __builtin_amdgcn_fmed3f(v, S.lo, S.hi);
                    buf_store(v, ro, S.o_voff[mi][ni], S.eoff[e]);
                }
    };
    // the tile loop, unrolled by two so that the accumulator sets swap roles without register copies
    for (;;) {
        if (!run_tile(accA, accB)) { drain(accA); break; }
        if (!run_tile(accB, accA)) { drain(accB); break; }
    }
}

}  // namespace

// -> ML_OK and *eligible = 1 when the problem was launched on the pipelined kernel; *eligible = 0 when it is not a
// problem this kernel handles (the caller then uses the generic implicit-GEMM kernel).
int ml_conv1x1_pipe_eligible(const ml_conv2d_desc &d) {
    const bool shape_ok = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.dil == 1 && d.pad_t == 0 && d.pad_l == 0 &&
                          d.cpp_shift == 30 && d.group_cin_step == 0 && d.shuffle2x2 == 0 && d.out_bstride == 0 &&
                          d.math == ML_MATH_F32 && d.Ho == d.H && d.Wo == d.W;
    if (!shape_ok) return 0;
    if (d.span % 32 != 0 || d.span < 64 || d.span != d.span_pad || d.cout % 128 != 0 || d.n_pad != d.cout) return 0;
    if (d.cout > 4096) return 0;                          // bias vector in LDS
    if (d.act == ML_ACT_SIGMOID) return 0;
    if (d.in_cstride % 4 || d.in_coff % 4 || !ml_aligned16(d.in) || !ml_aligned16(d.wgt)) return 0;
    if (d.residual && (d.res_cstride != d.out_cstride || d.res_coff != d.out_coff)) return 0;
    // lane offsets are 32-bit and cover one 128-row tile only
    if ((long long)BM * d.in_cstride * 4 >= (1ll << 31) || (long long)(BM + 32) * d.out_cstride * 4 >= (1ll << 31)) return 0;
    return 1;
}

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
    // enough work units to fill 2 blocks on each of the 256 CUs: split a panel's N tiles into 2^gshift groups
    A.gshift = 0;
    while ((long long)A.panels << A.gshift < 512 && (A.NB >> A.gshift) % 2 == 0 && (A.NB >> A.gshift) > 1) ++A.gshift;
    A.NBG = A.NB >> A.gshift;
    A.units = A.panels << A.gshift;
    A.grid = A.units < 512 ? A.units : 512;
    A.lo = d.act == ML_ACT_NONE ? -3.402823466e38f : 0.f;
    A.hi = d.act == ML_ACT_RELU6 ? 6.f : 3.402823466e38f;
    const int lds_bytes = PIPE_LDS_BYTES + 4 * d.cout;
    if (d.residual) {
        static std::atomic<unsigned long long> ok{0};
        if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(conv1x1_pipe_kernel<true>), PIPE_LDS_BYTES + 4 * 4096, ok, "conv1x1_pipe")) return rc;
        hipLaunchKernelGGL(conv1x1_pipe_kernel<true>, dim3(A.grid), dim3(256), lds_bytes, s, A);
    } else {
        static std::atomic<unsigned long long> ok{0};
        if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(conv1x1_pipe_kernel<false>), PIPE_LDS_BYTES + 4 * 4096, ok, "conv1x1_pipe")) return rc;
        hipLaunchKernelGGL(conv1x1_pipe_kernel<false>, dim3(A.grid), dim3(256), lds_bytes, s, A);
    }
    ML_CHECK_LAUNCH("conv1x1_pipe");
    *eligible = 1;
    return ML_OK;
}
