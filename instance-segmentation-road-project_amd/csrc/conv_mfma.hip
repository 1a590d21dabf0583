// Implicit-GEMM NHWC convolution on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   M = B*Ho*Wo output pixels, N = output channels, K = taps x span_pad.
//   A (pixels x K) is gathered straight from the NHWC activation: for one tap the `span`
//   input channels of a pixel are contiguous, so a K-chunk of 32 floats is one 128-byte line
//   per output pixel -- no im2col buffer.  Out-of-image taps use the buffer out-of-range rule (zeros land in LDS).
//   B (N x K) is the host-packed weight matrix, k contiguous.
//   Block = 256 threads = 4 waves; wave tile = TM x TN MFMA tiles of 32x32; K-chunk = 32 floats (128 B per row),
//   staged with LDS-direct buffer loads (buffer_load_dwordx4 ... lds: no register round trip, no ds_write) into
//   unpadded 128-byte rows whose 16-byte k-groups are XOR-swizzled (conflict-free ds_read_b128), double buffered
//   so the next chunk's loads -- issued in pieces between the MFMAs -- fly under the current chunk's 64 MFMAs.
//   Lane (r = lane&31, h = lane>>5) reads 4 consecutive k per ds_read_b128; MFMA step j pairs
//   k = 8*ks + j (h=0) with k = 8*ks + 4 + j (h=1) -- A and B use the same pairing, so the sum
//   over K is unchanged.  Result = a k-ordered fp32 fma chain (exact fp32, no reduced precision).
//
//   Epilogue: the accumulator tile is transposed through LDS (the staging buffers are dead by
//   then) so every lane owns 4 consecutive output channels of one pixel: bias / residual /
//   activation run on float4 and a wave stores 2 x 512 contiguous bytes per instruction (the
//   scalar C/D layout would store 128-byte fragments and serialise the residual loads).
//   Destinations: a channel slice of a wider buffer (concat fusion), a per-image strided view
//   (Reshape+Concatenate of the detection heads), or a 2x2 pixel shuffle (Conv2DTranspose).
//
//   One launch can carry several independent problems of the same tile shape (the un-shared
//   head towers run the same conv at 5 pyramid levels: the 4-block P7 problem rides along with
//   the 1024-block P3 problem instead of being its own latency-bound launch), and a problem with
//   few tiles but a long K can be split along K: slices write raw partial tiles to a workspace
//   slab and a second kernel reduces them in a FIXED order (deterministic, no atomics).
//   Block -> tile map is XCD aware: the NB column tiles that share one 128-pixel A panel get
//   ids congruent mod 8, i.e. the same XCD / L2; for spatial (3x3 ...) convs each XCD additionally takes a
//   CONTIGUOUS range of panels, so the halo rows two neighbouring panels share are fetched into one L2.
//   1x1 convs with K <= 512 on maps of >= 64x64 pixels go to the persistent pipelined kernel (conv1x1_pipe.hip)
//   instead; activations >= 2 GiB are cut into image groups (split_by_image_groups below).
#include "common.h"


namespace {

constexpr int LDS_LD = 32;   // f32 math: floats per staged row.  No padding: rows are filled by LDS-direct loads
                             // (64 lanes x 16 B land contiguously); bank conflicts are avoided by an XOR swizzle
                             // of the 16-byte k-groups inside a row: slot = kgroup ^ ((row >> 1) & 7)
constexpr int LDS_LD_H = 40; // f16 math: halves per staged row (32 + 8 pad: 80-byte rows, conflict-free b128 reads)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// (out-of-image taps: the buffer loads below return zeros for out-of-range offsets -- no post-load select,
// which would force an s_waitcnt vmcnt before the MFMA block)
constexpr int MAXP = ML_CONV_MAX_PROBLEMS;

// Exact unsigned division by a launch-invariant divisor for 0 <= n < 2^31 (Granlund-Montgomery):
// s = ceil(log2 d), mul = ceil(2^(31+s) / d) (< 2^32), n / d = umulhi(n, mul) >> (s - 1).  d == 1: shift < 0.
struct FastDiv { unsigned mul; int shift; };

static FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    if (d <= 1) { f.mul = 0; f.shift = -1; return f; }
    int s = 0;
    while ((1ull << s) < d) ++s;
    const unsigned long long num = 1ull << (31 + s);
    f.mul = (unsigned)((num + d - 1) / d);
    f.shift = s - 1;
    return f;
}

__device__ __forceinline__ int fast_div(int n, const FastDiv f) {
    return f.shift < 0 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shift);
}

struct Problem {
    ml_conv2d_desc d;
    FastDiv div_howo, div_wo;
    unsigned in_bytes, wgt_bytes;   // buffer-resource extents (num_records) of the activation / weight tensors
    int MB, NB, M, ncpt, ktot;
    int splits, cps;          // K slices and chunks per slice (splits == 1: direct epilogue)
    int blocks_per_split;     // 8*ceil(MB/8)*NB
    float *slab;              // [splits][MB*BM][n_pad] when splits > 1
};

struct MultiArgs {
    int n;
    int start[MAXP + 1];      // prefix sum of blocks per problem
    Problem p[MAXP];
};

// buffer_load_dwordx4 ... lds: 16 bytes per lane from (resource + voff + soff) straight into LDS at
// dst + lane * 16 (dst is wave-uniform and travels in M0); out-of-range lanes write zeros.  Counted in vmcnt.
// (The builtin only exists in the device pass.)
__device__ __forceinline__ void load_b128_to_lds(__amdgpu_buffer_rsrc_t rsrc, float *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}

// ---- shared epilogue math: bias + residual + activation + addressing for 4 consecutive channels
__device__ __forceinline__ void store_out4(const ml_conv2d_desc &p, int HoWo, int m, int n, f32x4 v, bool vec_ok,
                                           bool add_residual = true) {
    const int co = p.shuffle2x2 ? (p.cout >> 2) : p.cout;
    int ab = 0, o = n;
    if (p.shuffle2x2) { ab = n / co; o = n - ab * co; }
    size_t base;
    if (p.shuffle2x2) {
        const int b = m / HoWo;
        const int rr = m - b * HoWo;
        const int oy = rr / p.Wo;
        const int ox = rr - oy * p.Wo;
        base = (((size_t)b * (2 * p.Ho) + 2 * oy + (ab >> 1)) * (size_t)(2 * p.Wo) + 2 * ox + (ab & 1)) * p.out_cstride;
    } else if (p.out_bstride) {
        const int b = m / HoWo;
        base = (size_t)b * (size_t)p.out_bstride + (size_t)(m - b * HoWo) * p.out_cstride;
    } else {
        base = (size_t)m * p.out_cstride;
    }
    base += p.out_coff + o;
    if (vec_ok) {
        if (p.bias) v += *reinterpret_cast<const f32x4 *>(p.bias + o);
        if (p.residual && add_residual)
            v += *reinterpret_cast<const f32x4 *>(p.residual + (size_t)m * p.res_cstride + p.res_coff + n);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ml_apply_act(v[e], p.act);
        if (p.out_f16) {
            const f16x4 hv = {(_Float16)r[0], (_Float16)r[1], (_Float16)r[2], (_Float16)r[3]};
            *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(p.out) + base) = hv;
        } else {
            *reinterpret_cast<f32x4 *>(p.out + base) = r;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= p.cout) break;
            float x = v[e];
            if (p.bias) x += p.bias[o + e];
            if (p.residual) x += p.residual[(size_t)m * p.res_cstride + p.res_coff + n + e];
            x = ml_apply_act(x, p.act);
            if (p.out_f16) reinterpret_cast<_Float16 *>(p.out)[base + e] = (_Float16)x;
            else p.out[base + e] = x;
        }
    }
}

__device__ __forceinline__ bool out_vec_ok(const ml_conv2d_desc &p) {
    const int co = p.shuffle2x2 ? (p.cout >> 2) : p.cout;
    bool ok = (co % 4 == 0) && (p.out_cstride % 4 == 0) && (p.out_coff % 4 == 0) && (p.out_bstride % 4 == 0) &&
              ((((uintptr_t)p.out) & 15) == 0);
    if (p.bias) ok = ok && ((((uintptr_t)p.bias) & 15) == 0);
    if (p.residual) ok = ok && (p.res_cstride % 4 == 0) && (p.res_coff % 4 == 0) && ((((uintptr_t)p.residual) & 15) == 0);
    return ok;
}

// (split_hi_lo, split_hi_lo_pair: common.h)


// F16 = true: the "fp16 MFMA path" (BASELINE config 5): activations and weights stay fp32 in HBM, are
// rounded to fp16 (RNE) on their way into LDS, and the contraction runs on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation -- 2 instructions of 32 cycles per 32-deep chunk and tile pair instead of 16 of 64.
// Everything outside the K loop (addressing, prefetch pieces, epilogue, split-K) is shared.
// MATH = ML_MATH_F16S: fp16 STORAGE (the heads of the fp16 path): activations and weights are IEEE half in HBM.  A K
// chunk is still 128 bytes per row (64 halves), so staging (LDS-direct loads, XOR swizzle), fragment addressing and the
// prefetch pieces are the f32 code with the element size changed; a chunk is 4 k-steps of v_mfma_f32_32x32x16_f16
// (one ds_read_b128 = the 8 halves a lane feeds).  Accumulation, bias and activation are fp32; the output is half
// (`out_f16`, one rounding at the store) or fp32 (the prediction tensors detect.hip reads).
// MATH = ML_MATH_F32X3: fp32 tensors, fp32-grade products on the f16 matrix pipe.  Every operand is written as
// x = hi + 2^-11 lo (two halves, 22 bits, split_hi_lo above; the weights arrive already split, masklab_hip.h) and a product as
// hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) -- three v_mfma_f32_32x32x16_f16 per 16-deep step into TWO fp32 accumulator sets
// (the cross terms keep their own, folded in once at the end).  Each f16 x f16 product is exact in fp32; what is dropped is
// lo_a lo_b <= 2^-22 |a b|, below the rounding of the fp32 accumulation itself.  3 x 8 passes against 8 x 16 for the same
// 16-deep step on v_mfma_f32_32x32x2_f32: 5.3 x the fp32 matrix rate.  Staging, addressing, epilogue: the f32 code.
// GNS = the conv -> (activation) -> GroupNormalization pairs of the heads (engine/layers/detection.py:120-125,
// semantic.py:205-213): the epilogue also sums its tile's stored values (sum, sum of squares: a float4 folded in fp32, then
// fp64 -- the same folding as gn_stats_kernel) and writes one pair per WAVE (4 per tile) to `gn_partials`; the GroupNorm
// apply pass adds the pairs of a chunk's tiles in order instead of re-reading the tensor (csrc/groupnorm.hip).  A separate
// instantiation: the kernel every other conv runs is untouched.
template <int WAVES_M, int WAVES_N, int TM, int TN, int MATH, bool GNS = false, int NSTAGE = 2>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (GNS || MATH == ML_MATH_F32X3) ? 2 : 1)      // (second figure: waves per SIMD;
conv_mfma_kernel(const MultiArgs args) {                                          //  GNS: 262 registers otherwise)
    constexpr int NT = 64 * WAVES_M * WAVES_N;     // threads per block: 256 (4 waves), or 512 for the 256-row X3 tile
    constexpr int RPP = NT / 8;                    // rows one staging pass of the block covers (8 lanes x 16 B per row)
    static_assert(!GNS || (NT == 256 && WAVES_M * TM == 4) || (NT == 512 && WAVES_M * TM == 8),
                  "the GroupNorm partial sums: 4 pairs per 128-row tile, from 4 waves or from 8 waves on 256 rows");
    static_assert(NSTAGE == 2 || (NSTAGE == 3 && MATH == ML_MATH_F32X3), "the 3-deep ring is the X3 form");
    constexpr bool F16 = MATH == ML_MATH_F16;      // fp32 tensors, converted on the way into (padded, half) LDS rows
    constexpr bool HS = MATH == ML_MATH_F16S;      // half tensors, staged like fp32 ones
    constexpr bool X3 = MATH == ML_MATH_F32X3;     // fp32 tensors, staged like ML_MATH_F32; split products on the f16 MFMA
    constexpr int ES = HS ? 2 : 4;                 // bytes per tensor element
    constexpr int KC = HS ? 64 : 32;               // elements per K chunk
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    constexpr int A_LD = BM / RPP;  // float4 loads per thread per chunk
    constexpr int B_LD = BN / RPP;
    constexpr int BUF = F16 ? (BM + BN) * LDS_LD_H / 2 : (BM + BN) * LDS_LD;    // floats per staging buffer
    constexpr int C_LD = BN + 4;   // epilogue tile row stride (floats)
    // (the launcher sizes the dynamic LDS as max(two staging buffers, epilogue tile))
    extern __shared__ __align__(16) float lds[];

    // ---- which problem / tile / K slice
    int pi = 0;
    while (pi + 1 < args.n && (int)blockIdx.x >= args.start[pi + 1]) ++pi;
    const Problem &P = args.p[pi];
    const ml_conv2d_desc &p = P.d;
    int id = blockIdx.x - args.start[pi];
    const int slice = id / P.blocks_per_split;
    id -= slice * P.blocks_per_split;
    const int NB = P.NB, M = P.M, ncpt = P.ncpt, ktot = P.ktot;
    // XCD-aware tile assignment: blocks whose ids are congruent mod 8 share an XCD (and its L2).  The NB column tiles of
    // one 128-pixel A panel always do.  With a spatial kernel, vertically adjacent panels read each other's rows (taps
    // kh = 0..2): each XCD then takes a CONTIGUOUS range of panels, so a halo row is fetched into one L2, not three
    // (round-robin panels measured 2.0x the algorithmic HBM bytes on the tower convs, profiles/r02a_traffic.json).
    const int xcd = id & 7;
    const int jj = id >> 3;
    // (fixed-capacity RoI batches keep round-robin panels: their live tiles are the FIRST slots of every image, which a
    // contiguous split would pile onto a few XCDs -- 129 us instead of ~45 for the 1-image mask head)
    const int mt = (p.KH > 1 && !p.live) ? xcd * ((P.MB + 7) >> 3) + jj / NB : (jj / NB) * 8 + xcd;
    const int nt = jj % NB;
    if (mt >= P.MB) return;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    // Fixed-capacity RoI batches (the mask head without a host read of the RoI counts): image i of this problem is LIVE
    // iff i % live_period < max(1, *live).  A tile all of whose images are dead computes nothing and stores nothing
    // (nobody reads those rows); rows of dead images inside a live tile are computed from whatever their input holds.
    if (p.live) {
        const int lim = max(1, *p.live);
        // decided on ABSOLUTE image indices: the tile's images a0..a1 are all dead iff they lie in ONE period (then their
        // slots are i0..i1 with i1 - i0 == a1 - a0) and the first slot is already past the limit.  (A tile of three or
        // more small images can cross a period boundary with i0 <= i1: slot 0 of the next image is live.)
        const int a0 = fast_div(m0, P.div_howo), a1 = fast_div(min(m0 + BM, P.M) - 1, P.div_howo);
        const int i0 = a0 % p.live_period, i1 = a1 % p.live_period;
        if (i0 >= lim && a1 - a0 == i1 - i0) return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int ld_row = tid >> 3;
    // f32 math: lane (row, slot) of a staging load FETCHES k-group slot ^ ((row >> 1) & 7), so the linear LDS
    // image the load writes is the swizzled one (row + 32 i has the same swizzle key)
    // (in ELEMENTS: a 16-byte k-group is 4 floats or 8 halves)
    const int ld_c = F16 ? (tid & 7) * 4 : (((tid & 7) ^ ((ld_row >> 1) & 7)) * (16 / ES));

    // ---- per-thread pixel coordinates of the A rows it stages.  a_voff[i] = byte offset of the
    // row's tap-(0,0) pixel, channel gofs + ld_c: per chunk only a wave-uniform (tap, channel) offset
    // is added, so the K loop's address arithmetic is one 64-bit add + two compares per row.
    int a_iy0[A_LD], a_ix0[A_LD];
    int a_voff[A_LD];          // byte offset (may be "negative" for halo rows; only used when the tap is valid)
    const int HoWo = p.Ho * p.Wo;
    const int gofs = p.in_coff + nt * p.group_cin_step;
    // 1x1 / stride-1 / unpadded convs (most of the backbone): every tap of a valid row is inside the image,
    // so the per-row compares vanish and invalid tail rows rely on their out-of-range base offset.
    const bool nohalo = (p.KH == 1) && (p.KW == 1) && (p.pad_t == 0) && (p.pad_l == 0) && (p.stride == 1) &&
                        (p.cpp_shift == 30) && (p.span % KC == 0);
    // Setup and epilogue run beside the co-resident block's MFMA stream, which leaves them about one VALU
    // issue slot per 64-cycle MFMA (measured: 2 400 cycles for the ~140 instructions below, 9 100 for the
    // ~150 of the store loop) -- so what counts here is the instruction COUNT, not the latency.
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + ld_row + RPP * i;
        if (nohalo) {
            // output pixel m reads input pixel m: no (b, y, x) decomposition, no divisions
            a_iy0[i] = 0;
            a_ix0[i] = 0;
            a_voff[i] = m < M ? (int)(((long long)m * (long long)p.in_cstride + gofs + ld_c) * ES) : (int)0x80000000;
        } else if (m < M) {
            const int b = fast_div(m, P.div_howo);
            const int r = m - b * HoWo;
            const int oy = fast_div(r, P.div_wo);
            const int ox = r - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
            a_voff[i] = (int)((((long long)b * p.H * p.W + (long long)a_iy0[i] * p.W + a_ix0[i]) * (long long)p.in_cstride +
                               gofs + ld_c) * ES);
        } else {
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
            a_voff[i] = (int)0x80000000;       // stays out of range after adding a (small, positive) tap offset
        }
    }
    // hot descriptor fields in registers: the descriptor lives in kernarg memory and would be
    // re-fetched (s_load + lgkmcnt(0), which also drains LDS) inside the K loop otherwise
    const int pH = p.H, pW = p.W, pKW = p.KW, pdil = p.dil, pspan = p.span, pshift = p.cpp_shift;
    const long long pcs = p.in_cstride;
    const float *pin = p.in;

    // chunk range of this K slice
    const int total_chunks = p.KH * p.KW * ncpt;
    const int kc_begin = slice * P.cps;
    const int kc_end = min(kc_begin + P.cps, total_chunks);
    // state of the NEXT chunk to load: kc -> (cc, kh, kw), TAPS INNERMOST.  All taps of one 32-channel slice run back to back:
    // the blocks of an XCD (a contiguous range of panels, in step with one another) then work on rows x one channel slice
    // -- ~1 MB at a 128-wide map -- for KH x KW chunks in a row, which the 4 MB L2 holds; with the channel slices innermost
    // every tap walked the whole channel depth (4 MB and more) before the next tap came back to the same lines, and a
    // 3x3 conv fetched its input 2-8 times from beyond L2 (round 3 shape PMC).  The weights stay packed tap-major.
    const int ntaps = p.KH * p.KW;
    int cc = kc_begin / ntaps;
    const int tap0 = kc_begin - cc * ntaps;
    int kw = tap0 % p.KW;
    int kh = tap0 / p.KW;

    // ---- residual tile prefetch: the epilogue's 4-channel x 16-row ownership is known up front, so
    // the (HBM-latency-bound) residual reads are issued now and fly under the whole K loop.
    constexpr int V_PER_ROW = BN / 4;               // float4 per tile row
    constexpr int ROWS_PER_PASS = NT / V_PER_ROW;   // rows covered by the block per pass
    constexpr int E_ROWS = BM / ROWS_PER_PASS;      // rows per thread in the epilogue
    const int c4 = (tid % V_PER_ROW) * 4;
    const int r0 = tid / V_PER_ROW;
    const int n = n0 + c4;
    const bool direct = (P.splits == 1);
    const bool vec_ok = out_vec_ok(p) && (n + 4 <= p.cout);
    f32x4 res[E_ROWS];
    // (X3: two accumulator sets at two blocks per CU leave no room for 32 prefetched registers: the residual is read in the epilogue)
    const bool pre_res = !X3 && direct && p.residual && vec_ok && !p.shuffle2x2 && p.act != ML_ACT_SIGMOID;
    if (pre_res) {
#pragma unroll
        for (int i = 0; i < E_ROWS; ++i) {
            const int m = m0 + r0 + i * ROWS_PER_PASS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M) v = *reinterpret_cast<const f32x4 *>(p.residual + (size_t)m * p.res_cstride + p.res_coff + n);
            res[i] = v;
        }
    }

    f32x4 areg[A_LD], breg[B_LD];
    // running wave-uniform state of the NEXT chunk: tap offsets (dy, dx), element offset toff of
    // (tap, channel chunk) relative to a row's tap-(0,0) pixel -- updated with scalar adds only
    int dy = kh * pdil, dx = kw * pdil;
    int toff = (int)(((long long)dy * pW + dx) * pcs) + cc * KC;
    const int pKH = p.KH;
    const int step_kw = (int)(pdil * pcs);
    const int step_kh = (int)((long long)pdil * pW * pcs) - (pKW - 1) * (int)(pdil * pcs);
    const int step_cc = KC - (int)(((long long)(pKH - 1) * pdil * pW + (long long)(pKW - 1) * pdil) * pcs);
    // The next chunk's prefetch is split into PIECES (one A row or one B row each: ~12 VALU + 1 global
    // load) that the K loop pins between individual MFMAs with sched_barrier(0): a wave that issues MFMAs
    // back to back owns its SIMD's issue port, so non-MFMA work only overlaps matrix work when it sits in
    // the 64-cycle shadow of the wave's OWN MFMAs.  All pieces are branch-free (selects / masks only).
    int nx_px = 0;
    bool nx_cok = true;
    int dst_buf = 0;                                      // staging buffer the pieces fill
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto piece_begin = [&]() {
        const int c = cc * KC + ld_c;
        nx_px = c >> pshift;                     // 0 unless a tap spans pixels (NHWC4 stems)
        nx_cok = c < pspan;
    };
    // Buffer loads: address = resource base + per-lane byte offset (+ scalar offset); any offset >= num_records
    // returns 0, so out-of-image taps need no zero line and no select on the DATA -- one v_cndmask on the offset.
    // The prefetch of the iteration after the last one is issued all the same (branch-free pieces) but through
    // resources with num_records = 0: every lane is out of range, nothing is fetched (it used to re-read the last
    // chunk: +50 % loads on K = 64 convs, +25 % on K = 128).  One s_cselect per chunk.
    const int in_bytes = (int)P.in_bytes, wgt_bytes = (int)P.wgt_bytes;
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)pin, 0, in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, wgt_bytes, 0x00020000);
    int b_voff[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) b_voff[i] = ((n0 + ld_row + RPP * i) * ktot + ld_c) * ES;
    auto piece_a = [&](int i) {
        int vo = a_voff[i] + toff * ES;
        if (!nohalo) {                            // wave-uniform
            const int iy = a_iy0[i] + dy;
            const int ix = a_ix0[i] + dx;
            const bool ok = ((unsigned)iy < (unsigned)pH) && ((unsigned)(ix + nx_px) < (unsigned)pW) && nx_cok;
            vo = ok ? vo : (int)0x80000000;      // out of range => hardware returns zeros
        }
        if constexpr (F16) {
            areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 0, 0));
        } else {
            // global -> LDS without a register round trip: this wave's 64 x 16 B = rows 32i + 8w .. +7 of the tile
            load_b128_to_lds(rsrc_a, lds + dst_buf * BUF + (RPP * i + 8 * wave_u) * LDS_LD, vo, 0);
        }
    };
    auto piece_b = [&](int i, int kc) {
        if constexpr (F16) {
            breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[i], ((kh * pKW + kw) * ncpt + cc) * 128, 0));
        } else {
            load_b128_to_lds(rsrc_b, lds + dst_buf * BUF + (BM + RPP * i + 8 * wave_u) * LDS_LD, b_voff[i], ((kh * pKW + kw) * ncpt + cc) * 128);
        }
    };
    auto piece_end = [&]() {                     // advance (kh, kw, cc) and the running offsets with selects
        // (0 / 1 arithmetic, no select chains: with `?:` on three levels the compiler kept this state in scratch memory)
        const int w = (kw + 1 == pKW) ? 1 : 0;
        const int hh = w & ((kh + 1 == pKH) ? 1 : 0);
        toff += step_kw + w * (step_kh - step_kw) + hh * (step_cc - step_kh);
        dx = (1 - w) * (dx + pdil);
        dy = (1 - hh) * (dy + w * pdil);
        kw = (1 - w) * (kw + 1);
        kh = (1 - hh) * (kh + w);
        cc += hh;
    };
    auto load_chunk = [&](int kc) {              // un-interleaved form (prologue)
        piece_begin();
#pragma unroll
        for (int i = 0; i < A_LD; ++i) piece_a(i);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) piece_b(i, kc);
        piece_end();
    };
    auto store_chunk = [&](int buf) {
        if constexpr (F16) {
            _Float16 *As = reinterpret_cast<_Float16 *>(lds + buf * BUF);
            _Float16 *Bs = As + BM * LDS_LD_H;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const f16x4 hv = {(_Float16)areg[i][0], (_Float16)areg[i][1], (_Float16)areg[i][2], (_Float16)areg[i][3]};
                *reinterpret_cast<f16x4 *>(As + (ld_row + RPP * i) * LDS_LD_H + ld_c) = hv;
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const f16x4 hv = {(_Float16)breg[i][0], (_Float16)breg[i][1], (_Float16)breg[i][2], (_Float16)breg[i][3]};
                *reinterpret_cast<f16x4 *>(Bs + (ld_row + RPP * i) * LDS_LD_H + ld_c) = hv;
            }
            return;
        }
        // f32 math: the data is already on its way into LDS; it has landed when vmcnt reaches 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    const int r = lane & 31;
    const int h = lane >> 5;
    // Block-uniform: this tile takes the fast epilogue.  Then the bias is not added there (64 VALU adds per
    // thread) but is what the accumulators start from -- a lane's 16 registers of one 32x32 tile all belong
    // to ONE output column, so it costs one scalar load per tile column block.
    const bool fast_blk = direct && (out_vec_ok(p) || p.out_f16) && !p.shuffle2x2 && p.act != ML_ACT_SIGMOID && (p.cout % 4 == 0);
    f32x16 acc[TM][TN];
    f32x16 accx[X3 ? TM : 1][X3 ? TN : 1];             // X3: the two cross terms, in units of 2^-11
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int col = n0 + wn * TN * 32 + ni * 32 + r;
        const float b0 = (fast_blk && p.bias && col < p.cout) ? p.bias[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[mi][ni][e] = b0;
                if constexpr (X3) accx[mi][ni][e] = 0.f;
            }
    }

    const int a_off = F16 ? (wm * TM * 32 + r) * LDS_LD_H + h * 8 : (wm * TM * 32 + r) * LDS_LD;
    const int b_off = F16 ? BM * LDS_LD_H + (wn * TN * 32 + r) * LDS_LD_H + h * 8 : BM * LDS_LD + (wn * TN * 32 + r) * LDS_LD;
    const int swz = (r >> 1) & 7;                         // f32 math: k-group kg of row r sits in slot kg ^ swz

    // NSTAGE = 3 (the 256-row X3 tile): a ring of three staging buffers, the prefetch runs TWO chunks ahead and a chunk
    // ends with a counted wait (the loads of the chunk issued one iteration earlier have landed; this iteration's are
    // still in flight) -- an X3 chunk is 24 MFMAs of 32 cycles, shorter than the L2 -> LDS latency.
    constexpr int NLOADS = A_LD + B_LD;              // LDS-direct loads per thread and chunk
    constexpr int AHEAD = NSTAGE - 1;
    if (kc_begin < kc_end) {
        load_chunk(kc_begin);
        if constexpr (NSTAGE == 3) {
            const bool two = kc_begin + 1 < kc_end;
            rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)pin, 0, two ? in_bytes : 0, 0x00020000);
            rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, two ? wgt_bytes : 0, 0x00020000);
            dst_buf = 1;
            load_chunk(two ? kc_begin + 1 : kc_begin);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
        } else {
            store_chunk(0);
        }
    }
    if constexpr (NSTAGE == 3) __builtin_amdgcn_s_barrier();
    else __syncthreads();

    if constexpr (X3 && NSTAGE == 3) {
        // ---- the 256-row X3 tile: a software pipeline over the 16-deep steps.  The MFMAs of one step run while the NEXT
        // step's fragments are read from LDS and its A fragment is split (VALU) -- within a wave, instruction by instruction:
        // two waves of a SIMD that reach a barrier together would otherwise split together and then queue for the matrix
        // pipe together (measured: the un-pipelined form gains nothing from the larger tile).  The chunk's barrier sits
        // BETWEEN its two steps: behind it the next chunk (requested a whole chunk earlier) is visible, so the second
        // step's MFMAs cover the first fragments of that chunk; the buffer the new requests overwrite was last read
        // before the previous barrier.
        static_assert(TM == 1, "one A fragment per wave");
        constexpr int NPA = (NLOADS + 1) / 2;                 // loads requested in the first half of a chunk
        const float neg_scale = -2048.f;
        f16x8 ahA, alA, ahB, alB, bhA[TN], bhB[TN], bl[TN];
        f32x4 rx0, rx1;                                       // the next step's A fragment as staged (fp32)
        unsigned hw[4], lw[4];
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        auto ld_a_raw = [&](const float *b, int ks) {
            const float *row = b + a_off;
            rx0 = *reinterpret_cast<const f32x4 *>(row + (((ks * 4 + 2 * h) ^ swz) * 4));
            rx1 = *reinterpret_cast<const f32x4 *>(row + (((ks * 4 + 2 * h + 1) ^ swz) * 4));
        };
        auto ld_bh = [&](const float *b, int ks, f16x8 (&bh)[TN]) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                bh[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(b + b_off + ni * 32 * LDS_LD + (((ks * 2 + h) ^ swz) * 4)));
        };
        auto ld_bl = [&](const float *b, int ks) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                bl[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(b + b_off + ni * 32 * LDS_LD + (((4 + ks * 2 + h) ^ swz) * 4)));
        };
        auto split_pair = [&](int q) {                        // elements 2q, 2q + 1 of the staged fragment: 4 VALU instructions
            const float xa = q < 2 ? rx0[2 * (q & 1)] : rx1[2 * (q & 1)], xb = q < 2 ? rx0[2 * (q & 1) + 1] : rx1[2 * (q & 1) + 1];
            split_hi_lo_pair(xa, xb, neg_scale, hw[q], lw[q]);
        };
        auto pack = [&](f16x8 &hi, f16x8 &lo) {
            const u32x4 H = {hw[0], hw[1], hw[2], hw[3]}, L = {lw[0], lw[1], lw[2], lw[3]};
            hi = __builtin_bit_cast(f16x8, H);
            lo = __builtin_bit_cast(f16x8, L);
        };
        auto piece = [&](int q, int kcn) {
            if (q < A_LD) piece_a(q);
            else if (q < A_LD + B_LD) piece_b(q - A_LD, kcn);
            else piece_end();
        };
        // one step: 12 MFMAs on (ahC, alC, bhC, bl); the next step (buffer nb, step nks) is fetched into (ahN, alN, bhN, bl)
        auto step = [&](const f16x8 &ahC, const f16x8 &alC, f16x8 (&bhC)[TN], const float *nb, int nks, f16x8 &ahN, f16x8 &alN,
                        f16x8 (&bhN)[TN], int q0, int q1, int kcn) {
            int q = q0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {                 // cross term 1: hi_a x lo_b
                accx[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahC, bl[ni], accx[0][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ni == 0) { ld_a_raw(nb, nks); ld_bh(nb, nks, bhN); }
                else if (q < q1) piece(q++, kcn);
                __builtin_amdgcn_sched_barrier(0);
            }
            ld_bl(nb, nks);                                   // (bl is dead: the next step's low halves of B)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {                 // cross term 2: lo_a x hi_b; the next A fragment is split beside it
                accx[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alC, bhC[ni], accx[0][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                split_pair(ni);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {                 // hi_a x hi_b
                acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahC, bhC[ni], acc[0][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < q1) piece(q++, kcn);
                __builtin_amdgcn_sched_barrier(0);
            }
            pack(ahN, alN);
        };
        static_assert(TN == 4, "split_pair(ni) covers the 4 pairs of a fragment");
        if (kc_begin < kc_end) {                              // fragments of (first chunk, step 0): exposed once per tile
            ld_a_raw(lds, 0);
            ld_bh(lds, 0, bhA);
            ld_bl(lds, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) split_pair(q);
            pack(ahA, alA);
        }
        int buf = 0;
        for (int kc = kc_begin; kc < kc_end; ++kc) {
            const bool more = kc + 2 < kc_end;
            const int kc_next = more ? kc + 2 : kc;
            rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)pin, 0, more ? in_bytes : 0, 0x00020000);
            rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, more ? wgt_bytes : 0, 0x00020000);
            const float *base = lds + buf * BUF;
            const int nbuf = buf == 2 ? 0 : buf + 1;
            dst_buf = buf == 0 ? 2 : buf - 1;
            piece_begin();
            __builtin_amdgcn_sched_barrier(0);
            step(ahA, alA, bhA, base, 1, ahB, alB, bhB, 0, NPA, kc_next);
            // the chunk after this one has landed (requested during the previous iteration; only this iteration's NPA
            // requests may still be in flight) -- and every wave is past its last read of the buffer being refilled
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA) : "memory");
            __builtin_amdgcn_s_barrier();
            step(ahB, alB, bhB, lds + nbuf * BUF, 0, ahA, alA, bhA, NPA, NLOADS + 1, kc_next);
            buf = nbuf;
        }
    } else {
    int buf = 0;
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const bool more = kc + AHEAD < kc_end;
        // Unconditional prefetch, no branch: on the last iteration(s) it goes through empty resources (no traffic).
        const int kc_next = more ? kc + AHEAD : kc;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)pin, 0, more ? in_bytes : 0, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, more ? wgt_bytes : 0, 0x00020000);
        const float *base = lds + buf * BUF;
        dst_buf = NSTAGE == 3 ? (buf == 0 ? 2 : buf - 1) : (buf ^ 1);
        piece_begin();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NPIECE = A_LD + B_LD + 1;              // + piece_end
        constexpr int NMFMA = 16 * TM * TN;
        constexpr int GAP = NMFMA >= 4 * NPIECE ? 2 : 1;     // pieces ride early so the loads have the rest of the block to land
        int placed = 0;
        if constexpr (F16) {
            // lane (r, h) holds A[row r][k = 16 ks + 8 h + j], j = 0..7 (one ds_read_b128), B likewise
            const _Float16 *baseh = reinterpret_cast<const _Float16 *>(base);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[TM], b[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    a[mi] = *reinterpret_cast<const f16x8 *>(baseh + a_off + mi * 32 * LDS_LD_H + ks * 16);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    b[ni] = *reinterpret_cast<const f16x8 *>(baseh + b_off + ni * 32 * LDS_LD_H + ks * 16);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                        if (placed < NPIECE) {               // one prefetch piece per MFMA
                            __builtin_amdgcn_sched_barrier(0);
                            if (placed < A_LD) piece_a(placed);
                            else if (placed < A_LD + B_LD) piece_b(placed - A_LD, kc_next);
                            else piece_end();
                            ++placed;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
#pragma unroll
            for (int q = 0; q < NPIECE; ++q)                 // what did not fit between the MFMAs
                if (q >= placed) {
                    if (q < A_LD) piece_a(q);
                    else if (q < A_LD + B_LD) piece_b(q - A_LD, kc_next);
                    else piece_end();
                }
        } else if constexpr (X3) {
            // lane (r, h) feeds A[row r][k = 16 ks + 8 h + j], j = 0..7: fp32 k-groups 4 ks + 2 h and + 1 of its row, split
            // here; the B row of a chunk is 32 hi halves (k-groups 0..3) then 32 scaled lo halves (k-groups 4..7)
            const float neg_scale = -2048.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const float *row = base + a_off + mi * 32 * LDS_LD;
                    split_hi_lo(*reinterpret_cast<const f32x4 *>(row + (((ks * 4 + 2 * h) ^ swz) * 4)),
                                *reinterpret_cast<const f32x4 *>(row + (((ks * 4 + 2 * h + 1) ^ swz) * 4)), neg_scale, ah[mi], al[mi]);
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const float *row = base + b_off + ni * 32 * LDS_LD;
                    bh[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(row + (((ks * 2 + h) ^ swz) * 4)));
                    bl[ni] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(row + (((4 + ks * 2 + h) ^ swz) * 4)));
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) {
                            if (t == 0) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                            else if (t == 1) accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], accx[mi][ni], 0, 0, 0);
                            else accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], accx[mi][ni], 0, 0, 0);
                            if (placed < NPIECE) {               // one prefetch piece per MFMA
                                __builtin_amdgcn_sched_barrier(0);
                                if (placed < A_LD) piece_a(placed);
                                else if (placed < A_LD + B_LD) piece_b(placed - A_LD, kc_next);
                                else piece_end();
                                ++placed;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
            }
#pragma unroll
            for (int q = 0; q < NPIECE; ++q)                 // what did not fit between the MFMAs (narrow tiles)
                if (q >= placed) {
                    if (q < A_LD) piece_a(q);
                    else if (q < A_LD + B_LD) piece_b(q - A_LD, kc_next);
                    else piece_end();
                }
        } else if constexpr (HS) {
            // lane (r, h) feeds A[row r][k = 16 ks + 8 h + j], j = 0..7: the 16-byte k-group 2 ks + h of its row
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    a[mi] = *reinterpret_cast<const f32x4 *>(base + a_off + mi * 32 * LDS_LD + (((ks * 2 + h) ^ swz) * 4));
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    b[ni] = *reinterpret_cast<const f32x4 *>(base + b_off + ni * 32 * LDS_LD + (((ks * 2 + h) ^ swz) * 4));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[mi]),
                                                                             __builtin_bit_cast(f16x8, b[ni]), acc[mi][ni], 0, 0, 0);
                        if (placed < NPIECE) {               // one prefetch piece per MFMA
                            __builtin_amdgcn_sched_barrier(0);
                            if (placed < A_LD) piece_a(placed);
                            else if (placed < A_LD + B_LD) piece_b(placed - A_LD, kc_next);
                            else piece_end();
                            ++placed;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
#pragma unroll
            for (int q = 0; q < NPIECE; ++q)                 // what did not fit between the MFMAs (narrow tiles)
                if (q >= placed) {
                    if (q < A_LD) piece_a(q);
                    else if (q < A_LD + B_LD) piece_b(q - A_LD, kc_next);
                    else piece_end();
                }
        } else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[mi] = *reinterpret_cast<const f32x4 *>(base + a_off + mi * 32 * LDS_LD + (((ks * 2 + h) ^ swz) * 4));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                b[ni] = *reinterpret_cast<const f32x4 *>(base + b_off + ni * 32 * LDS_LD + (((ks * 2 + h) ^ swz) * 4));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
                        const int idx = ((ks * 4 + j) * TM + mi) * TN + ni;      // compile-time after unrolling
                        if (idx % GAP == GAP - 1 && placed < NPIECE) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (placed < A_LD) piece_a(placed);
                            else if (placed < A_LD + B_LD) piece_b(placed - A_LD, kc_next);
                            else piece_end();
                            ++placed;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
        }
        if constexpr (F16) {
            if (more) store_chunk(buf ^ 1);
        } else if constexpr (NSTAGE == 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");      // the NEXT chunk's rows have landed
        } else {
            store_chunk(buf ^ 1);        // always: the (dropped) last prefetch must have landed before the epilogue re-uses the LDS
        }
        // (__syncthreads() is a fence too: the compiler puts s_waitcnt vmcnt(0) in front of it, which would wait for the
        // chunk that was only just requested; the bare barrier is enough -- every wave's reads of `buf` were consumed by
        // its MFMAs before it gets here, and the next writes into `buf` are issued after the barrier)
        if constexpr (NSTAGE == 3) __builtin_amdgcn_s_barrier();
        else __syncthreads();
        buf = NSTAGE == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
    }
    }
    if constexpr (NSTAGE == 3) {         // (the dropped last prefetches: nothing may still be writing when the epilogue re-uses the LDS)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Transpose the BM x BN tile through LDS (all staging reads finished at the barrier above).
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * TM * 32 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int col = wn * TN * 32 + ni * 32 + r;
                float v = acc[mi][ni][e];
                if constexpr (X3) v = fmaf(accx[mi][ni][e], 0x1p-11f, v);
                lds[row * C_LD + col] = v;
            }
    __syncthreads();

    if (!direct) {
        // raw partial tile -> slab[slice][m][n] (n_pad pitch); bias/act happen in the reduce kernel
        float *slab = P.slab + (size_t)slice * (size_t)(P.MB * BM) * p.n_pad;
#pragma unroll 4
        for (int rr = r0; rr < BM; rr += ROWS_PER_PASS) {
            const int m = m0 + rr;
            if (m >= M) break;
            *reinterpret_cast<f32x4 *>(slab + (size_t)m * p.n_pad + n) = *reinterpret_cast<const f32x4 *>(lds + rr * C_LD + c4);
        }
        return;
    }
    if (n >= p.cout) return;
    if (fast_blk) {
        // fast path (every backbone / tower conv): dense or channel-sliced NHWC destination, float4 lanes,
        // ReLU / ReLU6 / none as a branch-free clamp.  Row pointers advance by a constant stride.
        // ML_ACT_NONE stores the sum as it is (no clamp: NaN / Inf stay what they are); block-uniform branch
        const bool clampv = p.act != ML_ACT_NONE;
        const float lo = 0.f;
        const float hi = (p.act == ML_ACT_RELU6) ? 6.f : 3.402823466e38f;
        const bool late_res = p.residual && !pre_res;
        const size_t cs = p.out_cstride;
        size_t off;
        if (p.out_bstride) {
            // per-image strided destination: rows of one tile may straddle images, keep the division
            off = 0;
        } else {
            off = (size_t)(m0 + r0) * cs + p.out_coff + n;
        }
        // all LDS reads first (the accumulator registers are free now), then the stores: a serial
        // read -> wait -> store chain cost 5300 cycles per tile
        f32x4 tile_v[E_ROWS];
#pragma unroll
        for (int i = 0; i < E_ROWS; ++i)
            tile_v[i] = *reinterpret_cast<const f32x4 *>(lds + (r0 + i * ROWS_PER_PASS) * C_LD + c4);
        // (the bias is already in the accumulators; the clamp is one v_med3_f32 per element)
        const bool full = (m0 + BM <= M);            // block-uniform: no per-row range checks
        if (p.out_f16) {
            // fp16-storage body behind this conv (the stem): same tile ownership, 4 halves (8 bytes) per lane and row
            _Float16 *oh = reinterpret_cast<_Float16 *>(p.out);
            [[maybe_unused]] double gsh = 0.0, gqh = 0.0;
#pragma unroll
            for (int i = 0; i < E_ROWS; ++i) {
                const int m = m0 + r0 + i * ROWS_PER_PASS;
                if (m < M) {
                    f32x4 v = tile_v[i];
                    if (clampv) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
                    }
                    const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4 *>(oh + off + (size_t)i * ROWS_PER_PASS * cs) = hv;
                    if constexpr (GNS && NT == 256) {
                        // the GroupNorm behind a half head conv normalises the STORED (rounded) values: sum those
                        const float h0 = (float)hv[0], h1 = (float)hv[1], h2 = (float)hv[2], h3 = (float)hv[3];
                        gsh += (double)((h0 + h1) + (h2 + h3));
                        gqh += (double)fmaf(h3, h3, fmaf(h2, h2, fmaf(h1, h1, h0 * h0)));
                    }
                }
            }
            if constexpr (GNS && NT == 256) {
                if (p.gn_partials) {                           // (block-uniform; whole tiles only: launch_multi checks M % BM)
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        gsh += __shfl_down(gsh, o, 64);
                        gqh += __shfl_down(gqh, o, 64);
                    }
                    if (lane == 0) {                           // one pair per WAVE, as in the fp32 form below
                        p.gn_partials[2 * ((size_t)mt * 4 + wave)] = gsh;
                        p.gn_partials[2 * ((size_t)mt * 4 + wave) + 1] = gqh;
                    }
                }
            }
        } else if (full && !late_res && !p.out_bstride) {
            float *op = p.out + off;
            const size_t step = (size_t)ROWS_PER_PASS * cs;
            constexpr int HALVES = BM / 128;                 // 128-row tiles the partial sums are kept for
            double gs[HALVES] = {}, gq[HALVES] = {};
#pragma unroll
            for (int i = 0; i < E_ROWS; ++i) {
                f32x4 v = tile_v[i];
                if (pre_res) v += res[i];
                if (clampv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
                }
                *reinterpret_cast<f32x4 *>(op + (size_t)i * step) = v;
                if constexpr (GNS) {
                    const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
                    const float q4 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
                    gs[i * HALVES / E_ROWS] += (double)s4;   // (rows r0 + i * ROWS_PER_PASS, r0 < ROWS_PER_PASS)
                    gq[i * HALVES / E_ROWS] += (double)q4;
                }
            }
            if constexpr (GNS) {
                if (p.gn_partials) {                           // (block-uniform)
#pragma unroll
                    for (int hh = 0; hh < HALVES; ++hh)
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            gs[hh] += __shfl_down(gs[hh], o, 64);
                            gq[hh] += __shfl_down(gq[hh], o, 64);
                        }
                    if constexpr (NT == 256) {
                        // one pair per WAVE (a tile's rows 2w, 2w + 1 mod 8): no cross-wave step, no barrier
                        if (lane == 0) {
                            p.gn_partials[2 * ((size_t)mt * 4 + wave)] = gs[0];
                            p.gn_partials[2 * ((size_t)mt * 4 + wave) + 1] = gq[0];
                        }
                    } else {
                        // 8 waves on a 256-row tile: waves w and w + 4 hold the rows 2w, 2w + 1 mod 8 of BOTH 128-row
                        // halves; the pair of a (half, slot) is the sum of the two, through LDS behind the transposed tile
                        double *scr = reinterpret_cast<double *>(lds + BM * C_LD);
                        if (lane == 0) {
#pragma unroll
                            for (int hh = 0; hh < HALVES; ++hh) {
                                scr[(wave * HALVES + hh) * 2] = gs[hh];
                                scr[(wave * HALVES + hh) * 2 + 1] = gq[hh];
                            }
                        }
                        __syncthreads();
                        if (wave < 4 && lane == 0) {
#pragma unroll
                            for (int hh = 0; hh < HALVES; ++hh) {
                                const size_t slot = 2 * (((size_t)mt * HALVES + hh) * 4 + wave);
                                p.gn_partials[slot] = scr[(wave * HALVES + hh) * 2] + scr[((wave + 4) * HALVES + hh) * 2];
                                p.gn_partials[slot + 1] = scr[(wave * HALVES + hh) * 2 + 1] + scr[((wave + 4) * HALVES + hh) * 2 + 1];
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < E_ROWS; ++i) {
                const int m = m0 + r0 + i * ROWS_PER_PASS;
                if (m < M) {
                    f32x4 v = tile_v[i];
                    if (pre_res) v += res[i];
                    if (late_res) v += *reinterpret_cast<const f32x4 *>(p.residual + (size_t)m * p.res_cstride + p.res_coff + n);
                    if (clampv) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
                    }
                    size_t o = off + (size_t)i * ROWS_PER_PASS * cs;
                    if (p.out_bstride) {
                        const int b = fast_div(m, P.div_howo);
                        o = (size_t)b * (size_t)p.out_bstride + (size_t)(m - b * HoWo) * cs + p.out_coff + n;
                    }
                    *reinterpret_cast<f32x4 *>(p.out + o) = v;
                }
            }
        }
    } else {
        // generic path (sigmoid heads, Conv2DTranspose pixel shuffle, non-multiple-of-4 channel counts):
        // deliberately NOT unrolled -- it is the rare case and used to bloat the epilogue to 13.8 k instructions
#pragma unroll 1
        for (int i = 0; i < E_ROWS; ++i) {
            const int rr = r0 + i * ROWS_PER_PASS;
            const int m = m0 + rr;
            if (m >= M) break;
            f32x4 v = *reinterpret_cast<const f32x4 *>(lds + rr * C_LD + c4);
            store_out4(p, HoWo, m, n, v, vec_ok, true);
        }
    }
}

// out = act(sum_s slab[s] + bias + residual), slices summed in index order (deterministic).
// One launch reduces every split problem of a multi-problem conv launch.
struct ReduceArgs {
    int n;
    int start[MAXP + 1];      // prefix sum of reduce blocks per problem
    int m_pad[MAXP];
    Problem p[MAXP];
};

__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const ReduceArgs args) {
    int pi = 0;
    while (pi + 1 < args.n && (int)blockIdx.x >= args.start[pi + 1]) ++pi;
    const Problem &P = args.p[pi];
    const ml_conv2d_desc &p = P.d;
    const float *__restrict__ slab = P.slab;
    const int splits = P.splits, M = P.M, m_pad = args.m_pad[pi];
    const int V = p.n_pad / 4;
    const long long idx = (long long)(blockIdx.x - args.start[pi]) * 256 + threadIdx.x;
    if (idx >= (long long)M * V) return;
    const int m = (int)(idx / V);
    const int n = (int)(idx % V) * 4;
    if (n >= p.cout) return;
    if (p.live && (m / (p.Ho * p.Wo)) % p.live_period >= max(1, *p.live)) return;      // a row of an image that does not exist
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s)
        v += *reinterpret_cast<const f32x4 *>(slab + ((size_t)s * m_pad + m) * p.n_pad + n);
    const bool vec_ok = out_vec_ok(p) && (n + 4 <= p.cout);
    store_out4(p, p.Ho * p.Wo, m, n, v, vec_ok);
}

int pick_tile(int cout, int tile) {
    if (tile == 4 || tile == 5) return 1;   // the pipelined 1x1 kernels (4: 128x128 tiles; 5: the half 256x256 one) pack like tile 1
    if (tile >= 1 && tile <= 3) return tile;
    if (cout <= 32) return 3;
    if (cout <= 64) return 2;
    if (cout <= 96) return 3;
    return 1;
}

// generic = the checks of the implicit-GEMM kernel in this file (the persistent 1x1 kernel has its own eligibility test)
int validate(const ml_conv2d_desc &d, bool generic = true) {
    ML_REQUIRE(d.in && d.wgt && d.out, "conv2d: null tensor pointer");
    ML_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0, "conv2d: bad spatial dims");
    ML_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0, "conv2d: bad kernel geometry");
    const bool hs = d.math == ML_MATH_F16S;
    const int es = hs ? 2 : 4, q16 = 16 / es, kc = hs ? 64 : 32;          // element bytes, elements per 16 bytes / per chunk
    ML_REQUIRE(d.span > 0 && d.span % q16 == 0, "conv2d: span %d must be a positive multiple of %d", d.span, q16);
    ML_REQUIRE(d.span_pad == (d.span + kc - 1) / kc * kc, "conv2d: span_pad %d != span %d rounded up to %d", d.span_pad, d.span, kc);
    ML_REQUIRE(d.in_cstride % q16 == 0 && d.in_coff % q16 == 0 && d.group_cin_step % q16 == 0,
               "conv2d: input channel stride/offset must be multiples of %d (16-byte loads)", q16);
    if (hs)
        ML_REQUIRE(d.cpp_shift == 30 && d.group_cin_step == 0,
                   "conv2d: fp16 storage takes no image (row-span) input and no grouped windows");
    ML_REQUIRE(ml_aligned16(d.in) && ml_aligned16(d.wgt), "conv2d: in/wgt must be 16-byte aligned");
    ML_REQUIRE(d.cpp_shift >= 0 && d.cpp_shift <= 30, "conv2d: bad cpp_shift");
    ML_REQUIRE(d.cout > 0 && d.out_cstride > 0 && d.out_coff >= 0 && d.out_bstride >= 0, "conv2d: bad output channels");
    ML_REQUIRE((long long)d.B * d.H * d.W < (1ll << 31) / 2, "conv2d: too many input pixels for int32 indexing");
    ML_REQUIRE((long long)d.B * d.Ho * d.Wo < (1ll << 31) - 256, "conv2d: too many output pixels");
    ML_REQUIRE((long long)d.H * d.W * d.in_cstride * es < (1ll << 31),
               "conv2d: ONE image of the input must be < 2 GiB (32-bit buffer offsets inside an image group)");
    ML_REQUIRE((long long)d.n_pad * d.KH * d.KW * d.span_pad * es < (1ll << 31), "conv2d: weight tensor must be < 2 GiB");
    if (d.shuffle2x2) {
        ML_REQUIRE(d.cout % 4 == 0 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.Ho == d.H && d.Wo == d.W,
                   "conv2d: shuffle2x2 needs a 1x1 stride-1 problem with cout = 4*Cout");
        ML_REQUIRE(d.out_coff + d.cout / 4 <= d.out_cstride, "conv2d: output slice exceeds buffer channels");
        ML_REQUIRE(d.residual == nullptr && d.out_bstride == 0, "conv2d: shuffle2x2 takes no residual / batch stride");
    } else {
        ML_REQUIRE(d.out_coff + d.cout <= d.out_cstride, "conv2d: output slice exceeds buffer channels");
    }
    if (d.residual) ML_REQUIRE(d.res_coff + d.cout <= d.res_cstride, "conv2d: residual slice exceeds buffer");
    ML_REQUIRE(d.out_f16 == 0 || d.out_f16 == 1, "conv2d: out_f16 must be 0 or 1");
    if (d.live)
        ML_REQUIRE(d.live_period >= 1 && d.B % d.live_period == 0 && (long long)d.Ho * d.Wo * d.live_period >= 128,
                   "conv2d: `live` needs live_period >= 1 dividing B and at least one 128-row tile per period");
    if (d.out_f16 && generic)
        ML_REQUIRE((d.math == ML_MATH_F16 || hs) && !d.residual && !d.shuffle2x2 && d.out_bstride == 0 && d.act != ML_ACT_SIGMOID &&
                       d.cout % 4 == 0 && d.out_cstride % 4 == 0 && d.out_coff % 4 == 0 && (((uintptr_t)d.out) & 7) == 0,
                   "conv2d: out_f16 needs the fp16 MFMA mode, the dense fast epilogue (no residual / shuffle / batch "
                   "stride / sigmoid) and 4-channel alignment");
    if (d.cpp_shift == 30) {
        ML_REQUIRE(d.in_coff + d.span <= d.in_cstride || d.group_cin_step > 0,
                   "conv2d: input slice exceeds buffer channels");
    }
    const int t = pick_tile(d.cout, d.tile);
    const int bn = t == 1 ? 128 : (t == 2 ? 64 : 32);
    ML_REQUIRE(d.n_pad >= d.cout && d.n_pad % bn == 0, "conv2d: n_pad %d must be a multiple of the N tile %d", d.n_pad, bn);
    if (d.group_cin_step) {
        ML_REQUIRE(t == 3, "conv2d: grouped mode needs the 32-wide N tile");
        ML_REQUIRE(d.in_coff + (d.n_pad / 32 - 1) * d.group_cin_step + d.span <= d.in_cstride,
                   "conv2d: grouped input slices exceed buffer channels");
    }
    return ML_OK;
}

// where the pipelined 1x1 kernel is used by default (per-launch A/B inside the model, gpurun_out/launches_r02a vs r02d:
// x1.2-1.3 at K = 64-128, x1.05-1.15 at K = 256-512, nothing at K >= 1024; it loses where there are few rows -- no
// split-K: 64 panels of K = 2048 took 131 us against 50).  The rule looks at ONE image's pixel count, never at the
// batch: an image's results must not depend on the shard it is computed in (tests/test_gpu_model.py).
bool pipe_preferred(const ml_conv2d_desc &d) {
    if (!(d.KH == 1 && d.KW == 1 && d.span <= 512 && (long long)d.H * d.W >= 4096)) return false;
    // ML_MATH_F32X3 (round 4, gpurun_out/r04x_ab_x3*.txt): x1.3-1.4 where the launch is HBM-bound (K <= 256, or a residual
    // to read: 5.2-5.5 TB/s against 3.4-3.9), x1.2 at 512 -> 1024 + residual; 512 -> 256 without one is compute-bound in
    // that mode and 2-8 % faster on the generic 256-row tile
    if (d.math == ML_MATH_F32X3) return d.span <= 256 || d.residual != nullptr;
    return true;
}

// split-K heuristic: few tiles and a long K => slice K so that ~2 blocks per CU are in flight
int choose_splits(long long tiles, int chunks) {
    if (tiles >= 192 || chunks < 16) return 1;
    long long want = (512 + tiles - 1) / tiles;
    int max_by_k = chunks / 8;           // keep >= 8 chunks per slice
    int s = (int)(want < max_by_k ? want : max_by_k);
    if (s > 64) s = 64;
    return s < 2 ? 1 : s;
}

// host copy of the kernel's out_vec_ok(): the fast (float4, LDS-transposed) epilogue -- the only one that writes gn_partials
static bool host_out_vec_ok(const ml_conv2d_desc &p) {
    const int co = p.shuffle2x2 ? (p.cout >> 2) : p.cout;
    bool ok = (co % 4 == 0) && (p.out_cstride % 4 == 0) && (p.out_coff % 4 == 0) && (p.out_bstride % 4 == 0) && ml_aligned16(p.out);
    if (p.bias) ok = ok && ml_aligned16(p.bias);
    if (p.residual) ok = ok && (p.res_cstride % 4 == 0) && (p.res_coff % 4 == 0) && ml_aligned16(p.residual);
    return ok;
}

// The K-slice count of every problem of one launch on BM x BN tiles with KC-deep chunks -- THE place the decision is taken
// (launch_multi and the reporting entry ml_conv2d_launch_splits both call it).  split_tiles >= 0: the tile count the
// decision is taken on (see narrow_tile_for_small_launch), else this launch's own.  Split-K looks at the whole launch:
// five pyramid levels of one image are 171 tiles together -- a third of the chip -- and each tile then walks all 36
// chunks alone (92 us); slicing K fills the other CUs.  Fixed-capacity RoI batches (`live`): an image's RoIs are spread
// over the launch's RoI levels, so about 1 / levels of the nominal tiles are live -- the decision is taken on that
// estimate (the host does not know the counts).
static void plan_splits(const ml_conv2d_desc *descs, int n, int BM, int BN, int KC, bool have_ws, long long ws_bytes,
                        long long split_tiles, int *splits_out, int *cps_out = nullptr) {
    long long launch_tiles = 0, ws_off = 0;
    int n_live = 0;
    for (int i = 0; i < n; ++i) n_live += descs[i].live != nullptr;
    for (int i = 0; i < n; ++i) {
        const long long M = (long long)descs[i].B * descs[i].Ho * descs[i].Wo;
        const long long t = ((M + BM - 1) / BM) * (descs[i].n_pad / BN);
        launch_tiles += descs[i].live ? (t + n_live - 1) / n_live : t;
    }
    if (split_tiles >= 0) launch_tiles = split_tiles;
    for (int i = 0; i < n; ++i) {
        const ml_conv2d_desc &d = descs[i];
        const long long M = (long long)d.B * d.Ho * d.Wo;
        const long long MB = (M + BM - 1) / BM;
        const int chunks = d.KH * d.KW * (d.span_pad / KC);
        int splits = have_ws ? choose_splits(launch_tiles, chunks) : 1;     // (the reduce kernel stores half too)
        const long long slab_bytes = (long long)splits * MB * BM * d.n_pad * 4;
        if (splits > 1 && ws_off + slab_bytes > ws_bytes) splits = 1;
        const int cps = (chunks + splits - 1) / splits;
        splits = (chunks + cps - 1) / cps;           // drop empty trailing slices
        if (splits > 1) ws_off += (slab_bytes + 255) / 256 * 256;
        splits_out[i] = splits;
        if (cps_out) cps_out[i] = cps;
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int MATH, bool GNS = false, int NSTAGE = 2>
int launch_multi(const ml_conv2d_desc *descs, int n, void *workspace, long long ws_bytes, hipStream_t s,
                 long long split_tiles = -1) {
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    constexpr bool F16 = MATH == ML_MATH_F16;
    constexpr int ES = MATH == ML_MATH_F16S ? 2 : 4, KC = MATH == ML_MATH_F16S ? 64 : 32;
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int STAGE_BYTES = F16 ? 2 * (BM + BN) * LDS_LD_H * 2 : NSTAGE * (BM + BN) * LDS_LD * 4;
    constexpr int EPI_BYTES = BM * (BN + 4) * 4 + (GNS && NT == 512 ? 256 : 0);   // the epilogue's transposed tile re-uses the staging LDS
                                                                                  // (+ 8 waves x 2 halves x 2 doubles of GroupNorm partial sums)
    constexpr int LDS_BYTES0 = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    constexpr int LDS_BYTES = LDS_BYTES0;
    auto kern = conv_mfma_kernel<WAVES_M, WAVES_N, TM, TN, MATH, GNS, NSTAGE>;
    static std::atomic<unsigned long long> lds_ok{0};      // per kernel instantiation, one bit per device
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, lds_ok, "conv2d")) return rc;
    MultiArgs args;
    args.n = n;
    long long start = 0, ws_off = 0;
    int planned[MAXP], planned_cps[MAXP];
    plan_splits(descs, n, BM, BN, KC, workspace != nullptr, ws_bytes, split_tiles, planned, planned_cps);
    for (int i = 0; i < n; ++i) {
        const ml_conv2d_desc &d = descs[i];
        Problem &P = args.p[i];
        P.d = d;
        const long long M = (long long)d.B * d.Ho * d.Wo;
        P.M = (int)M;
        P.MB = (int)((M + BM - 1) / BM);
        P.NB = d.n_pad / BN;
        P.ncpt = d.span_pad / KC;
        P.ktot = d.KH * d.KW * d.span_pad;
        P.div_howo = make_fastdiv((unsigned)(d.Ho * d.Wo));
        P.div_wo = make_fastdiv((unsigned)d.Wo);
        P.in_bytes = (unsigned)((long long)d.B * d.H * d.W * d.in_cstride * ES);
        P.wgt_bytes = (unsigned)((long long)d.n_pad * P.ktot * ES);
        const int chunks = d.KH * d.KW * P.ncpt;
        P.blocks_per_split = (P.MB + 7) / 8 * 8 * P.NB;
        const int splits = planned[i];
        const long long slab_bytes = (long long)splits * P.MB * BM * d.n_pad * 4;
        if (d.gn_partials)
            ML_REQUIRE(GNS && splits == 1 && BN == 128 && d.cout == 128 && d.n_pad == 128 && M % BM == 0 && !d.residual &&
                           !d.out_bstride && (!d.out_f16 || MATH == ML_MATH_F16S) && !d.shuffle2x2 && d.act != ML_ACT_SIGMOID &&
                           !d.live && (d.out_f16 || host_out_vec_ok(d)),
                       "conv2d: gn_partials needs a launch that is neither narrowed nor split along K (ml_conv2d_gn_min_launch_tiles() "
                       "tiles of 128 x 128), cout = 128, whole 128-row tiles, no residual, and a dense fp32 destination on the "
                       "vector epilogue (out / bias 16-byte aligned, out_cstride and out_coff multiples of 4)");
        P.splits = splits;
        P.cps = planned_cps[i];
        P.slab = nullptr;
        if (P.splits > 1) {
            P.slab = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ws_off);
            ws_off += (slab_bytes + 255) / 256 * 256;
        } else {
            P.cps = chunks;
        }
        args.start[i] = (int)start;
        start += (long long)P.blocks_per_split * P.splits;
        ML_REQUIRE(start < (1ll << 31), "conv2d: grid too large");
    }
    args.start[n] = (int)start;
    hipLaunchKernelGGL(kern, dim3((unsigned)start), dim3(NT), LDS_BYTES, s, args);
    ML_CHECK_LAUNCH("conv2d");
    ReduceArgs ra;
    ra.n = 0;
    long long rblocks = 0;
    for (int i = 0; i < n; ++i) {
        const Problem &P = args.p[i];
        if (P.splits > 1) {
            const long long work = (long long)P.M * (P.d.n_pad / 4);
            ra.p[ra.n] = P;
            ra.m_pad[ra.n] = P.MB * BM;
            ra.start[ra.n] = (int)rblocks;
            rblocks += (work + 255) / 256;
            ++ra.n;
        }
    }
    if (ra.n > 0) {
        ra.start[ra.n] = (int)rblocks;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, ra);
        ML_CHECK_LAUNCH("conv2d split-K reduce");
    }
    return ML_OK;
}

}  // namespace

int ml_conv1x1_pipe_try(const ml_conv2d_desc &d, hipStream_t s, int *eligible);     // conv1x1_pipe.hip
int ml_conv1x1_h256_try(const ml_conv2d_desc &d, hipStream_t s, int *took);          // conv1x1_h256.hip
int ml_conv1x1_h256_eligible(const ml_conv2d_desc &d);
// half tensors: where the 256 x 256-tile kernel is preferred over the 128 x 128 pipelined one (per-launch A/B,
// gpurun_out/r03_h256_ab.txt): K >= 512, or K >= 256 with at least two 256-wide N tiles
static bool h256_preferred(const ml_conv2d_desc &d) {
    return d.tile == 5 || (d.tile == 0 && (d.span >= 512 || (d.span >= 256 && d.cout >= 512)));
}

extern "C" int ml_conv2d_ntile(int32_t cout, int32_t tile) {
    const int t = pick_tile(cout, tile);
    return t == 1 ? 128 : (t == 2 ? 64 : 32);
}

extern "C" int64_t ml_conv2d_workspace_bytes(void) { return 512ll << 20; }   // (split-K slabs of fixed-capacity RoI batches are sized for every slot)

int ml_conv1x1_pipe_eligible(const ml_conv2d_desc &d);                               // conv1x1_pipe.hip
// fp16 storage: the persistent kernel takes every 1x1 problem it can run, except those the generic kernel would cut
// along K (few tiles, long K: the laterals of the coarse pyramid levels, the image-pooling branch) -- it has no split-K.
// A residual (the ResNeXt conv3 of every block) is only implemented there.
static bool pipe_preferred_half(const ml_conv2d_desc &d) {
    if (d.residual || d.tile == 4) return true;
    const long long M = (long long)d.B * d.Ho * d.Wo;
    const long long tiles = ((M + 127) / 128) * ((d.cout + 127) / 128);
    return tiles >= 192 || d.span / 64 < 16;
}

extern "C" int ml_conv2d_uses_pipe(const ml_conv2d_desc *d) {
    if (!d) return 0;
    if (d->math == ML_MATH_F16S && h256_preferred(*d) && ml_conv1x1_h256_eligible(*d)) return 2;
    if (d->math == ML_MATH_F16S) return pipe_preferred_half(*d) && ml_conv1x1_pipe_eligible(*d);
    if (d->tile == 4) return ml_conv1x1_pipe_eligible(*d);
    return d->tile == 0 && pipe_preferred(*d) && ml_conv1x1_pipe_eligible(*d);
}

// The generic kernel addresses its input through 32-bit buffer offsets: a problem whose activation is >= 2 GiB (e.g. 32
// images of 256x256x256 fp32, the largest batch the reference's MoldBatch takes, engine/layers/misc.py:273-284) is
// cut into groups of whole images, each its own problem of the same launch.  -> number of problems written, or < 0.
static int split_by_image_groups(const ml_conv2d_desc *descs, int n, ml_conv2d_desc *out, int cap) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const ml_conv2d_desc &d = descs[i];
        const int es_in = d.math == ML_MATH_F16S ? 2 : 4, es_out = d.out_f16 ? 2 : 4;
        const long long img_bytes = (long long)d.H * d.W * d.in_cstride * es_in;
        const long long limit = (1ll << 31) - 16;
        long long per = img_bytes > 0 ? limit / img_bytes : d.B;
        if (per < 1) per = 1;
        if (per >= d.B) {
            if (m >= cap) return -1;
            out[m++] = d;
            continue;
        }
        if (d.live) return -2;          // (image groups would renumber the images a `live` period counts)
        // gn_partials is indexed by the problem's 128-row tile: a group starts at tile b0 * Ho * Wo / 128 of the whole
        // tensor, which must be a whole number for every group
        if (d.gn_partials && (per * d.Ho * d.Wo) % 128) return -3;
        const long long in_img = (long long)d.H * d.W * d.in_cstride;
        const long long out_pix = d.shuffle2x2 ? 4ll * d.Ho * d.Wo : (long long)d.Ho * d.Wo;
        const long long out_img = d.out_bstride ? d.out_bstride : out_pix * d.out_cstride;
        const long long res_img = (long long)d.Ho * d.Wo * d.res_cstride;
        for (long long b0 = 0; b0 < d.B; b0 += per) {
            if (m >= cap) return -1;
            ml_conv2d_desc g = d;
            g.B = (int)(d.B - b0 < per ? d.B - b0 : per);
            g.in = reinterpret_cast<const float *>(reinterpret_cast<const char *>(d.in) + b0 * in_img * es_in);
            g.out = reinterpret_cast<float *>(reinterpret_cast<char *>(d.out) + b0 * out_img * es_out);
            if (d.residual) g.residual = d.residual + b0 * res_img;
            if (d.gn_partials) g.gn_partials = d.gn_partials + (b0 * d.Ho * d.Wo / 128) * 8;   // [tile][4 waves][sum, sum of squares]
            out[m++] = g;
        }
    }
    return m;
}

// A launch that cannot fill the chip with 128-wide tiles (small batches: one 512x512 image gives the towers 43 tiles)
// is bound by ONE tile's K loop -- 2 us per 32-deep chunk.  With 32- (or 64-) wide tiles the same launch is 4 (2) times
// as many blocks whose chunks take a quarter (half) of the matrix time.  Results are bit-identical: every output is the
// same k-ordered chain whatever the tile shape, and the split-K decision is still taken on the 128-wide tile count, so
// the partial sums are cut at the same k.  -> tile code to use (t0 = keep) and that tile count.
static int narrow_tile_for_small_launch(const ml_conv2d_desc *descs, int n, int t0, bool have_ws, long long *ref_tiles) {
    *ref_tiles = -1;
    if (t0 == 3) return t0;
    const int ref_bn = t0 == 1 ? 128 : 64;
    long long tiles = 0;
    int n_live = 0;
    for (int i = 0; i < n; ++i) n_live += descs[i].live != nullptr;
    auto tiles_of = [&](const ml_conv2d_desc &d) {          // (live problems: the estimate launch_multi uses)
        const long long M = (long long)d.B * d.Ho * d.Wo;
        const long long t = ((M + 127) / 128) * (d.n_pad / ref_bn);
        return d.live ? (t + n_live - 1) / n_live : t;
    };
    for (int i = 0; i < n; ++i) {
        if (descs[i].group_cin_step) return t0;
        tiles += tiles_of(descs[i]);
    }
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ml_conv2d_desc &d = descs[i];
        const int chunks = d.KH * d.KW * (d.span_pad / (d.math == ML_MATH_F16S ? 64 : 32));
        const int splits = have_ws ? choose_splits(tiles, chunks) : 1;
        blocks += tiles_of(d) * splits;
    }
    const long long resident = ml_resident_blocks(2);
    // Fixed-capacity RoI batches (round 4): their live tiles are a third of the launch, known to the device only.  Cutting
    // K to fill the chip gave the 1-image mask head 616 blocks of 128 x 128 tiles -- one per CU, three rounds, 87 us per conv
    // and a reduce launch behind each; the same 616 blocks as 128 x 32 tiles of the WHOLE K sum sit three to a CU.  So for
    // these launches narrow tiles come before K slices (ref_tiles = 192: choose_splits cuts nothing from there), and the
    // narrow forms may overfill the resident slots by half (dead tiles return at once).  1 x 512^2 MobileNet graph: 1.71 ->
    // 1.59 ms on one box.  (The same preference for ALL small launches of moderate K -- towers, FPN -- was slower: 1.59-1.61
    // against 1.52-1.56 ms; their K slices stay.)
    if (n_live > 0 && ref_bn == 128) {
        if (tiles * 4 <= resident * 3 / 2) { *ref_tiles = 192; return 3; }
        if (tiles >= 192 && tiles * 2 <= resident * 3 / 2) { *ref_tiles = tiles; return 2; }
    }
    *ref_tiles = tiles;
    if (blocks * (ref_bn / 32) <= resident) return 3;
    if (ref_bn == 128 && blocks * 2 <= resident) return 2;
    *ref_tiles = -1;
    return t0;
}

// ML_MATH_F32X3, one 1x1 problem with a residual and a K loop of at most 8 chunks (the conv3 of ResNeXt stages 1 and 2):
// HBM-bound, and half of a tile's time is its epilogue (residual in, result out) -- 128 x 64 tiles put three blocks on a
// CU whose epilogues and K loops overlap (scripts/experiments/x3_tile_probe.py: 382 -> 355 us and 256 -> 238 us; longer
// K loops lose).  Bit-identical results (same k-ordered chains).
static int x3_adjust_tile(const ml_conv2d_desc *descs, int n, int t) {
    if (descs[0].math != ML_MATH_F32X3 || t != 1 || n != 1) return t;
    const ml_conv2d_desc &d = descs[0];
    const bool short_k_res = d.residual && d.KH == 1 && d.KW == 1 && d.span_pad / 32 <= 8 && !d.gn_partials && !d.live;
    return short_k_res ? 2 : t;
}

// ML_MATH_F32X3: 256 x 128 tiles (8 waves, one block per CU, 3-deep ring, software-pipelined steps) once they fill the
// chip -- 48 KB staged per chunk for twice the MFMAs of a 128-row tile's 32 KB.  Results are bit-identical to the 128-row kernel's: the same k-ordered chains.
static bool x3_uses_256_row_tiles(const ml_conv2d_desc *descs, int n, int t) {
    if (t != 1 || descs[0].math != ML_MATH_F32X3) return false;
    long long big = 0;
    for (int i = 0; i < n; ++i) {
        const long long M = (long long)descs[i].B * descs[i].Ho * descs[i].Wo;
        if (descs[i].gn_partials && M % 256) return false;      // (the partial sums are written by whole tiles only)
        // a residual is read in the epilogue (no registers to prefetch it) and one block per CU has nobody to cover that:
        // with a K loop of a few chunks only (the 128 -> 256 conv3 of stage 1) two 128-row blocks per CU are faster
        if (descs[i].residual && descs[i].KH * descs[i].KW * (descs[i].span_pad / 32) <= 4) return false;
        big += ((M + 255) / 256) * (descs[i].n_pad / 128);
    }
    return big >= ml_resident_blocks(1);
}

extern "C" int ml_conv2d_multi_f32(const ml_conv2d_desc *descs_in, int32_t n_in, void *workspace, int64_t workspace_bytes,
                                   void *stream) {
    ML_REQUIRE(descs_in != nullptr && n_in >= 1 && n_in <= MAXP, "conv2d: need 1..%d problems", MAXP);
    // the persistent 1x1 kernel has no tensor-size limit: try it before any splitting
    if (n_in == 1 && (descs_in[0].tile == 0 || descs_in[0].tile == 4 || descs_in[0].tile == 5)) {
        const int rc0 = validate(descs_in[0], false);
        if (rc0 != ML_OK) return rc0;
        ML_REQUIRE(descs_in[0].math >= ML_MATH_F32 && descs_in[0].math <= ML_MATH_F32X3, "conv2d: unknown math mode %d", descs_in[0].math);
        int took = 0;
        const bool half = descs_in[0].math == ML_MATH_F16S;
        if (half && h256_preferred(descs_in[0])) {
            const int rc = ml_conv1x1_h256_try(descs_in[0], reinterpret_cast<hipStream_t>(stream), &took);
            if (rc != ML_OK) return rc;
            if (took) return ML_OK;
            ML_REQUIRE(descs_in[0].tile != 5, "conv2d: tile = 5 (256 x 256 half kernel) does not apply to this problem");
        }
        if (half ? pipe_preferred_half(descs_in[0]) : (descs_in[0].tile == 4 || pipe_preferred(descs_in[0]))) {
            const int rc = ml_conv1x1_pipe_try(descs_in[0], reinterpret_cast<hipStream_t>(stream), &took);
            if (rc != ML_OK) return rc;
            if (took) return ML_OK;
        }
        ML_REQUIRE(descs_in[0].tile != 4, "conv2d: tile = 4 (pipelined 1x1 kernel) does not apply to this problem");
    }
    ml_conv2d_desc split[MAXP];
    const int n = split_by_image_groups(descs_in, n_in, split, MAXP);
    ML_REQUIRE(n != -2, "conv2d: a problem with `live` images must stay below 2 GiB of activations");
    ML_REQUIRE(n != -3, "conv2d: gn_partials on an activation >= 2 GiB needs image groups of whole 128-row tiles");
    ML_REQUIRE(n >= 1, "conv2d: too many >= 2 GiB activations in one launch (more than %d image groups)", MAXP);
    const ml_conv2d_desc *descs = split;
    int t0 = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = validate(descs[i]);
        if (rc != ML_OK) return rc;
        const int t = pick_tile(descs[i].cout, descs[i].tile);
        if (i == 0) t0 = t;
        ML_REQUIRE(t == t0, "conv2d: all problems of one launch must use the same tile shape");
        ML_REQUIRE(descs[i].math == descs[0].math, "conv2d: all problems of one launch must use the same math mode");
        ML_REQUIRE(!(descs[i].math == ML_MATH_F16S && descs[i].residual),
                   "conv2d: the fp16-storage form of the generic kernel takes no residual (only the persistent 1x1 kernel does: "
                   "stride 1, cout %% 128 == 0, span %% 64 == 0, half output)");
    }
    ML_REQUIRE(descs[0].math >= ML_MATH_F32 && descs[0].math <= ML_MATH_F32X3, "conv2d: unknown math mode %d", descs[0].math);
    if (workspace) ML_REQUIRE((((uintptr_t)workspace) & 255) == 0, "conv2d: workspace must be 256-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    long long ref_tiles = -1;
    const int t = x3_adjust_tile(descs, n, narrow_tile_for_small_launch(descs, n, t0, workspace != nullptr, &ref_tiles));
    if (descs[0].math == ML_MATH_F16S) {
        bool gns_h = false;
        for (int i = 0; i < n; ++i) gns_h = gns_h || descs[i].gn_partials != nullptr;
        ML_REQUIRE(!gns_h || t == 1, "conv2d: gn_partials needs the 128 x 128 kernel (a launch of ml_conv2d_gn_min_launch_tiles() tiles)");
        if (gns_h) return launch_multi<2, 2, 2, 2, ML_MATH_F16S, true>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        switch (t) {
            case 1: return launch_multi<2, 2, 2, 2, ML_MATH_F16S>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            case 2: return launch_multi<2, 2, 2, 1, ML_MATH_F16S>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            default: return launch_multi<4, 1, 1, 1, ML_MATH_F16S>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        }
    }
    if (descs[0].math == ML_MATH_F16) {
        switch (t) {
            case 1: return launch_multi<2, 2, 2, 2, ML_MATH_F16>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            case 2: return launch_multi<2, 2, 2, 1, ML_MATH_F16>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            default: return launch_multi<4, 1, 1, 1, ML_MATH_F16>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        }
    }
    bool any_gns = false;
    for (int i = 0; i < n; ++i) any_gns = any_gns || descs[i].gn_partials != nullptr;
    ML_REQUIRE(!any_gns || t == 1, "conv2d: gn_partials needs the fp32 128 x 128 kernel (a launch of >= 257 tiles)");
    if (descs[0].math == ML_MATH_F32X3) {
        // Wave layout 4 x 1 (a wave = 32 rows x the whole N tile): the A fragments -- the operand that is split in
        // registers -- are read and split by ONE wave instead of two (+3-5 % on the 3x3 convs against 2 x 2 waves).
        if (x3_uses_256_row_tiles(descs, n, t)) {
            if (any_gns) return launch_multi<8, 1, 1, 4, ML_MATH_F32X3, true, 3>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            return launch_multi<8, 1, 1, 4, ML_MATH_F32X3, false, 3>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        }
        if (any_gns) return launch_multi<4, 1, 1, 4, ML_MATH_F32X3, true>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        switch (t) {
            case 1: return launch_multi<4, 1, 1, 4, ML_MATH_F32X3>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            case 2: return launch_multi<4, 1, 1, 2, ML_MATH_F32X3>(descs, n, workspace, workspace_bytes, s, ref_tiles);
            default: return launch_multi<4, 1, 1, 1, ML_MATH_F32X3>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        }
    }
    if (any_gns) return launch_multi<2, 2, 2, 2, ML_MATH_F32, true>(descs, n, workspace, workspace_bytes, s, ref_tiles);
    switch (t) {
        case 1: return launch_multi<2, 2, 2, 2, ML_MATH_F32>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        case 2: return launch_multi<2, 2, 2, 1, ML_MATH_F32>(descs, n, workspace, workspace_bytes, s, ref_tiles);
        default: return launch_multi<4, 1, 1, 1, ML_MATH_F32>(descs, n, workspace, workspace_bytes, s, ref_tiles);
    }
}

// N-tile width (128 / 64 / 32) the generic kernel will use for this launch (after the small-launch narrowing); 0 on bad
// arguments.  For reporting: which kernel instantiation a launch's time belongs to.
extern "C" int ml_conv2d_launch_ntile(const ml_conv2d_desc *descs, int32_t n, int32_t has_workspace) {
    if (!descs || n < 1 || n > MAXP) return 0;
    const int t0 = pick_tile(descs[0].cout, descs[0].tile);
    long long ref_tiles = -1;
    const int t = x3_adjust_tile(descs, n, narrow_tile_for_small_launch(descs, n, t0, has_workspace != 0, &ref_tiles));
    return t == 1 ? 128 : (t == 2 ? 64 : 32);
}

// M-tile height (128 / 256) of the same launch; 0 on bad arguments.
extern "C" int ml_conv2d_launch_mtile(const ml_conv2d_desc *descs, int32_t n, int32_t has_workspace) {
    if (!descs || n < 1 || n > MAXP) return 0;
    const int t0 = pick_tile(descs[0].cout, descs[0].tile);
    long long ref_tiles = -1;
    const int t = x3_adjust_tile(descs, n, narrow_tile_for_small_launch(descs, n, t0, has_workspace != 0, &ref_tiles));
    return x3_uses_256_row_tiles(descs, n, t) ? 256 : 128;
}

// K slices per problem of the launch ml_conv2d_multi_f32 would make for these problems (1 = not split; the persistent
// 1x1 kernels never split).  splits: n ints.  For reporting / tests: which launches cut their K sum differently when
// the batch changes.  ML_OK, or ML_E_BADARG.
extern "C" int ml_conv2d_launch_splits(const ml_conv2d_desc *descs, int32_t n, int64_t workspace_bytes, int32_t *splits) {
    ML_REQUIRE(descs && splits && n >= 1 && n <= MAXP, "conv2d_launch_splits: need 1..%d problems", MAXP);
    for (int i = 0; i < n; ++i) splits[i] = 1;
    if (n == 1 && ml_conv2d_uses_pipe(descs)) return ML_OK;
    const int t0 = pick_tile(descs[0].cout, descs[0].tile);
    long long ref_tiles = -1;
    const int t = x3_adjust_tile(descs, n, narrow_tile_for_small_launch(descs, n, t0, workspace_bytes > 0, &ref_tiles));
    const int BM = x3_uses_256_row_tiles(descs, n, t) ? 256 : 128;
    const int BN = t == 1 ? 128 : (t == 2 ? 64 : 32);
    const int KC = descs[0].math == ML_MATH_F16S ? 64 : 32;
    plan_splits(descs, n, BM, BN, KC, workspace_bytes > 0, workspace_bytes, ref_tiles, splits);
    return ML_OK;
}

// Smallest launch (in 128 x 128 tiles, all problems together) that ml_conv2d_multi_f32 neither narrows to 128 x 64 / 128 x 32
// tiles nor cuts along K on THIS device -- the size from which ml_conv2d_desc.gn_partials may be used.
extern "C" int64_t ml_conv2d_gn_min_launch_tiles(void) {
    const long long narrow_below = ml_resident_blocks(2) / 2 + 1;      // (blocks * 2 <= resident -> 128 x 64 tiles)
    return narrow_below > 192 ? narrow_below : 192;                    // (choose_splits: no K slices from 192 tiles)
}

extern "C" int ml_conv2d_f32(const ml_conv2d_desc *dp, void *stream) {
    ML_REQUIRE(dp != nullptr, "conv2d: null descriptor");
    return ml_conv2d_multi_f32(dp, 1, nullptr, 0, stream);
}
