// Tail of the mask head in ONE kernel: Conv2DTranspose(2x2, stride 2) + bias + activation, then Conv2D 1x1 + bias +
// activation (reference engine/layers/instance.py:196-201 builds the pair, :226-233 runs it: `deconv` -> ReLU ->
// `output` -> sigmoid).  Unfused, the transposed conv writes a [R, 2h, 2w, C_mid] map (321 MB at 800 RoIs) that the
// 1x1 conv reads straight back, and both go through the generic (pixel-shuffle / sigmoid) epilogue of conv_mfma.hip.
//
// A 2x2 stride-2 transposed conv is four independent 1x1 GEMMs, one per output position q = (dy, dx):
//     T_q[pixel, c] = act( sum_k X[pixel, k] * Wd[q][c][k] + bd[c] )
// and the 1x1 conv contracts T_q over c for the same pixel.  A block owns (128 input pixels, one position q); its four
// waves own 32 pixels each and ALL C_mid channels, and compute the TRANSPOSED product T_q^T = Wd[q] * X^T on
// v_mfma_f32_32x32x2_f32 (A = weights, B = pixels).  The C/D layout then holds, in lane l, pixel l & 31 and -- in
// register e of channel tile t -- channel 32 t + (e & 3) + 8 (e >> 2) + 4 (l >> 5): exactly the A-operand layout
// (row = lane & 31, k = lane >> 5) of a second MFMA over the pixel rows.  So the accumulators feed the 1x1 conv
// directly, no LDS transposition, no HBM round trip: per register one MFMA whose B operand is the matching pair of
// rows of the 1x1 kernel (a lane table prepared by the host, see masklab_hip.h).
// Numerics: exact fp32 products; the 1x1 conv sums its channels in the order (t, e, lane half), not 0 .. C_mid-1
// (~1e-7 relative).
#include <type_traits>
#include "common.h"

namespace {

constexpr int TM = 128;                            // input pixels per block
constexpr int ROWB = 128;                          // bytes of K per staged row and chunk (32 floats)
constexpr int DO_MAX_PROB = 4;

struct DoProb {
    const void *x, *wd;          // fp32, or IEEE half when the launch is the fp16-storage one (HALF)
    const float *bd, *wo, *bo;
    float *out;
    const int *live;             // fixed-capacity RoI batch: RoI slot j of an image exists iff j < max(1, *live); or NULL
    long long M;                 // input pixels = rois * hw
    int hw, w, n_l, tile0;       // pixels per RoI map, map width, RoIs per image, first tile of this problem
    float r_hw, r_w, r_nl;       // reciprocals (index arithmetic below corrects the rounding)
    long long img_stride, base;  // out elements: between images / of this level's first RoI in image 0
};
struct DoArgs {
    DoProb p[DO_MAX_PROB];
    int nprob, K, ncls, cp, tiles, out_sigmoid;
    float mid_lo, mid_hi, out_lo, out_hi;      // activations as clamps (the output one unless out_sigmoid)
};

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}
// resource over [ptr, ptr + bytes).  Extents are kept in 32 bits (a tile's 128 rows; the launcher checks they fit): 64-bit
// compares / selects have no scalar form, and a descriptor that went through the VALU makes the compiler wrap every
// load using it in a readfirstlane loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)ptr, 0, (int)bytes, 0x00020000);
}
// n / d for 0 <= n < 2^24 (exact in float), d >= 1: one multiply and a two-sided correction
__device__ __forceinline__ int div_small(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    if (q * d > n) --q;
    if ((q + 1) * d <= n) ++q;
    return q;
}

// where a work unit lives (all scalar)
struct Loc {
    int pi;                      // problem index, -1 = no such unit
    const char *x;               // first pixel row of the tile
    unsigned x_bytes;            // bytes of the tile's rows that exist (rows past the problem's end read as zeros)
    const char *w;               // Wd[pos]
    int m0;                      // first pixel of the tile inside its problem
};

typedef _Float16 f16x8o __attribute__((ext_vector_type(8)));

// One K chunk of HALF tensors (64 deep = the same 128 bytes per row): 4 k-steps of v_mfma_f32_32x32x16_f16, the lane's
// ds_read_b128 holding the 8 halves it feeds; staging, swizzle and look-ahead loads as below.
template <int NT, bool FIRST>
__device__ __forceinline__ void do_chunk_h(f32x16 (&acc)[NT], const char *rd, char *wr, int x_off, int w_off, int h, int swz,
                                           int wave, const int (&x_voff)[4], const int (&w_voff)[NT],
                                           __amdgpu_buffer_rsrc_t rx_nx, __amdgpu_buffer_rsrc_t rw_nx, int soff) {
    constexpr int NSLOT = 4 * NT;
    f32x4 fx[2], fw[2][NT];
    auto read_frags = [&](int ks, f32x4 &x, f32x4 (&w)[NT]) {
        const int slot = ((ks * 2 + h) ^ swz) * 16;
        x = *reinterpret_cast<const f32x4 *>(rd + x_off + slot);
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const f32x4 *>(rd + w_off + t * 32 * ROWB + slot);
    };
    read_frags(0, fx[0], fw[0]);
    static_for<0, NSLOT>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;             // ks * NT + t
        constexpr int t = idx % NT, ks = idx / NT;
        if constexpr (t == NT / 2 && ks < 3) read_frags(ks + 1, fx[(ks + 1) & 1], fw[(ks + 1) & 1]);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f16x8o a = __builtin_bit_cast(f16x8o, fw[ks & 1][t]), b = __builtin_bit_cast(f16x8o, fx[ks & 1]);
        if constexpr (FIRST && ks == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (idx < 4) lds_dma16(rx_nx, wr + (32 * idx + 8 * wave) * ROWB, x_voff[idx], soff);
        else if constexpr (idx < 4 + NT) lds_dma16(rw_nx, wr + (TM + 32 * (idx - 4) + 8 * wave) * ROWB, w_voff[idx - 4], soff);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One K chunk (32 deep) of the wave's [C_mid channels x 32 pixels] slice; the next chunk's LDS-direct loads ride
// behind the first MFMAs.  FIRST: the chain starts from C = 0.
template <int NT, bool FIRST>
__device__ __forceinline__ void do_chunk(f32x16 (&acc)[NT], const char *rd, char *wr, int x_off, int w_off, int h, int swz,
                                         int wave, const int (&x_voff)[4], const int (&w_voff)[NT],
                                         __amdgpu_buffer_rsrc_t rx_nx, __amdgpu_buffer_rsrc_t rw_nx, int soff) {
    constexpr int NSLOT = 16 * NT;                  // MFMAs per chunk and wave
    f32x4 fx[2], fw[2][NT];
    auto read_frags = [&](int ks, f32x4 &x, f32x4 (&w)[NT]) {
        const int slot = ((ks * 2 + h) ^ swz) * 16;
        x = *reinterpret_cast<const f32x4 *>(rd + x_off + slot);
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const f32x4 *>(rd + w_off + t * 32 * ROWB + slot);
    };
    read_frags(0, fx[0], fw[0]);
    static_for<0, NSLOT>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;             // (ks * 4 + j) * NT + t
        constexpr int t = idx % NT, j = (idx / NT) & 3, ks = idx / (4 * NT);
        if constexpr (idx % (4 * NT) == 2 * NT && ks < 3) read_frags(ks + 1, fx[(ks + 1) & 1], fw[(ks + 1) & 1]);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (FIRST && ks == 0 && j == 0)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[0][t][0], fx[0][0], zero, 0, 0, 0);
        else
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[ks & 1][t][j], fx[ks & 1][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (idx < 4) lds_dma16(rx_nx, wr + (32 * idx + 8 * wave) * ROWB, x_voff[idx], soff);
        else if constexpr (idx < 4 + NT) lds_dma16(rw_nx, wr + (TM + 32 * (idx - 4) + 8 * wave) * ROWB, w_voff[idx - 4], soff);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// A block is persistent: it walks work units (tile of 128 pixels, position) u = block, block + grid, ... and its chunk
// stream crosses unit boundaries -- the first chunk of the next unit is fetched behind the last chunk's MFMAs and lands
// while the 1x1 conv and the stores of the finished unit run.
template <int NT, bool HALF>                        // C_mid = 32 NT; HALF: x and wd are IEEE half (fp16-storage mask head)
__global__ void __launch_bounds__(256, NT <= 4 ? 2 : 1)
deconv_out_kernel(const DoArgs A) {
    constexpr int CM = 32 * NT;
    constexpr int ES = HALF ? 2 : 4;                 // bytes per element of x / wd
    constexpr int KC = ROWB / ES;                    // K elements per chunk
    constexpr int BUFB = (TM + CM) * ROWB;
    extern __shared__ __align__(16) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int K = A.K, nk = K / KC;
    const int grid = gridDim.x;                      // a multiple of 32: a block keeps its position and its XCD slot

    // unit -> (tile, position): the four positions of a tile read the same pixels, so they sit on ONE XCD (block b
    // runs on XCD b & 7) in the same round, and three of the four reads hit its L2
    const int pos = ((int)blockIdx.x & 31) >> 3;
    // (constant indices + uniform selects: a dynamic index into the kernel argument would be copied to scratch memory)
    auto prob = [&](int pi) {
        DoProb P = A.p[0];
        if (pi == 1) P = A.p[1];
        if (pi == 2) P = A.p[2];
        if (pi == 3) P = A.p[3];
        return P;
    };
    // (advances `u` past units whose tile holds only RoI slots that do not exist -- fixed-capacity batches, `live`)
    auto locate = [&](int &u) {
        Loc L;
        L.pi = -1; L.x = reinterpret_cast<const char *>(A.p[0].x); L.x_bytes = 0; L.w = reinterpret_cast<const char *>(A.p[0].wd); L.m0 = 0;
        for (;; u += grid) {
            const int tile = (u >> 5) * 8 + (u & 7);
            if (tile >= A.tiles) break;
            int pi = 0;
            if (A.nprob > 1 && tile >= A.p[1].tile0) pi = 1;
            if (A.nprob > 2 && tile >= A.p[2].tile0) pi = 2;
            if (A.nprob > 3 && tile >= A.p[3].tile0) pi = 3;
            const DoProb P = prob(pi);
            const int m0t = (tile - P.tile0) * TM;
            if (P.live) {
                const int lim = max(1, *P.live);
                const int last = min(m0t + TM, (int)P.M) - 1;
                const int r0 = div_small(m0t, P.hw, P.r_hw), r1 = div_small(last, P.hw, P.r_hw);
                const int j0 = r0 - div_small(r0, P.n_l, P.r_nl) * P.n_l, j1 = r1 - div_small(r1, P.n_l, P.r_nl) * P.n_l;
                // all RoIs r0..r1 of the tile are dead iff they sit in ONE image (slots j0..j1, j1 - j0 == r1 - r0) whose
                // first slot here is already past the limit (a tile spanning >= 3 RoIs may cross into the next image)
                if (j0 >= lim && r1 - r0 == j1 - j0) continue;
            }
            L.pi = pi;
            L.m0 = m0t;
            L.x = reinterpret_cast<const char *>(P.x) + (long long)L.m0 * K * ES;
            const int rows = (int)P.M - L.m0;                 // >= 1: the tile exists
            L.x_bytes = (unsigned)((rows < TM ? rows : TM) * K * ES);
            L.w = reinterpret_cast<const char *>(P.wd) + (long long)pos * CM * K * ES;
            break;
        }
        return L;
    };

    // ---- LDS: [2 staging buffers: 128 pixel rows + CM weight rows, 128 B each, XOR-swizzled 16-byte groups]
    //           [1x1 lane table][transposed-conv bias][1x1 bias] of the CURRENT problem [4 wave-private result scratches]
    float *lds_wo = reinterpret_cast<float *>(lds + 2 * BUFB);
    const int n_wo = NT * 16 * 2 * A.cp;
    float *lds_bd = lds_wo + n_wo;
    float *lds_bo = lds_bd + CM;
    float *lds_scr = lds_bo + 32;                 // 4 waves x [32 pixels][cp] floats

    const int ld_row = tid >> 3;
    const int ld_g = (tid & 7) ^ ((ld_row >> 1) & 7);
    int x_voff[4], w_voff[NT];
#pragma unroll
    for (int i = 0; i < 4; ++i) x_voff[i] = (ld_row + 32 * i) * K * ES + ld_g * 16;
#pragma unroll
    for (int i = 0; i < NT; ++i) w_voff[i] = (ld_row + 32 * i) * K * ES + ld_g * 16;
    const int swz = (r >> 1) & 7;
    const int x_off = (32 * wave + r) * ROWB, w_off = (TM + r) * ROWB;
    const unsigned w_bytes = (unsigned)(CM * K * ES);

    int u = blockIdx.x;
    Loc cur = locate(u);
    if (cur.pi < 0) return;
    {   // chunk 0 of the first unit
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(cur.x, cur.x_bytes), rw = rsrc_of(cur.w, w_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rx, lds + (32 * i + 8 * wave) * ROWB, x_voff[i], 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) lds_dma16(rw, lds + (TM + 32 * i + 8 * wave) * ROWB, w_voff[i], 0);
    }
    int buf = 0, tab_pi = -1;
    const int col = r < A.cp ? r : A.cp - 1;                       // lanes past the padded class count read a dummy column
    const float *tab = lds_wo + h * A.cp + col;
    const int cp2 = 2 * A.cp;
    for (;;) {
        int u_nx = u + grid;
        const Loc nxt = locate(u_nx);
        if (cur.pi != tab_pi) {                       // first unit / the block crossed into the next RoI level
            const DoProb P = prob(cur.pi);
            __builtin_amdgcn_s_barrier();             // nobody reads the old tables any more
            for (int i = tid; i < n_wo; i += 256) lds_wo[i] = P.wo[i];
            for (int i = tid; i < CM; i += 256) lds_bd[i] = P.bd ? P.bd[i] : 0.f;
            if (tid < 32) lds_bo[tid] = (P.bo && tid < A.ncls) ? P.bo[tid] : 0.f;
            tab_pi = cur.pi;                          // (visible after the next chunk barrier, which every unit has)
        }
        const __amdgpu_buffer_rsrc_t rx = rsrc_of(cur.x, cur.x_bytes), rw = rsrc_of(cur.w, w_bytes);
        const __amdgpu_buffer_rsrc_t rx_n = rsrc_of(nxt.x, nxt.x_bytes), rw_n = rsrc_of(nxt.w, nxt.pi >= 0 ? w_bytes : 0u);
        f32x16 acc[NT];
        for (int kc = 0; kc < nk; ++kc) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this chunk has landed (my share; and my table writes)
            __builtin_amdgcn_s_barrier();                              // ... everyone's; and the other buffer is free
            const char *rd = lds + buf * BUFB;
            char *wr = lds + (buf ^ 1) * BUFB;
            const bool more = kc + 1 < nk;            // the next chunk: of this unit, or chunk 0 of the next one (or nothing)
            const __amdgpu_buffer_rsrc_t rx_nx = more ? rx : rx_n, rw_nx = more ? rw : rw_n;
            const int soff = more ? (kc + 1) * ROWB : 0;
            if constexpr (HALF) {
                if (kc == 0) do_chunk_h<NT, true>(acc, rd, wr, x_off, w_off, h, swz, wave, x_voff, w_voff, rx_nx, rw_nx, soff);
                else do_chunk_h<NT, false>(acc, rd, wr, x_off, w_off, h, swz, wave, x_voff, w_voff, rx_nx, rw_nx, soff);
            } else {
                if (kc == 0) do_chunk<NT, true>(acc, rd, wr, x_off, w_off, h, swz, wave, x_voff, w_voff, rx_nx, rw_nx, soff);
                else do_chunk<NT, false>(acc, rd, wr, x_off, w_off, h, swz, wave, x_voff, w_voff, rx_nx, rw_nx, soff);
            }
            buf ^= 1;
        }

        // ---- bias + activation, then the 1x1 conv straight from the accumulators: two independent chains
        const float bo = lds_bo[r];
        f32x16 y0, y1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { y0[e] = bo; y1[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds_bd + 32 * t + 8 * g + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    const int e = 4 * g + c;
                    const float a0 = __builtin_amdgcn_fmed3f(acc[t][e] + bv[c], A.mid_lo, A.mid_hi);
                    const float a1 = __builtin_amdgcn_fmed3f(acc[t][e + 1] + bv[c + 1], A.mid_lo, A.mid_hi);
                    y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, tab[(t * 16 + e) * cp2], y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, tab[(t * 16 + e + 1) * cp2], y1, 0, 0, 0);
                }
            }

        // ---- store.  The result sits as lane = class, register e = pixel (e & 3) + 8 (e >> 2) + 4 h of this wave's 32:
        // through a wave-private LDS scratch (LDS operations of one wave execute in order: no barrier) it becomes
        // lane = pixel, so the index arithmetic (three divisions) and the sigmoid run once per pixel and class instead
        // of 16 times per lane, and a pixel's classes leave as consecutive floats.
        {
            float *scr = lds_scr + wave * 32 * A.cp;
            if (r < A.cp) {
#pragma unroll
                for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * h) * A.cp + r] = y0[e] + y1[e];
            }
            const int mi = cur.m0 + 32 * wave + lane;
            const DoProb P = prob(cur.pi);            // (selected here, not before the K loop: 26 scalars less to keep alive)
            if (lane < 32 && mi < (int)P.M) {
                const int dy = pos >> 1, dx = pos & 1;
                const int per_roi = 4 * P.hw * A.ncls;
                const int roi = div_small(mi, P.hw, P.r_hw), rem = mi - roi * P.hw;
                const int y = div_small(rem, P.w, P.r_w), x = rem - y * P.w;
                const int img = div_small(roi, P.n_l, P.r_nl), j = roi - img * P.n_l;
                float *dst = P.out + ((long long)img * P.img_stride + P.base + (long long)j * per_roi +
                                      (long long)(((2 * y + dy) * 2 * P.w + 2 * x + dx) * A.ncls));
                const float *src = scr + lane * A.cp;
                for (int c = 0; c < A.ncls; ++c) {
                    const float v = src[c];
                    dst[c] = A.out_sigmoid ? 1.f / (1.f + expf(-v)) : __builtin_amdgcn_fmed3f(v, A.out_lo, A.out_hi);
                }
            }
        }
        if (nxt.pi < 0) break;
        cur = nxt;
        u = u_nx;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the last chunk's look-ahead loads were empty, but are counted)
}

template <int NT, bool HALF>
int launch_do(const DoArgs &A, hipStream_t s) {
    auto kern = deconv_out_kernel<NT, HALF>;
    const int bytes = 2 * (TM + 32 * NT) * ROWB + (NT * 16 * 2 * A.cp + 32 * NT + 32 + 4 * 32 * A.cp) * 4;
    static std::atomic<unsigned long long> ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 2 * (TM + 32 * NT) * ROWB + (NT * 16 * 2 * 32 + 32 * NT + 32 + 4 * 32 * 32) * 4,
                                       ok, "deconv2x2_out1x1"))
        return rc;
    const int units = ((A.tiles + 7) / 8) * 32;          // (tile, position) pairs, tiles rounded up to whole groups of 8
    const int resident = ml_resident_blocks(NT <= 4 ? 2 : 1);
    const int grid = units < resident ? units : resident;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, s, A);
    ML_CHECK_LAUNCH("deconv2x2_out1x1");
    return ML_OK;
}

}  // namespace

static int deconv_out_entry(const ml_deconv_out_problem *probs, int32_t nprob, int32_t K, int32_t c_mid, int32_t ncls,
                            int32_t cp, int32_t act_mid, int32_t act_out, void *stream, bool half) {
    const int es = half ? 2 : 4, kc = half ? 64 : 32;
    ML_REQUIRE(probs && nprob >= 1 && nprob <= DO_MAX_PROB, "deconv2x2_out1x1: 1..%d problems per launch", DO_MAX_PROB);
    ML_REQUIRE(K >= kc && K % kc == 0, "deconv2x2_out1x1: input channels (%d) must be a multiple of %d", K, kc);
    ML_REQUIRE(c_mid == 128 || c_mid == 256, "deconv2x2_out1x1: transposed-conv filters must be 128 or 256 (got %d)", c_mid);
    ML_REQUIRE(ncls >= 1 && ncls <= 32 && cp >= ncls && cp <= 32 && (cp & (cp - 1)) == 0,
               "deconv2x2_out1x1: 1..32 classes, table column count a power of two >= classes (got %d, %d)", ncls, cp);
    DoArgs A;
    ML_REQUIRE(act_mid == ML_ACT_NONE || act_mid == ML_ACT_RELU || act_mid == ML_ACT_RELU6,
               "deconv2x2_out1x1: the transposed conv's activation must be none / relu / relu6");
    const float big = 3.402823466e38f;
    A.nprob = nprob; A.K = K; A.ncls = ncls; A.cp = cp;
    A.mid_lo = act_mid == ML_ACT_NONE ? -big : 0.f; A.mid_hi = act_mid == ML_ACT_RELU6 ? 6.f : big;
    A.out_sigmoid = act_out == ML_ACT_SIGMOID;
    A.out_lo = (act_out == ML_ACT_RELU || act_out == ML_ACT_RELU6) ? 0.f : -big; A.out_hi = act_out == ML_ACT_RELU6 ? 6.f : big;
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        const ml_deconv_out_problem &q = probs[i];
        ML_REQUIRE(q.x && q.wd && q.wo_table && q.out, "deconv2x2_out1x1: null pointer in problem %d", i);
        ML_REQUIRE(ml_aligned16(q.x) && ml_aligned16(q.wd) && (!q.bd || ml_aligned16(q.bd)),
                   "deconv2x2_out1x1: x / wd / bd must be 16-byte aligned (problem %d)", i);
        ML_REQUIRE(q.M >= 1 && q.M < (1ll << 24), "deconv2x2_out1x1: 1 <= input pixels < 2^24 per problem (got %lld)", (long long)q.M);
        ML_REQUIRE(q.w >= 1 && q.hw >= q.w && q.hw % q.w == 0 && q.rois_per_image >= 1 && q.M % q.hw == 0 &&
                       (q.M / q.hw) % q.rois_per_image == 0,
                   "deconv2x2_out1x1: problem %d: pixels (%lld) must be whole maps of %d (width %d) for whole images of %d RoIs",
                   i, (long long)q.M, q.hw, q.w, q.rois_per_image);
        ML_REQUIRE((long long)(TM + 256) * K * es < (1ll << 31), "deconv2x2_out1x1: K too large");
        DoProb &P = A.p[i];
        P.x = q.x; P.wd = q.wd; P.bd = q.bd; P.wo = q.wo_table; P.bo = q.bo; P.out = q.out; P.live = q.live;
        if (q.live) ML_REQUIRE((long long)q.hw * q.rois_per_image >= TM, "deconv2x2_out1x1: `live` needs at least one tile per image");
        P.M = q.M; P.hw = q.hw; P.w = q.w; P.n_l = q.rois_per_image; P.tile0 = tiles;
        P.r_hw = 1.f / (float)q.hw; P.r_w = 1.f / (float)q.w; P.r_nl = 1.f / (float)q.rois_per_image;
        P.img_stride = q.out_image_stride; P.base = q.out_base;
        tiles += (int)((q.M + TM - 1) / TM);
    }
    for (int i = nprob; i < DO_MAX_PROB; ++i) A.p[i] = A.p[0];
    A.tiles = tiles;
    if (half) return c_mid == 128 ? launch_do<4, true>(A, (hipStream_t)stream) : launch_do<8, true>(A, (hipStream_t)stream);
    return c_mid == 128 ? launch_do<4, false>(A, (hipStream_t)stream) : launch_do<8, false>(A, (hipStream_t)stream);
}

extern "C" int ml_deconv2x2_out1x1_f32(const ml_deconv_out_problem *probs, int32_t nprob, int32_t K, int32_t c_mid,
                                       int32_t ncls, int32_t cp, int32_t act_mid, int32_t act_out, void *stream) {
    return deconv_out_entry(probs, nprob, K, c_mid, ncls, cp, act_mid, act_out, stream, false);
}

extern "C" int ml_deconv2x2_out1x1_f16(const ml_deconv_out_problem *probs, int32_t nprob, int32_t K, int32_t c_mid,
                                       int32_t ncls, int32_t cp, int32_t act_mid, int32_t act_out, void *stream) {
    return deconv_out_entry(probs, nprob, K, c_mid, ncls, cp, act_mid, act_out, stream, true);
}
