// The reference's chunk-wise GroupNormalization (engine/normalization.py:116-160).
// Sample n's flat H*W*C vector is cut into G contiguous chunks of L = HWC/G floats;
//   y = (x - mean_g) / sqrt(var_g + eps) * gamma[j] + beta[j],  j = g*(C/G) + (c mod C/G).
// HBM-bound: algorithmic traffic = 1 read + 1 write of x (8 B/elt).
// Chunks of up to 4096 floats (the small pyramid levels, all RoI maps) take the ONE-PASS kernel: one
// block per (sample, chunk) holds the chunk in registers (<= 4 float4 per thread), reduces mean and
// then sum((x-mean)^2) across the block, normalises and writes -- 8 B/elt, one launch.
// Larger chunks read x twice (stats pass + apply pass) = 12 B/elt, the second read usually served by
// L2/Infinity Cache:
//   pass 1: grid (S, N*G): each block sums a slice of one chunk in fp64 (sum, sum of squares)
//           -> workspace[(n*G+g)*S + s]   (fp64 partials: no cancellation issue in E[x^2]-mean^2)
//   pass 2: grid (S2, N*G): each block folds the S partials, then normalises its slice.
// Storage type T: float, or IEEE half (the heads of the fp16 path: x and y are half, 8 per 16-byte access; statistics
// are fp64 sums of the stored values, the normalisation runs in fp32 and is rounded ONCE at the store).
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));

// one 16-byte access = VW<T> elements, handled as floats
template <class T> struct VW { static constexpr int value = 16 / sizeof(T); };
template <class T>
__device__ __forceinline__ void vload(const T *p, float (&v)[VW<T>::value]) {
    if constexpr (std::is_same<T, float>::value) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(p);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
        const f16x8g x = *reinterpret_cast<const f16x8g *>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
    }
}
template <class T>
__device__ __forceinline__ void vstore(T *p, const float (&v)[VW<T>::value]) {
    if constexpr (std::is_same<T, float>::value) {
        const f32x4 x = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(p) = x;
    } else {
        const f16x8g x = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3],
                          (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
        *reinterpret_cast<f16x8g *>(p) = x;
    }
}

constexpr int GN_TPB = 256;
constexpr int GN_MAX_SPLIT = 64;
// Largest chunk the one-pass (register-resident) kernel takes.  Measured on MI355X: one block per chunk wins
// while the chunk is small (8x8 .. 32x32 maps, 14x14 RoI maps: 6.3 vs 9.8 us, 8.0 vs 10.2 us); from 64x64x128
// maps up (chunk 32 768 floats, 128 fat blocks) the sliced two-pass form has the parallelism and is faster
// (19.8 vs 23.7 us), even though it reads x twice.
constexpr int GN_ONEPASS_MAX = 4096;

struct GnPlan { int S; long long slice; };

// slices are multiples of vw*GN_TPB elements (vw = 4 floats / 8 halves per access) so every block runs whole sweeps
static GnPlan gn_plan(long long L, int NG, int vw = 4) {
    long long want = (2048 + NG - 1) / NG;            // aim at >= 2048 blocks on the chip
    if (want < 1) want = 1;
    if (want > GN_MAX_SPLIT) want = GN_MAX_SPLIT;
    const long long unit = (long long)vw * GN_TPB;
    long long slice = ((L + want - 1) / want + unit - 1) / unit * unit;
    int S = (int)((L + slice - 1) / slice);
    return {S, slice};
}

template <class T, bool VEC4 = true>
__device__ __forceinline__ void gn_stats_body(const T *__restrict__ x, double *__restrict__ ws, long long L,
                                              long long slice, int S, int ng, int s) {
    constexpr int W = VW<T>::value;
    const T *p = x + (long long)ng * L;
    const long long lo = (long long)s * slice;
    const long long hi = min(lo + slice, L);
    double sum = 0.0, sq = 0.0;
    if (VEC4) {
        // a vector is folded in fp32 first (3 adds, 4 fma per 4 values), then joins the fp64 running sums: the kernel
        // was bound by its fp64 instruction count (12 per float4), not by HBM
        for (long long i = lo + threadIdx.x * W; i < hi; i += GN_TPB * W) {
            float v[W];
            vload<T>(p + i, v);
            float s4 = (v[0] + v[1]) + (v[2] + v[3]);
            float q4 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
            if constexpr (W == 8) {
                s4 += (v[4] + v[5]) + (v[6] + v[7]);
                q4 += fmaf(v[7], v[7], fmaf(v[6], v[6], fmaf(v[5], v[5], v[4] * v[4])));
            }
            sum += (double)s4;
            sq += (double)q4;
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += GN_TPB) { const double d = (double)(float)p[i]; sum += d; sq += d * d; }
    }
    // wave reduce (64 lanes) then across the 4 waves
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off, 64);
        sq += __shfl_down(sq, off, 64);
    }
    __shared__ double red[2][GN_TPB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sum; red[1][wave] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < GN_TPB / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        ws[((long long)ng * S + s) * 2 + 0] = a;
        ws[((long long)ng * S + s) * 2 + 1] = b;
    }
}

template <class T, bool VEC4>
__global__ void gn_stats_kernel(const T *__restrict__ x, double *__restrict__ ws, long long L, long long slice,
                                int S) {
    gn_stats_body<T, VEC4>(x, ws, L, slice, S, blockIdx.y, blockIdx.x);
}

template <class T, bool VEC4 = true>
__device__ __forceinline__ void gn_apply_body(const T *x, T *y, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, const double *__restrict__ ws, long long L,
                                              long long slice, int S, int C, int G, float eps, int relu, int out_cs,
                                              int out_co, int ng, int s, int Sp = -1) {
    constexpr int W = VW<T>::value;
    const int g = ng % G;
    const int cg = C / G;
    if (Sp < 0) Sp = S;
    double sum = 0, sq = 0;
    for (int i = 0; i < Sp; ++i) {  // uniform, L2-resident, fixed order
        sum += ws[((long long)ng * Sp + i) * 2 + 0];
        sq += ws[((long long)ng * Sp + i) * 2 + 1];
    }
    const double meand = sum / (double)L;
    double vard = sq / (double)L - meand * meand;
    if (vard < 0) vard = 0;
    const float mean = (float)meand;
    const float rstd = (float)(1.0 / sqrt(vard + (double)eps));
    const T *p = x + (long long)ng * L;
    const bool dense = (out_cs == C);
    // dense: y has x's layout.  sliced: element f of the sample -> y[n][f / C][out_co + f % C]
    const long long HWC = L * G;
    T *q = dense ? y + (long long)ng * L : y + (long long)(ng / G) * (HWC / C) * out_cs + out_co;
    const long long lo = (long long)s * slice;
    const long long hi = min(lo + slice, L);
    // flat index inside the sample = g*L + i ; channel = that mod C
    const long long gbase = (long long)g * L;
    if (VEC4 && (cg % W == 0) && HWC < (1ll << 31)) {
        // Hot form: a vector never straddles a gamma period (cg % W == 0) and the flat index fits 32 bits, so
        // channel / pixel / gamma offset advance incrementally -- no division or modulo inside the loop (the
        // 64-bit `%` and `/` per vector made this pass VALU-bound)
        const unsigned Cu = (unsigned)C, cgu = (unsigned)cg;
        const long long i0 = lo + threadIdx.x * W;
        const unsigned f0 = (unsigned)(gbase + i0);
        unsigned c0 = f0 % Cu, pix = f0 / Cu, cm = c0 % cgu;
        const unsigned stepc = (GN_TPB * (unsigned)W) % Cu, steppix = (GN_TPB * (unsigned)W) / Cu, stepm = (GN_TPB * (unsigned)W) % cgu;
        const float *gp = gamma ? gamma + g * cg : nullptr;
        const float *bp = beta ? beta + g * cg : nullptr;
        for (long long i = i0; i < hi; i += GN_TPB * W) {
            float v[W], o[W];
            vload<T>(p + i, v);
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = (v[e] - mean) * rstd;
            if (gp) {
#pragma unroll
                for (int e4 = 0; e4 < W; e4 += 4) {
                    const f32x4 gv = *reinterpret_cast<const f32x4 *>(gp + cm + e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e4 + e] *= gv[e];
                }
            }
            if (bp) {
#pragma unroll
                for (int e4 = 0; e4 < W; e4 += 4) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp + cm + e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e4 + e] += bv[e];
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < W; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            if (dense) vstore<T>(q + i, o);
            else vstore<T>(q + (long long)pix * out_cs + c0, o);
            c0 += stepc;
            pix += steppix;
            if (c0 >= Cu) { c0 -= Cu; ++pix; }
            cm += stepm;
            if (cm >= cgu) cm -= cgu;
        }
    } else if (VEC4) {
        for (long long i = lo + threadIdx.x * W; i < hi; i += GN_TPB * W) {
            float v[W], o[W];
            vload<T>(p + i, v);
            const int c0 = (int)((gbase + i) % C);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                int c = c0 + e;
                if (c >= C) c -= C;
                const int j = g * cg + (c % cg);
                float t = (v[e] - mean) * rstd;
                if (gamma) t *= gamma[j];
                if (beta) t += beta[j];
                o[e] = relu ? fmaxf(t, 0.f) : t;
            }
            if (dense) vstore<T>(q + i, o);
            else vstore<T>(q + ((gbase + i) / C) * out_cs + c0, o);
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += GN_TPB) {
            const int c = (int)((gbase + i) % C);
            const int j = g * cg + (c % cg);
            float t = ((float)p[i] - mean) * rstd;
            if (gamma) t *= gamma[j];
            if (beta) t += beta[j];
            const float r = relu ? fmaxf(t, 0.f) : t;
            if (dense) q[i] = (T)r;
            else q[((gbase + i) / C) * out_cs + c] = (T)r;
        }
    }
}

template <class T, bool VEC4>
__global__ void gn_apply_kernel(const T *x, T *y, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const double *__restrict__ ws, long long L,
                                long long slice, int S, int C, int G, float eps, int relu, int out_cs, int out_co) {
    gn_apply_body<T, VEC4>(x, y, gamma, beta, ws, L, slice, S, C, G, eps, relu, out_cs, out_co, blockIdx.y, blockIdx.x);
}

// ---- one-pass form: the chunk lives in registers.  Block (TPB threads) = one (sample, chunk).
template <int TPB>
__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                       // `red` may still be read from the previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) t += red[w];
    return t;
}

template <class T, int TPB, int VPT>
__device__ __forceinline__ void gn_onepass_body(const T *__restrict__ x, T *__restrict__ y,
                                                const float *__restrict__ gamma, const float *__restrict__ beta, int L,
                                                int C, int G, float eps, int relu, int out_cs, int out_co, int ng) {
    constexpr int W = VW<T>::value;
    __shared__ double red[TPB / 64];
    const int g = ng % G;
    const int cg = C / G;
    const T *p = x + (long long)ng * L;
    float v[VPT][W];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * W;
#pragma unroll
        for (int e = 0; e < W; ++e) v[k][e] = 0.f;
        if (i < L) {
            vload<T>(p + i, v[k]);
            sum += ((double)v[k][0] + (double)v[k][1]) + ((double)v[k][2] + (double)v[k][3]);
            if constexpr (W == 8) sum += ((double)v[k][4] + (double)v[k][5]) + ((double)v[k][6] + (double)v[k][7]);
        }
    }
    const double meand = block_sum<TPB>(sum, red) / (double)L;
    const float mean = (float)meand;
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * W;
        if (i < L) {
#pragma unroll
            for (int e = 0; e < W; ++e) { const double d = (double)v[k][e] - meand; sq += d * d; }
        }
    }
    const double vard = block_sum<TPB>(sq, red) / (double)L;
    const float rstd = (float)(1.0 / sqrt(vard + (double)eps));
    const bool dense = (out_cs == C);
    const long long HWC = (long long)L * G;
    T *q = dense ? y + (long long)ng * L : y + (long long)(ng / G) * (HWC / C) * out_cs + out_co;
    const long long gbase = (long long)g * L;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * W;
        if (i >= L) continue;
        const int c0 = (int)((gbase + i) % C);
        float o[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            int c = c0 + e;
            if (c >= C) c -= C;
            const int j = g * cg + (c % cg);
            float t = (v[k][e] - mean) * rstd;
            if (gamma) t *= gamma[j];
            if (beta) t += beta[j];
            o[e] = relu ? fmaxf(t, 0.f) : t;
        }
        if (dense) vstore<T>(q + i, o);
        else vstore<T>(q + ((gbase + i) / C) * out_cs + c0, o);
    }
}

template <class T, int TPB, int VPT>
__global__ void __launch_bounds__(TPB)
gn_onepass_kernel(const T *__restrict__ x, T *__restrict__ y, const float *__restrict__ gamma,
                  const float *__restrict__ beta, int L, int C, int G, float eps, int relu, int out_cs, int out_co) {
    gn_onepass_body<T, TPB, VPT>(x, y, gamma, beta, L, C, G, eps, relu, out_cs, out_co, blockIdx.x);
}

template <class T, int TPB, int VPT>
void launch_onepass(const T *x, T *y, const float *gamma, const float *beta, int NG, int L, int C, int G,
                    float eps, int relu, int out_cs, int out_co, hipStream_t s) {
    hipLaunchKernelGGL((gn_onepass_kernel<T, TPB, VPT>), dim3(NG), dim3(TPB), 0, s, x, y, gamma, beta, L, C, G, eps, relu,
                       out_cs, out_co);
}


// ------------------------------------------------------------------ several problems in one launch pair
// The un-shared towers normalise five pyramid levels (and the mask head three RoI levels) one after the other:
// the two coarsest levels are a few thousand floats each, i.e. a ~10 us launch apiece behind the 46 us P3 launch.
// ml_groupnorm_multi_f32 runs all of them as (1) one statistics launch over the problems with large chunks and
// (2) one launch that applies those AND normalises the small-chunk problems in their register-resident one-pass
// form.  Same per-problem arithmetic as the single-problem kernels (bit-identical results).
struct GnProb {
    const void *x;            // T = float or _Float16 (one type per launch)
    void *y;
    const float *gamma, *beta;
    double *ws;               // partials of this problem (two-pass only)
    long long L, slice;
    int NG, S, C, G, relu, out_cs, out_co, onepass_vpt;   // onepass_vpt: 0 = two-pass, else float4 per thread
    int Sp;                   // (sum, sum of squares) pairs per chunk in `ws`: S from gn_stats, or the producing conv's tiles
    float eps;
    const int *live;          // fixed-capacity RoI batches: sample n is live iff n % live_period < max(1, *live)
    int live_period;
};
__device__ __forceinline__ bool gn_dead(const GnProb &P, int ng) {
    return P.live && ((ng / P.G) % P.live_period) >= max(1, *P.live);
}
struct GnMulti {
    int n;
    int start[ML_GN_MAX_PROBLEMS + 1];
    GnProb p[ML_GN_MAX_PROBLEMS];
};

template <class T>
__global__ void __launch_bounds__(GN_TPB)
gn_multi_stats_kernel(const GnMulti A) {
    int pi = 0;
    while (pi + 1 < A.n && (int)blockIdx.x >= A.start[pi + 1]) ++pi;
    const GnProb &P = A.p[pi];
    const int id = blockIdx.x - A.start[pi];
    if (gn_dead(P, id / P.S)) return;
    gn_stats_body<T>(reinterpret_cast<const T *>(P.x), P.ws, P.L, P.slice, P.S, id / P.S, id % P.S);
}

template <class T>
__global__ void __launch_bounds__(GN_TPB)
gn_multi_apply_kernel(const GnMulti A) {
    int pi = 0;
    while (pi + 1 < A.n && (int)blockIdx.x >= A.start[pi + 1]) ++pi;
    const GnProb &P = A.p[pi];
    const int id = blockIdx.x - A.start[pi];
    const T *x = reinterpret_cast<const T *>(P.x);
    T *y = reinterpret_cast<T *>(P.y);
    if (gn_dead(P, P.onepass_vpt == 0 ? id / P.S : id)) return;          // (block-uniform)
    if (P.onepass_vpt == 0) {
        gn_apply_body<T>(x, y, P.gamma, P.beta, P.ws, P.L, P.slice, P.S, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co,
                         id / P.S, id % P.S, P.Sp);
    } else if (P.onepass_vpt == 1) {
        gn_onepass_body<T, GN_TPB, 1>(x, y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    } else if (P.onepass_vpt == 2) {
        gn_onepass_body<T, GN_TPB, 2>(x, y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    } else {
        if constexpr (std::is_same<T, float>::value)
            gn_onepass_body<T, GN_TPB, 4>(x, y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    }
}

}  // namespace

extern "C" int64_t ml_groupnorm_workspace_bytes(int32_t N, int32_t G) {
    return (int64_t)N * G * GN_MAX_SPLIT * 2 * (int64_t)sizeof(double);
}

static int gn_validate(const void *x, void *y, int32_t N, int64_t HWC, int32_t C, int32_t G, int32_t out_cstride,
                       int32_t out_coff) {
    ML_REQUIRE(x && y, "groupnorm: null pointer");
    ML_REQUIRE(N > 0 && HWC > 0 && C > 0 && G > 0, "groupnorm: bad dims");
    ML_REQUIRE(C >= G, "groupnorm: Number of groups (%d) cannot be more than the number of channels (%d).", G, C);
    ML_REQUIRE(C % G == 0, "groupnorm: Number of groups (%d) must be a multiple of the number of channels (%d).", G, C);
    ML_REQUIRE(HWC % C == 0 && HWC % G == 0, "groupnorm: H*W*C (%lld) must be divisible by C and by G", (long long)HWC);
    ML_REQUIRE((long long)N * G < 65536, "groupnorm: N*G too large for grid.y");
    ML_REQUIRE(out_cstride >= C && out_coff >= 0 && out_coff + C <= out_cstride, "groupnorm: bad output slice");
    ML_REQUIRE(out_cstride == C ? out_coff == 0 : true, "groupnorm: dense output must have out_coff 0");
    return ML_OK;
}

template <class T>
static int gn_multi(const ml_gn_desc *descs, int32_t n, void *workspace, int64_t workspace_bytes, void *stream) {
    constexpr int W = VW<T>::value;
    GnMulti st, ap;
    st.n = 0;
    ap.n = n;
    long long sb = 0, ab = 0, ws_off = 0;
    for (int i = 0; i < n; ++i) {
        const ml_gn_desc &d = descs[i];
        if (int rc = gn_validate(d.x, d.y, d.N, d.HWC, d.C, d.G, d.out_cstride, d.out_coff)) return rc;
        const long long L = d.HWC / d.G;
        const bool vec = (L % W == 0) && (d.C % W == 0) && (d.out_cstride % W == 0) && (d.out_coff % W == 0) &&
                         ml_aligned16(d.x) && ml_aligned16(d.y);
        ML_REQUIRE(vec, "groupnorm_multi: problem %d needs 16-byte aligned tensors and chunk / channel counts that are "
                   "multiples of %d (use ml_groupnorm_chunk_f32 / _f16 otherwise)", i, W);
        GnProb P;
        P.x = d.x; P.y = d.y; P.gamma = d.gamma; P.beta = d.beta;
        P.L = L; P.NG = d.N * d.G; P.C = d.C; P.G = d.G; P.relu = d.relu; P.out_cs = d.out_cstride; P.out_co = d.out_coff;
        P.eps = d.eps;
        P.live = d.live; P.live_period = d.live_period;
        if (d.live) ML_REQUIRE(d.live_period >= 1 && d.N % d.live_period == 0, "groupnorm_multi: live_period must divide N");
        P.ws = nullptr; P.S = 1; P.slice = L; P.onepass_vpt = 0; P.Sp = -1;
        if (L <= GN_ONEPASS_MAX) {
            const int vn = (int)((L + W - 1) / W);
            P.onepass_vpt = vn <= 256 ? 1 : (vn <= 512 ? 2 : 4);
            ap.start[i] = (int)ab;
            ab += P.NG;
        } else if (d.partials) {
            // the producing conv summed its tiles (ml_conv2d_desc.gn_partials): no statistics pass over this tensor
            ML_REQUIRE(d.n_partials >= 1 && d.n_partials <= 4096,
                       "groupnorm_multi: problem %d: 1..4096 fp64 (sum, sum of squares) pairs per chunk", i);
            const GnPlan plan = gn_plan(L, P.NG, W);
            P.S = plan.S; P.slice = plan.slice;
            P.ws = const_cast<double *>(d.partials);
            P.Sp = d.n_partials;
            ap.start[i] = (int)ab;
            ab += (long long)P.NG * P.S;
        } else {
            const GnPlan plan = gn_plan(L, P.NG, W);
            P.S = plan.S; P.slice = plan.slice;
            const long long bytes = (long long)P.NG * P.S * 2 * (long long)sizeof(double);
            ML_REQUIRE(ws_off + bytes <= workspace_bytes, "groupnorm_multi: workspace too small (%lld bytes needed so far)",
                       ws_off + bytes);
            P.ws = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + ws_off);
            ws_off += (bytes + 255) / 256 * 256;
            st.start[st.n] = (int)sb;
            st.p[st.n++] = P;
            sb += (long long)P.NG * P.S;
            ap.start[i] = (int)ab;
            ab += (long long)P.NG * P.S;
        }
        ap.p[i] = P;
        ML_REQUIRE(ab < (1ll << 31) && sb < (1ll << 31), "groupnorm_multi: grid too large");
    }
    st.start[st.n] = (int)sb;
    ap.start[n] = (int)ab;
    hipStream_t s = (hipStream_t)stream;
    if (st.n > 0) hipLaunchKernelGGL(gn_multi_stats_kernel<T>, dim3((unsigned)sb), dim3(GN_TPB), 0, s, st);
    hipLaunchKernelGGL(gn_multi_apply_kernel<T>, dim3((unsigned)ab), dim3(GN_TPB), 0, s, ap);
    ML_CHECK_LAUNCH("groupnorm_multi");
    return ML_OK;
}

extern "C" int ml_groupnorm_multi_f32(const ml_gn_desc *descs, int32_t n, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
    ML_REQUIRE(descs && n >= 1 && n <= ML_GN_MAX_PROBLEMS && workspace, "groupnorm_multi: need 1..%d problems and a workspace",
               ML_GN_MAX_PROBLEMS);
    for (int i = 1; i < n; ++i)
        ML_REQUIRE(descs[i].dtype == descs[0].dtype, "groupnorm_multi: the problems of one launch must share the storage type");
    ML_REQUIRE(descs[0].dtype == 0 || descs[0].dtype == 1, "groupnorm_multi: dtype must be 0 (float) or 1 (half)");
    return descs[0].dtype ? gn_multi<_Float16>(descs, n, workspace, workspace_bytes, stream)
                          : gn_multi<float>(descs, n, workspace, workspace_bytes, stream);
}

template <class T>
static int gn_single(const T *x, T *y, const float *gamma, const float *beta, int32_t N, int64_t HWC, int32_t C, int32_t G,
                     float eps, int32_t relu, int32_t out_cstride, int32_t out_coff, void *workspace, void *stream) {
    constexpr int W = VW<T>::value;
    ML_REQUIRE(workspace, "groupnorm: null pointer");
    if (int rc = gn_validate(x, y, N, HWC, C, G, out_cstride, out_coff)) return rc;
    const long long L = HWC / G;
    const bool vec = (L % W == 0) && (C % W == 0) && (out_cstride % W == 0) && (out_coff % W == 0) &&
                     ml_aligned16(x) && ml_aligned16(y);
    const GnPlan plan = gn_plan(L, N * G, vec ? W : 4);
    hipStream_t s = (hipStream_t)stream;
    double *ws = reinterpret_cast<double *>(workspace);
    const dim3 grid(plan.S, N * G);
    if (vec && L <= GN_ONEPASS_MAX) {
        // one pass, chunk in registers: vectors per thread chosen so that 256 * VPT * W >= L
        const int NG = N * G, Li = (int)L;
        const int vn = (Li + W - 1) / W;
#define GN1(TPB, VPT) launch_onepass<T, TPB, VPT>(x, y, gamma, beta, NG, Li, C, G, eps, relu, out_cstride, out_coff, s)
        if (vn <= 256) GN1(256, 1);
        else if (vn <= 512) GN1(256, 2);
        else if constexpr (W == 4) GN1(256, 4);
#undef GN1
    } else if (vec) {
        hipLaunchKernelGGL((gn_stats_kernel<T, true>), grid, dim3(GN_TPB), 0, s, x, ws, L, plan.slice, plan.S);
        hipLaunchKernelGGL((gn_apply_kernel<T, true>), grid, dim3(GN_TPB), 0, s, x, y, gamma, beta, ws, L, plan.slice, plan.S, C,
                           G, eps, relu, out_cstride, out_coff);
    } else {
        hipLaunchKernelGGL((gn_stats_kernel<T, false>), grid, dim3(GN_TPB), 0, s, x, ws, L, plan.slice, plan.S);
        hipLaunchKernelGGL((gn_apply_kernel<T, false>), grid, dim3(GN_TPB), 0, s, x, y, gamma, beta, ws, L, plan.slice, plan.S,
                           C, G, eps, relu, out_cstride, out_coff);
    }
    ML_CHECK_LAUNCH("groupnorm");
    return ML_OK;
}

extern "C" int ml_groupnorm_chunk_f32(const float *x, float *y, const float *gamma, const float *beta, int32_t N,
                                      int64_t HWC, int32_t C, int32_t G, float eps, int32_t relu, int32_t out_cstride,
                                      int32_t out_coff, void *workspace, void *stream) {
    return gn_single<float>(x, y, gamma, beta, N, HWC, C, G, eps, relu, out_cstride, out_coff, workspace, stream);
}

extern "C" int ml_groupnorm_chunk_f16(const void *x, void *y, const float *gamma, const float *beta, int32_t N,
                                      int64_t HWC, int32_t C, int32_t G, float eps, int32_t relu, int32_t out_cstride,
                                      int32_t out_coff, void *workspace, void *stream) {
    return gn_single<_Float16>(reinterpret_cast<const _Float16 *>(x), reinterpret_cast<_Float16 *>(y), gamma, beta, N, HWC, C,
                               G, eps, relu, out_cstride, out_coff, workspace, stream);
}
