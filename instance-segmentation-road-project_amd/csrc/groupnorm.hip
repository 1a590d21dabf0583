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
#include "common.h"

namespace {

constexpr int GN_TPB = 256;
constexpr int GN_MAX_SPLIT = 64;
// Largest chunk the one-pass (register-resident) kernel takes.  Measured on MI355X: one block per chunk wins
// while the chunk is small (8x8 .. 32x32 maps, 14x14 RoI maps: 6.3 vs 9.8 us, 8.0 vs 10.2 us); from 64x64x128
// maps up (chunk 32 768 floats, 128 fat blocks) the sliced two-pass form has the parallelism and is faster
// (19.8 vs 23.7 us), even though it reads x twice.
constexpr int GN_ONEPASS_MAX = 4096;

struct GnPlan { int S; long long slice; };

// slices are multiples of 4*GN_TPB floats so every block runs whole float4 sweeps
static GnPlan gn_plan(long long L, int NG) {
    long long want = (2048 + NG - 1) / NG;            // aim at >= 2048 blocks on the chip
    if (want < 1) want = 1;
    if (want > GN_MAX_SPLIT) want = GN_MAX_SPLIT;
    const long long unit = 4 * GN_TPB;
    long long slice = ((L + want - 1) / want + unit - 1) / unit * unit;
    int S = (int)((L + slice - 1) / slice);
    return {S, slice};
}

template <bool VEC4 = true>
__device__ __forceinline__ void gn_stats_body(const float *__restrict__ x, double *__restrict__ ws, long long L,
                                              long long slice, int S, int ng, int s) {
    const float *p = x + (long long)ng * L;
    const long long lo = (long long)s * slice;
    const long long hi = min(lo + slice, L);
    double sum = 0.0, sq = 0.0;
    if (VEC4) {
        // a float4 is folded in fp32 first (3 adds, 4 fma), then joins the fp64 running sums: the kernel was
        // bound by its fp64 instruction count (12 per float4), not by HBM
        for (long long i = lo + threadIdx.x * 4; i < hi; i += GN_TPB * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + i);
            const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
            const float q4 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
            sum += (double)s4;
            sq += (double)q4;
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += GN_TPB) { const double d = p[i]; sum += d; sq += d * d; }
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

template <bool VEC4>
__global__ void gn_stats_kernel(const float *__restrict__ x, double *__restrict__ ws, long long L, long long slice,
                                int S) {
    gn_stats_body<VEC4>(x, ws, L, slice, S, blockIdx.y, blockIdx.x);
}

template <bool VEC4 = true>
__device__ __forceinline__ void gn_apply_body(const float *x, float *y, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, const double *__restrict__ ws, long long L,
                                              long long slice, int S, int C, int G, float eps, int relu, int out_cs,
                                              int out_co, int ng, int s) {
    const int g = ng % G;
    const int cg = C / G;
    double sum = 0, sq = 0;
    for (int i = 0; i < S; ++i) {  // uniform, L2-resident, S <= 64
        sum += ws[((long long)ng * S + i) * 2 + 0];
        sq += ws[((long long)ng * S + i) * 2 + 1];
    }
    const double meand = sum / (double)L;
    double vard = sq / (double)L - meand * meand;
    if (vard < 0) vard = 0;
    const float mean = (float)meand;
    const float rstd = (float)(1.0 / sqrt(vard + (double)eps));
    const float *p = x + (long long)ng * L;
    const bool dense = (out_cs == C);
    // dense: y has x's layout.  sliced: element f of the sample -> y[n][f / C][out_co + f % C]
    const long long HWC = L * G;
    float *q = dense ? y + (long long)ng * L : y + (long long)(ng / G) * (HWC / C) * out_cs + out_co;
    const long long lo = (long long)s * slice;
    const long long hi = min(lo + slice, L);
    // flat index inside the sample = g*L + i ; channel = that mod C
    const long long gbase = (long long)g * L;
    if (VEC4 && (cg % 4 == 0) && HWC < (1ll << 31)) {
        // Hot form: a float4 never straddles a gamma period (cg % 4 == 0) and the flat index fits 32 bits, so
        // channel / pixel / gamma offset advance incrementally -- no division or modulo inside the loop (the
        // 64-bit `%` and `/` per float4 made this pass VALU-bound)
        const unsigned Cu = (unsigned)C, cgu = (unsigned)cg;
        const long long i0 = lo + threadIdx.x * 4;
        const unsigned f0 = (unsigned)(gbase + i0);
        unsigned c0 = f0 % Cu, pix = f0 / Cu, cm = c0 % cgu;
        const unsigned stepc = (GN_TPB * 4u) % Cu, steppix = (GN_TPB * 4u) / Cu, stepm = (GN_TPB * 4u) % cgu;
        const float *gp = gamma ? gamma + g * cg : nullptr;
        const float *bp = beta ? beta + g * cg : nullptr;
        for (long long i = i0; i < hi; i += GN_TPB * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd;
            if (gp) o *= *reinterpret_cast<const f32x4 *>(gp + cm);
            if (bp) o += *reinterpret_cast<const f32x4 *>(bp + cm);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            if (dense) *reinterpret_cast<f32x4 *>(q + i) = o;
            else *reinterpret_cast<f32x4 *>(q + (long long)pix * out_cs + c0) = o;
            c0 += stepc;
            pix += steppix;
            if (c0 >= Cu) { c0 -= Cu; ++pix; }
            cm += stepm;
            if (cm >= cgu) cm -= cgu;
        }
    } else if (VEC4) {
        for (long long i = lo + threadIdx.x * 4; i < hi; i += GN_TPB * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + i);
            const int c0 = (int)((gbase + i) % C);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int c = c0 + e;
                if (c >= C) c -= C;
                const int j = g * cg + (c % cg);
                float t = (v[e] - mean) * rstd;
                if (gamma) t *= gamma[j];
                if (beta) t += beta[j];
                o[e] = relu ? fmaxf(t, 0.f) : t;
            }
            if (dense) *reinterpret_cast<f32x4 *>(q + i) = o;
            else *reinterpret_cast<f32x4 *>(q + ((gbase + i) / C) * out_cs + c0) = o;
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += GN_TPB) {
            const int c = (int)((gbase + i) % C);
            const int j = g * cg + (c % cg);
            float t = (p[i] - mean) * rstd;
            if (gamma) t *= gamma[j];
            if (beta) t += beta[j];
            const float r = relu ? fmaxf(t, 0.f) : t;
            if (dense) q[i] = r;
            else q[((gbase + i) / C) * out_cs + c] = r;
        }
    }
}

template <bool VEC4>
__global__ void gn_apply_kernel(const float *x, float *y, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const double *__restrict__ ws, long long L,
                                long long slice, int S, int C, int G, float eps, int relu, int out_cs, int out_co) {
    gn_apply_body<VEC4>(x, y, gamma, beta, ws, L, slice, S, C, G, eps, relu, out_cs, out_co, blockIdx.y, blockIdx.x);
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

template <int TPB, int VPT>
__device__ __forceinline__ void gn_onepass_body(const float *__restrict__ x, float *__restrict__ y,
                                                const float *__restrict__ gamma, const float *__restrict__ beta, int L,
                                                int C, int G, float eps, int relu, int out_cs, int out_co, int ng) {
    __shared__ double red[TPB / 64];
    const int g = ng % G;
    const int cg = C / G;
    const float *p = x + (long long)ng * L;
    f32x4 v[VPT];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * 4;
        v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < L) {
            v[k] = *reinterpret_cast<const f32x4 *>(p + i);
            sum += ((double)v[k][0] + (double)v[k][1]) + ((double)v[k][2] + (double)v[k][3]);
        }
    }
    const double meand = block_sum<TPB>(sum, red) / (double)L;
    const float mean = (float)meand;
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * 4;
        if (i < L) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[k][e] - meand; sq += d * d; }
        }
    }
    const double vard = block_sum<TPB>(sq, red) / (double)L;
    const float rstd = (float)(1.0 / sqrt(vard + (double)eps));
    const bool dense = (out_cs == C);
    const long long HWC = (long long)L * G;
    float *q = dense ? y + (long long)ng * L : y + (long long)(ng / G) * (HWC / C) * out_cs + out_co;
    const long long gbase = (long long)g * L;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = (k * TPB + threadIdx.x) * 4;
        if (i >= L) continue;
        const int c0 = (int)((gbase + i) % C);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = c0 + e;
            if (c >= C) c -= C;
            const int j = g * cg + (c % cg);
            float t = (v[k][e] - mean) * rstd;
            if (gamma) t *= gamma[j];
            if (beta) t += beta[j];
            o[e] = relu ? fmaxf(t, 0.f) : t;
        }
        if (dense) *reinterpret_cast<f32x4 *>(q + i) = o;
        else *reinterpret_cast<f32x4 *>(q + ((gbase + i) / C) * out_cs + c0) = o;
    }
}

template <int TPB, int VPT>
__global__ void __launch_bounds__(TPB)
gn_onepass_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ gamma,
                  const float *__restrict__ beta, int L, int C, int G, float eps, int relu, int out_cs, int out_co) {
    gn_onepass_body<TPB, VPT>(x, y, gamma, beta, L, C, G, eps, relu, out_cs, out_co, blockIdx.x);
}

template <int TPB, int VPT>
void launch_onepass(const float *x, float *y, const float *gamma, const float *beta, int NG, int L, int C, int G,
                    float eps, int relu, int out_cs, int out_co, hipStream_t s) {
    hipLaunchKernelGGL((gn_onepass_kernel<TPB, VPT>), dim3(NG), dim3(TPB), 0, s, x, y, gamma, beta, L, C, G, eps, relu,
                       out_cs, out_co);
}


// ------------------------------------------------------------------ several problems in one launch pair
// The un-shared towers normalise five pyramid levels (and the mask head three RoI levels) one after the other:
// the two coarsest levels are a few thousand floats each, i.e. a ~10 us launch apiece behind the 46 us P3 launch.
// ml_groupnorm_multi_f32 runs all of them as (1) one statistics launch over the problems with large chunks and
// (2) one launch that applies those AND normalises the small-chunk problems in their register-resident one-pass
// form.  Same per-problem arithmetic as the single-problem kernels (bit-identical results).
struct GnProb {
    const float *x;
    float *y;
    const float *gamma, *beta;
    double *ws;               // partials of this problem (two-pass only)
    long long L, slice;
    int NG, S, C, G, relu, out_cs, out_co, onepass_vpt;   // onepass_vpt: 0 = two-pass, else float4 per thread
    float eps;
};
struct GnMulti {
    int n;
    int start[ML_GN_MAX_PROBLEMS + 1];
    GnProb p[ML_GN_MAX_PROBLEMS];
};

__global__ void __launch_bounds__(GN_TPB)
gn_multi_stats_kernel(const GnMulti A) {
    int pi = 0;
    while (pi + 1 < A.n && (int)blockIdx.x >= A.start[pi + 1]) ++pi;
    const GnProb &P = A.p[pi];
    const int id = blockIdx.x - A.start[pi];
    gn_stats_body(P.x, P.ws, P.L, P.slice, P.S, id / P.S, id % P.S);
}

__global__ void __launch_bounds__(GN_TPB)
gn_multi_apply_kernel(const GnMulti A) {
    int pi = 0;
    while (pi + 1 < A.n && (int)blockIdx.x >= A.start[pi + 1]) ++pi;
    const GnProb &P = A.p[pi];
    const int id = blockIdx.x - A.start[pi];
    if (P.onepass_vpt == 0) {
        gn_apply_body(P.x, P.y, P.gamma, P.beta, P.ws, P.L, P.slice, P.S, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co,
                      id / P.S, id % P.S);
    } else if (P.onepass_vpt == 1) {
        gn_onepass_body<GN_TPB, 1>(P.x, P.y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    } else if (P.onepass_vpt == 2) {
        gn_onepass_body<GN_TPB, 2>(P.x, P.y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    } else {
        gn_onepass_body<GN_TPB, 4>(P.x, P.y, P.gamma, P.beta, (int)P.L, P.C, P.G, P.eps, P.relu, P.out_cs, P.out_co, id);
    }
}

}  // namespace

extern "C" int64_t ml_groupnorm_workspace_bytes(int32_t N, int32_t G) {
    return (int64_t)N * G * GN_MAX_SPLIT * 2 * (int64_t)sizeof(double);
}

static int gn_validate(const float *x, float *y, int32_t N, int64_t HWC, int32_t C, int32_t G, int32_t out_cstride,
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

extern "C" int ml_groupnorm_multi_f32(const ml_gn_desc *descs, int32_t n, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
    ML_REQUIRE(descs && n >= 1 && n <= ML_GN_MAX_PROBLEMS && workspace, "groupnorm_multi: need 1..%d problems and a workspace",
               ML_GN_MAX_PROBLEMS);
    GnMulti st, ap;
    st.n = 0;
    ap.n = n;
    long long sb = 0, ab = 0, ws_off = 0;
    for (int i = 0; i < n; ++i) {
        const ml_gn_desc &d = descs[i];
        if (int rc = gn_validate(d.x, d.y, d.N, d.HWC, d.C, d.G, d.out_cstride, d.out_coff)) return rc;
        const long long L = d.HWC / d.G;
        const bool vec4 = (L % 4 == 0) && (d.C % 4 == 0) && (d.out_cstride % 4 == 0) && (d.out_coff % 4 == 0) &&
                          ml_aligned16(d.x) && ml_aligned16(d.y);
        ML_REQUIRE(vec4, "groupnorm_multi: problem %d needs 16-byte aligned tensors and chunk / channel counts that are "
                   "multiples of 4 (use ml_groupnorm_chunk_f32 otherwise)", i);
        GnProb P;
        P.x = d.x; P.y = d.y; P.gamma = d.gamma; P.beta = d.beta;
        P.L = L; P.NG = d.N * d.G; P.C = d.C; P.G = d.G; P.relu = d.relu; P.out_cs = d.out_cstride; P.out_co = d.out_coff;
        P.eps = d.eps;
        P.ws = nullptr; P.S = 1; P.slice = L; P.onepass_vpt = 0;
        if (L <= GN_ONEPASS_MAX) {
            const int v4 = (int)((L + 3) / 4);
            P.onepass_vpt = v4 <= 256 ? 1 : (v4 <= 512 ? 2 : 4);
            ap.start[i] = (int)ab;
            ab += P.NG;
        } else {
            const GnPlan plan = gn_plan(L, P.NG);
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
    if (st.n > 0) hipLaunchKernelGGL(gn_multi_stats_kernel, dim3((unsigned)sb), dim3(GN_TPB), 0, s, st);
    hipLaunchKernelGGL(gn_multi_apply_kernel, dim3((unsigned)ab), dim3(GN_TPB), 0, s, ap);
    ML_CHECK_LAUNCH("groupnorm_multi");
    return ML_OK;
}

extern "C" int ml_groupnorm_chunk_f32(const float *x, float *y, const float *gamma, const float *beta, int32_t N,
                                      int64_t HWC, int32_t C, int32_t G, float eps, int32_t relu, int32_t out_cstride,
                                      int32_t out_coff, void *workspace, void *stream) {
    ML_REQUIRE(x && y && workspace, "groupnorm: null pointer");
    ML_REQUIRE(N > 0 && HWC > 0 && C > 0 && G > 0, "groupnorm: bad dims");
    ML_REQUIRE(C >= G, "groupnorm: Number of groups (%d) cannot be more than the number of channels (%d).", G, C);
    ML_REQUIRE(C % G == 0, "groupnorm: Number of groups (%d) must be a multiple of the number of channels (%d).", G, C);
    ML_REQUIRE(HWC % C == 0 && HWC % G == 0, "groupnorm: H*W*C (%lld) must be divisible by C and by G", (long long)HWC);
    ML_REQUIRE((long long)N * G < 65536, "groupnorm: N*G too large for grid.y");
    const long long L = HWC / G;
    const GnPlan plan = gn_plan(L, N * G);
    ML_REQUIRE(out_cstride >= C && out_coff >= 0 && out_coff + C <= out_cstride, "groupnorm: bad output slice");
    ML_REQUIRE(out_cstride == C ? out_coff == 0 : true, "groupnorm: dense output must have out_coff 0");
    const bool vec4 = (L % 4 == 0) && (C % 4 == 0) && (out_cstride % 4 == 0) && (out_coff % 4 == 0) &&
                      ml_aligned16(x) && ml_aligned16(y);
    hipStream_t s = (hipStream_t)stream;
    double *ws = reinterpret_cast<double *>(workspace);
    const dim3 grid(plan.S, N * G);
    if (vec4 && L <= GN_ONEPASS_MAX) {
        // one pass, chunk in registers: float4-per-thread chosen so that 256 * VPT * 4 >= L
        const int NG = N * G, Li = (int)L;
        const int v4 = (Li + 3) / 4;
#define GN1(TPB, VPT) launch_onepass<TPB, VPT>(x, y, gamma, beta, NG, Li, C, G, eps, relu, out_cstride, out_coff, s)
        if (v4 <= 256) GN1(256, 1);
        else if (v4 <= 512) GN1(256, 2);
        else GN1(256, 4);
#undef GN1
    } else if (vec4) {
        hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(GN_TPB), 0, s, x, ws, L, plan.slice, plan.S);
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(GN_TPB), 0, s, x, y, gamma, beta, ws, L, plan.slice, plan.S, C,
                           G, eps, relu, out_cstride, out_coff);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(GN_TPB), 0, s, x, ws, L, plan.slice, plan.S);
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(GN_TPB), 0, s, x, y, gamma, beta, ws, L, plan.slice, plan.S,
                           C, G, eps, relu, out_cstride, out_coff);
    }
    ML_CHECK_LAUNCH("groupnorm");
    return ML_OK;
}
