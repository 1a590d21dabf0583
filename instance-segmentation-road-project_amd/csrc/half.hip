// fp16-STORAGE helpers of the "fp16 MFMA path" (BASELINE config 5): the ResNeXt body keeps its activations in IEEE
// half between the convolutions; these are the byte-moving pieces around conv1x1_pipe.hip / gconv_mfma4.hip.
// All HBM-bound: 16-byte accesses per lane, one pass.
#include "common.h"

namespace {

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
constexpr int TPB = 256;

// ZeroPadding2D(1) + MaxPooling2D(3, 2) on a non-negative (post-ReLU) fp16 map: engine/backbone/ResNext.py:351-352
__global__ void maxpool3x3s2_h_kernel(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int H, int W, int C8,
                                      int Ho, int Wo, int pad_t, int pad_l, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    long long pix = idx / C8;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int C = C8 * 8;
    bool any_pad = false;
    f16x8h m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (_Float16)-65504.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * 2 - pad_t + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * 2 - pad_l + kw;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) { any_pad = true; continue; }
            const f16x8h x = *reinterpret_cast<const f16x8h *>(in + ((long long)(b * H + iy) * W + ix) * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
        }
    }
    if (any_pad) {  // the explicit ZeroPadding2D contributes zeros to the window
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = m[e] > (_Float16)0.f ? m[e] : (_Float16)0.f;
    }
    *reinterpret_cast<f16x8h *>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + c8 * 8) = m;
}

// out[b, oy, ox, :] = in[b, 2 oy, 2 ox, :]: the sampling of a 1x1 stride-2 convolution ('same' and 'valid' agree for
// k = 1), so that the strided shortcut convs (ResNext.py:199-203) run on the stride-1 pipelined kernel
__global__ void subsample2_h_kernel(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int H, int W, int C8,
                                    int Ho, int Wo, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    long long pix = idx / C8;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int C = C8 * 8;
    *reinterpret_cast<f16x8h *>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + c8 * 8) =
        *reinterpret_cast<const f16x8h *>(in + ((long long)(b * H + 2 * oy) * W + 2 * ox) * C + c8 * 8);
}

__global__ void cast_h2f_kernel(const _Float16 *__restrict__ in, float *__restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n8) return;
    const f16x8h x = *reinterpret_cast<const f16x8h *>(in + i * 8);
    f32x4 lo = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
    f32x4 hi = {(float)x[4], (float)x[5], (float)x[6], (float)x[7]};
    *reinterpret_cast<f32x4 *>(out + i * 8) = lo;
    *reinterpret_cast<f32x4 *>(out + i * 8 + 4) = hi;
}

// ---- the heads on half tensors (fp16 storage beyond the backbone body): same arithmetic as the fp32 kernels of
// pointwise.hip on the converted values, fp32 throughout, ONE rounding at the store
__device__ __forceinline__ void h8_to_f(const f16x8h x, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}
__device__ __forceinline__ f16x8h f_to_h8(const float (&v)[8]) {
    const f16x8h x = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3],
                      (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
    return x;
}

// tf.compat.v1.image.resize_bilinear(align_corners=True) (+ FPN Add / concat-slice store): engine/layers/misc.py:306
__global__ void bilinear_ac_h_kernel(const _Float16 *__restrict__ in, const _Float16 *__restrict__ add,
                                     _Float16 *__restrict__ out, int H, int W, int C8, int in_cs, int in_co, int Ho, int Wo,
                                     float sy, float sx, int add_cs, int add_co, int out_cs, int out_co, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    long long pix = idx / C8;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float fy = (float)oy * sy;
    const float fx = (float)ox * sx;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), x0 = max((int)flx, 0);
    const int y1 = min((int)ceilf(fy), H - 1), x1 = min((int)ceilf(fx), W - 1);
    const float ty = fy - fly, tx = fx - flx;
    const _Float16 *base = in + (long long)b * H * W * in_cs + in_co + c;
    float tl[8], tr[8], bl[8], br[8], v[8];
    h8_to_f(*reinterpret_cast<const f16x8h *>(base + ((long long)y0 * W + x0) * in_cs), tl);
    h8_to_f(*reinterpret_cast<const f16x8h *>(base + ((long long)y0 * W + x1) * in_cs), tr);
    h8_to_f(*reinterpret_cast<const f16x8h *>(base + ((long long)y1 * W + x0) * in_cs), bl);
    h8_to_f(*reinterpret_cast<const f16x8h *>(base + ((long long)y1 * W + x1) * in_cs), br);
    const long long opix = (long long)(b * Ho + oy) * Wo + ox;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float top = tl[e] + (tr[e] - tl[e]) * tx;
        const float bot = bl[e] + (br[e] - bl[e]) * tx;
        v[e] = top + (bot - top) * ty;
    }
    if (add) {
        float a[8];
        h8_to_f(*reinterpret_cast<const f16x8h *>(add + opix * add_cs + add_co + c), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a[e];
    }
    *reinterpret_cast<f16x8h *>(out + opix * out_cs + out_co + c) = f_to_h8(v);
}

// DepthwiseConv2D 3x3 (dilated): engine/layers/semantic.py:63-64; weights / bias fp32
__global__ void dwconv3x3_h_kernel(const _Float16 *__restrict__ in, const float *__restrict__ wgt,
                                   const float *__restrict__ bias, _Float16 *__restrict__ out, int H, int W, int C8, int in_cs,
                                   int in_co, int out_cs, int out_co, int Ho, int Wo, int stride, int dil, int pad_t,
                                   int pad_l, int act, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    long long pix = idx / C8;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int c = c8 * 8;
    const int C = C8 * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[c + e] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * stride - pad_t + kh * dil;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * stride - pad_l + kw * dil;
            if ((unsigned)ix >= (unsigned)W) continue;
            float x[8];
            h8_to_f(*reinterpret_cast<const f16x8h *>(in + ((long long)(b * H + iy) * W + ix) * in_cs + in_co + c), x);
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wgt + (kh * 3 + kw) * C + c);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wgt + (kh * 3 + kw) * C + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += x[e] * w0[e]; acc[4 + e] += x[4 + e] * w1[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = ml_apply_act(acc[e], act);
    *reinterpret_cast<f16x8h *>(out + ((long long)(b * Ho + oy) * Wo + ox) * out_cs + out_co + c) = f_to_h8(acc);
}

// tf.reduce_mean over H, W (semantic.py:149): grid (ceil(C8 / 8), B), block 256 = 32 row groups x 8 channel octets
// (128 contiguous bytes per row group); fp64 sums, the 32 partial sums of a channel added in a fixed order
__global__ void global_mean_h_kernel(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int HW, int C8) {
    __shared__ double red[32][8][8];
    const int q = threadIdx.x & 7;
    const int g = threadIdx.x >> 3;
    const int c8 = blockIdx.x * 8 + q;
    const int b = blockIdx.y;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < C8) {
        const _Float16 *p = in + (long long)b * HW * C8 * 8 + c8 * 8;
        for (int i = g; i < HW; i += 32) {
            float x[8];
            h8_to_f(*reinterpret_cast<const f16x8h *>(p + (long long)i * C8 * 8), x);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (double)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[g][q][e] = s[e];
    __syncthreads();
    if (g == 0 && c8 < C8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += red[k][q][e];
            o[e] = (float)(t / (double)HW);
        }
        *reinterpret_cast<f16x8h *>(out + (long long)b * C8 * 8 + c8 * 8) = f_to_h8(o);
    }
}

__global__ void cast_f2h_kernel(const float *__restrict__ in, _Float16 *__restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n8) return;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(in + i * 8);
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(in + i * 8 + 4);
    const f16x8h x = {(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3],
                      (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
    *reinterpret_cast<f16x8h *>(out + i * 8) = x;
}

unsigned grid_of(long long total) { return (unsigned)((total + TPB - 1) / TPB); }

}  // namespace

extern "C" int ml_resize_bilinear_ac_f16(const void *in, const void *add, void *out, int32_t B, int32_t H, int32_t W,
                                         int32_t C, int32_t in_cstride, int32_t in_coff, int32_t Ho, int32_t Wo,
                                         int32_t add_cstride, int32_t add_coff, int32_t out_cstride, int32_t out_coff,
                                         void *stream) {
    ML_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0,
               "resize_bilinear_f16: bad arguments (C %% 8)");
    ML_REQUIRE(in_cstride % 8 == 0 && in_coff % 8 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0,
               "resize_bilinear_f16: channel strides/offsets must be multiples of 8");
    ML_REQUIRE(in_coff + C <= in_cstride && out_coff + C <= out_cstride, "resize_bilinear_f16: slice exceeds buffer");
    if (add) ML_REQUIRE(add_cstride % 8 == 0 && add_coff % 8 == 0 && add_coff + C <= add_cstride && ml_aligned16(add),
                        "resize_bilinear_f16: bad add view");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "resize_bilinear_f16: pointers must be 16-byte aligned");
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(bilinear_ac_h_kernel, dim3(grid_of(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), reinterpret_cast<const _Float16 *>(add),
                       reinterpret_cast<_Float16 *>(out), H, W, C / 8, in_cstride, in_coff, Ho, Wo, sy, sx, add_cstride,
                       add_coff, out_cstride, out_coff, total);
    ML_CHECK_LAUNCH("resize_bilinear_f16");
    return ML_OK;
}

extern "C" int ml_dwconv3x3_f16(const void *in, const float *wgt, const float *bias, void *out, int32_t B, int32_t H,
                                int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff, int32_t out_cstride,
                                int32_t out_coff, int32_t Ho, int32_t Wo, int32_t stride, int32_t dil, int32_t pad_t,
                                int32_t pad_l, int32_t act, void *stream) {
    ML_REQUIRE(in && wgt && out, "dwconv3x3_f16: null pointer");
    ML_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "dwconv3x3_f16: bad dims (C %% 8)");
    ML_REQUIRE(in_cstride % 8 == 0 && in_coff % 8 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0,
               "dwconv3x3_f16: channel strides/offsets must be multiples of 8");
    ML_REQUIRE(in_coff + C <= in_cstride && out_coff + C <= out_cstride, "dwconv3x3_f16: slice exceeds buffer");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(wgt) && ml_aligned16(out), "dwconv3x3_f16: pointers must be 16-byte aligned");
    ML_REQUIRE((long long)B * H * W < (1ll << 31) && stride > 0 && dil > 0, "dwconv3x3_f16: geometry out of range");
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_h_kernel, dim3(grid_of(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), wgt, bias, reinterpret_cast<_Float16 *>(out), H, W, C / 8,
                       in_cstride, in_coff, out_cstride, out_coff, Ho, Wo, stride, dil, pad_t, pad_l, act, total);
    ML_CHECK_LAUNCH("dwconv3x3_f16");
    return ML_OK;
}

extern "C" int ml_global_mean_f16(const void *in, void *out, int32_t B, int32_t HW, int32_t C, void *stream) {
    ML_REQUIRE(in && out && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "global_mean_f16: bad arguments (C %% 8)");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "global_mean_f16: pointers must be 16-byte aligned");
    const int C8 = C / 8;
    hipLaunchKernelGGL(global_mean_h_kernel, dim3((C8 + 7) / 8, B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), reinterpret_cast<_Float16 *>(out), HW, C8);
    ML_CHECK_LAUNCH("global_mean_f16");
    return ML_OK;
}

extern "C" int ml_cast_f32_to_f16(const float *in, void *out, int64_t n, void *stream) {
    ML_REQUIRE(in && out && n > 0 && n % 8 == 0, "cast_f32_to_f16: n must be a positive multiple of 8");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "cast_f32_to_f16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(cast_f2h_kernel, dim3(grid_of(n / 8)), dim3(TPB), 0, (hipStream_t)stream, in,
                       reinterpret_cast<_Float16 *>(out), (long long)(n / 8));
    ML_CHECK_LAUNCH("cast_f32_to_f16");
    return ML_OK;
}

extern "C" int ml_maxpool3x3s2_f16(const void *in, void *out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
                                   int32_t Wo, int32_t pad_t, int32_t pad_l, void *stream) {
    ML_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_f16: bad arguments (C %% 8)");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "maxpool_f16: pointers must be 16-byte aligned");
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "maxpool_f16: too many pixels");
    const long long total = (long long)B * Ho * Wo * (C / 8);
    ML_REQUIRE(total < (1ll << 31) * TPB, "maxpool_f16: grid too large");
    hipLaunchKernelGGL(maxpool3x3s2_h_kernel, dim3(grid_of(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), reinterpret_cast<_Float16 *>(out), H, W, C / 8, Ho, Wo, pad_t,
                       pad_l, total);
    ML_CHECK_LAUNCH("maxpool_f16");
    return ML_OK;
}

extern "C" int ml_subsample2_f16(const void *in, void *out, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    ML_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "subsample2_f16: bad arguments (C %% 8)");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "subsample2_f16: pointers must be 16-byte aligned");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(subsample2_h_kernel, dim3(grid_of(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), reinterpret_cast<_Float16 *>(out), H, W, C / 8, Ho, Wo, total);
    ML_CHECK_LAUNCH("subsample2_f16");
    return ML_OK;
}

extern "C" int ml_cast_f16_to_f32(const void *in, float *out, int64_t n, void *stream) {
    ML_REQUIRE(in && out && n > 0 && n % 8 == 0, "cast_f16_to_f32: n must be a positive multiple of 8");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "cast_f16_to_f32: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(cast_h2f_kernel, dim3(grid_of(n / 8)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(in), out, (long long)(n / 8));
    ML_CHECK_LAUNCH("cast_f16_to_f32");
    return ML_OK;
}
