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

unsigned grid_of(long long total) { return (unsigned)((total + TPB - 1) / TPB); }

}  // namespace

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
