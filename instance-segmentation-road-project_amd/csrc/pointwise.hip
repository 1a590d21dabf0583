// HBM-bound NHWC kernels: preprocess, depthwise 3x3, max-pool, bilinear (align_corners),
// global mean, channel scale, fill.  All are float4-vectorised over channels so a wave reads
// 1 KiB contiguous per instruction; none has inter-block reuse, so no LDS staging is used
// (re-reads of the 3x3 halo are served by L1/L2).
#include "common.h"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(long long n) { return (unsigned)((n + TPB - 1) / TPB); }

// ------------------------------------------------------------------ preprocess
struct Affine3 { float m[3], d[3], s[3]; };

__global__ void preprocess_kernel(const void *in, int is_u8, float *out, long long npix, int out_c,
                                  int flip, Affine3 a) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= npix) return;
    float v[3];
    if (is_u8) {
        const unsigned char *p = reinterpret_cast<const unsigned char *>(in) + i * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
        const float *p = reinterpret_cast<const float *>(in) + i * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    }
    if (flip) { const float t = v[0]; v[0] = v[2]; v[2] = t; }
    // reference order: (x - mean) / divisor (+ shift); a true division keeps x/127.5 bit-close
    const float r0 = (v[0] - a.m[0]) / a.d[0] + a.s[0];
    const float r1 = (v[1] - a.m[1]) / a.d[1] + a.s[1];
    const float r2 = (v[2] - a.m[2]) / a.d[2] + a.s[2];
    if (out_c == 4) {
        f32x4 o = {r0, r1, r2, 0.f};
        *reinterpret_cast<f32x4 *>(out + i * 4) = o;
    } else {
        out[i * 3 + 0] = r0; out[i * 3 + 1] = r1; out[i * 3 + 2] = r2;
    }
}

// ------------------------------------------------------------------ depthwise 3x3
__global__ void dwconv3x3_kernel(const float *__restrict__ in, const float *__restrict__ wgt,
                                 const float *__restrict__ bias, float *__restrict__ out,
                                 int H, int W, int C4, int in_cs, int in_co, int out_cs, int out_co,
                                 int Ho, int Wo, int stride, int dil, int pad_t, int pad_l, int act,
                                 long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    long long pix = idx / C4;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int c = c4 * 4;
    const int C = C4 * 4;
    f32x4 acc = bias ? *reinterpret_cast<const f32x4 *>(bias + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * stride - pad_t + kh * dil;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * stride - pad_l + kw * dil;
            if ((unsigned)ix >= (unsigned)W) continue;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(
                in + ((long long)(b * H + iy) * W + ix) * in_cs + in_co + c);
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wgt + (kh * 3 + kw) * C + c);
            acc += x * w;
        }
    }
    f32x4 o;
    o[0] = ml_apply_act(acc[0], act); o[1] = ml_apply_act(acc[1], act);
    o[2] = ml_apply_act(acc[2], act); o[3] = ml_apply_act(acc[3], act);
    *reinterpret_cast<f32x4 *>(out + ((long long)(b * Ho + oy) * Wo + ox) * out_cs + out_co + c) = o;
}

// ------------------------------------------------------------------ max pool 3x3 s2 (zero pad, input >= 0)
__global__ void maxpool3x3s2_kernel(const float *__restrict__ in, float *__restrict__ out,
                                    int H, int W, int C4, int Ho, int Wo, int pad_t, int pad_l,
                                    long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    long long pix = idx / C4;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int C = C4 * 4;
    bool any_pad = false;
    f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * 2 - pad_t + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * 2 - pad_l + kw;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) { any_pad = true; continue; }
            const f32x4 x = *reinterpret_cast<const f32x4 *>(in + ((long long)(b * H + iy) * W + ix) * C + c4 * 4);
            m[0] = fmaxf(m[0], x[0]); m[1] = fmaxf(m[1], x[1]);
            m[2] = fmaxf(m[2], x[2]); m[3] = fmaxf(m[3], x[3]);
        }
    }
    if (any_pad) {  // the explicit ZeroPadding2D contributes zeros to the window
        m[0] = fmaxf(m[0], 0.f); m[1] = fmaxf(m[1], 0.f); m[2] = fmaxf(m[2], 0.f); m[3] = fmaxf(m[3], 0.f);
    }
    *reinterpret_cast<f32x4 *>(out + ((long long)(b * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
}

// ------------------------------------------------------------------ bilinear, align_corners=True
__global__ void bilinear_ac_kernel(const float *__restrict__ in, const float *__restrict__ add,
                                   float *__restrict__ out, int H, int W, int C4, int in_cs, int in_co,
                                   int Ho, int Wo, float sy, float sx, int add_cs, int add_co,
                                   int out_cs, int out_co, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C4) * 4;
    long long pix = idx / C4;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float fy = (float)oy * sy;
    const float fx = (float)ox * sx;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), x0 = max((int)flx, 0);
    const int y1 = min((int)ceilf(fy), H - 1), x1 = min((int)ceilf(fx), W - 1);
    const float ty = fy - fly, tx = fx - flx;
    const float *base = in + (long long)b * H * W * in_cs + in_co + c;
    const f32x4 tl = *reinterpret_cast<const f32x4 *>(base + ((long long)y0 * W + x0) * in_cs);
    const f32x4 tr = *reinterpret_cast<const f32x4 *>(base + ((long long)y0 * W + x1) * in_cs);
    const f32x4 bl = *reinterpret_cast<const f32x4 *>(base + ((long long)y1 * W + x0) * in_cs);
    const f32x4 br = *reinterpret_cast<const f32x4 *>(base + ((long long)y1 * W + x1) * in_cs);
    const f32x4 top = tl + (tr - tl) * tx;
    const f32x4 bot = bl + (br - bl) * tx;
    f32x4 v = top + (bot - top) * ty;
    const long long opix = (long long)(b * Ho + oy) * Wo + ox;
    if (add) v += *reinterpret_cast<const f32x4 *>(add + opix * add_cs + add_co + c);
    *reinterpret_cast<f32x4 *>(out + opix * out_cs + out_co + c) = v;
}

// ------------------------------------------------------------------ global mean over HW
// grid (ceil(C4 / 16), B), block 256 = 16 row-groups x 16 channel-quads (256 contiguous bytes per row group); fp64
// accumulation, the 16 partial sums of a channel added in a fixed order.  (The first layout -- 64 quads x 4 row groups
// -- gave ASPP's pooling branch 64 blocks for 67 MB: 79 us.)
__global__ void global_mean_kernel(const float *__restrict__ in, float *__restrict__ out, int HW, int C4) {
    __shared__ double red[16][16][4];
    const int q = threadIdx.x & 15;
    const int g = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 16 + q;
    const int b = blockIdx.y;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (c4 < C4) {
        const float *p = in + (long long)b * HW * C4 * 4 + c4 * 4;
        for (int i = g; i < HW; i += 16) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(p + (long long)i * C4 * 4);
            s0 += x[0]; s1 += x[1]; s2 += x[2]; s3 += x[3];
        }
    }
    red[g][q][0] = s0; red[g][q][1] = s1; red[g][q][2] = s2; red[g][q][3] = s3;
    __syncthreads();
    if (g == 0 && c4 < C4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][q][e];
            o[e] = (float)(t / (double)HW);
        }
        *reinterpret_cast<f32x4 *>(out + (long long)b * C4 * 4 + c4 * 4) = o;
    }
}

__global__ void scale_channels_kernel(float *__restrict__ x, const float *__restrict__ s, int HW, int C4,
                                      long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int b = (int)(idx / ((long long)C4 * HW));
    f32x4 v = *reinterpret_cast<f32x4 *>(x + idx * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(s + ((long long)b * C4 + c4) * 4);
    *reinterpret_cast<f32x4 *>(x + idx * 4) = v * sc;
}

__global__ void add_kernel(float *__restrict__ x, const float *__restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = *reinterpret_cast<f32x4 *>(x + i * 4);
    a += *reinterpret_cast<const f32x4 *>(y + i * 4);
    *reinterpret_cast<f32x4 *>(x + i * 4) = a;
}

__global__ void fill_kernel(float *x, float v, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) x[i] = v;
}

}  // namespace

extern "C" int ml_preprocess_f32(const void *in, int32_t is_u8, float *out, int64_t npix, int32_t out_c,
                                 int32_t flip, const float *mean, const float *div, const float *shift,
                                 void *stream) {
    ML_REQUIRE(in && out && npix > 0 && mean && div && shift, "preprocess: bad arguments");
    ML_REQUIRE(out_c == 3 || out_c == 4, "preprocess: out_c must be 3 or 4");
    ML_REQUIRE(div[0] != 0.f && div[1] != 0.f && div[2] != 0.f, "preprocess: zero divisor");
    Affine3 a;
    for (int k = 0; k < 3; ++k) { a.m[k] = mean[k]; a.d[k] = div[k]; a.s[k] = shift[k]; }
    if (out_c == 4) ML_REQUIRE(ml_aligned16(out), "preprocess: out must be 16-byte aligned");
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(npix)), dim3(TPB), 0, (hipStream_t)stream, in, is_u8, out,
                       (long long)npix, out_c, flip, a);
    ML_CHECK_LAUNCH("preprocess");
    return ML_OK;
}

extern "C" int ml_dwconv3x3_f32(const float *in, const float *wgt, const float *bias, float *out, int32_t B,
                                int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                                int32_t out_cstride, int32_t out_coff, int32_t Ho, int32_t Wo, int32_t stride,
                                int32_t dil, int32_t pad_t, int32_t pad_l, int32_t act, void *stream) {
    ML_REQUIRE(in && wgt && out, "dwconv3x3: null pointer");
    ML_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "dwconv3x3: bad dims (C %% 4)");
    ML_REQUIRE(in_cstride % 4 == 0 && in_coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0,
               "dwconv3x3: channel strides/offsets must be multiples of 4");
    ML_REQUIRE(in_coff + C <= in_cstride && out_coff + C <= out_cstride, "dwconv3x3: slice exceeds buffer");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(wgt) && ml_aligned16(out) && (!bias || ml_aligned16(bias)),
               "dwconv3x3: pointers must be 16-byte aligned");
    ML_REQUIRE((long long)B * H * W < (1ll << 31) && stride > 0 && dil > 0, "dwconv3x3: geometry out of range");
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in, wgt, bias, out,
                       H, W, C / 4, in_cstride, in_coff, out_cstride, out_coff, Ho, Wo, stride, dil, pad_t, pad_l,
                       act, total);
    ML_CHECK_LAUNCH("dwconv3x3");
    return ML_OK;
}

extern "C" int ml_maxpool3x3s2_f32(const float *in, float *out, int32_t B, int32_t H, int32_t W, int32_t C,
                                   int32_t Ho, int32_t Wo, int32_t pad_t, int32_t pad_l, void *stream) {
    ML_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad arguments");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "maxpool: pointers must be 16-byte aligned");
    ML_REQUIRE((long long)B * H * W < (1ll << 31), "maxpool: too many pixels");
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in, out, H, W,
                       C / 4, Ho, Wo, pad_t, pad_l, total);
    ML_CHECK_LAUNCH("maxpool");
    return ML_OK;
}

extern "C" int ml_resize_bilinear_ac_f32(const float *in, const float *add, float *out, int32_t B, int32_t H,
                                         int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff, int32_t Ho,
                                         int32_t Wo, int32_t add_cstride, int32_t add_coff, int32_t out_cstride,
                                         int32_t out_coff, void *stream) {
    ML_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0,
               "resize_bilinear: bad arguments");
    ML_REQUIRE(in_cstride % 4 == 0 && in_coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0,
               "resize_bilinear: channel strides/offsets must be multiples of 4");
    ML_REQUIRE(in_coff + C <= in_cstride && out_coff + C <= out_cstride, "resize_bilinear: slice exceeds buffer");
    if (add) ML_REQUIRE(add_cstride % 4 == 0 && add_coff % 4 == 0 && add_coff + C <= add_cstride && ml_aligned16(add),
                        "resize_bilinear: bad add view");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "resize_bilinear: pointers must be 16-byte aligned");
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(bilinear_ac_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in, add, out, H, W,
                       C / 4, in_cstride, in_coff, Ho, Wo, sy, sx, add_cstride, add_coff, out_cstride, out_coff, total);
    ML_CHECK_LAUNCH("resize_bilinear");
    return ML_OK;
}

extern "C" int ml_global_mean_f32(const float *in, float *out, int32_t B, int32_t HW, int32_t C, void *stream) {
    ML_REQUIRE(in && out && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "global_mean: bad arguments");
    ML_REQUIRE(ml_aligned16(in) && ml_aligned16(out), "global_mean: pointers must be 16-byte aligned");
    const int C4 = C / 4;
    hipLaunchKernelGGL(global_mean_kernel, dim3((C4 + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, in, out, HW, C4);
    ML_CHECK_LAUNCH("global_mean");
    return ML_OK;
}

extern "C" int ml_scale_channels_f32(float *x, const float *s, int32_t B, int32_t HW, int32_t C, void *stream) {
    ML_REQUIRE(x && s && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "scale_channels: bad arguments");
    ML_REQUIRE(ml_aligned16(x) && ml_aligned16(s), "scale_channels: pointers must be 16-byte aligned");
    const long long total = (long long)B * HW * (C / 4);
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, x, s, HW, C / 4,
                       total);
    ML_CHECK_LAUNCH("scale_channels");
    return ML_OK;
}

extern "C" int ml_add_f32(float *x, const float *y, int64_t n, void *stream) {
    ML_REQUIRE(x && y && n > 0 && n % 4 == 0, "add: bad arguments (n %% 4)");
    ML_REQUIRE(ml_aligned16(x) && ml_aligned16(y), "add: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, (hipStream_t)stream, x, y, (long long)(n / 4));
    ML_CHECK_LAUNCH("add");
    return ML_OK;
}

extern "C" int ml_fill_f32(float *x, float v, int64_t n, void *stream) {
    ML_REQUIRE(x && n >= 0, "fill: bad arguments");
    if (n == 0) return ML_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, v, (long long)n);
    ML_CHECK_LAUNCH("fill");
    return ML_OK;
}
