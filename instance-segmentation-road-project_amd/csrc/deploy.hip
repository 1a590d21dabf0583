// The deploy wrapper either side of the MaskLab forward (reference engine/retinamasklab.py:598-643):
//   DownSampleInput (misc.py:143-154)   images (u8/f32, any C) -> bilinear(align_corners) resize
//   TrimInstances   (instance.py:258-277) drop -1 padded RoIs, pick each RoI's class channel, re-mold
//   SemanticSmoothing (semantic.py:270-285) grey opening (erosion then dilation, flat k x k element)
//   UpSampleOutput  (misc.py:169-196)   scale boxes -> int32, masks / semantic map > 0.5 -> int32
// All of it is HBM-bound byte/element work: one thread per output element, coalesced along the
// fastest (channel / x) axis; the only cross-thread step is the per-image RoI compaction (LDS scan).
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int TPB = 256;
inline unsigned grid_for(long long n) { return (unsigned)((n + TPB - 1) / TPB); }

// ------------------------------------------------------------------ bilinear (align_corners), any C
// TF resize_bilinear kernel: in = out_index * (in-1)/(out-1); lower = floor, upper = min(ceil, in-1);
// top = tl + (tr - tl) * x_lerp; out = top + (bottom - top) * y_lerp.   u8 sources are cast first.
template <bool U8>
__global__ void resize_any_kernel(const void *__restrict__ in_, float *__restrict__ out_f, int32_t *__restrict__ out_i,
                                  float thr, int H, int W, int C, int Ho, int Wo, float sy, float sx,
                                  long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long pix = idx / C;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), x0 = max((int)flx, 0);
    const int y1 = min((int)ceilf(fy), H - 1), x1 = min((int)ceilf(fx), W - 1);
    const float ty = fy - fly, tx = fx - flx;
    auto at = [&](int y, int x) -> float {
        const long long o = (((long long)b * H + y) * W + x) * C + c;
        return U8 ? (float)reinterpret_cast<const unsigned char *>(in_)[o] : reinterpret_cast<const float *>(in_)[o];
    };
    const float tl = at(y0, x0), tr = at(y0, x1), bl = at(y1, x0), br = at(y1, x1);
    const float top = tl + (tr - tl) * tx;
    const float bot = bl + (br - bl) * tx;
    const float v = top + (bot - top) * ty;
    if (out_f) out_f[idx] = v;
    if (out_i) out_i[idx] = v > thr ? 1 : 0;
}

// ------------------------------------------------------------------ TrimInstances
// one block per image: rows with class != -1 keep their order (tf.where is row-major), are moved to
// the front, and the tail is filled with -1 (MoldBatch).  Mask channel = the row's class id.
__global__ void __launch_bounds__(256)
trim_instances_kernel(const float *__restrict__ boxes, const float *__restrict__ masks, float *__restrict__ out_boxes,
                      float *__restrict__ out_masks, int32_t *__restrict__ counts, int N, int hw, int C) {
    extern __shared__ int src[];                 // [N] source row of each output row
    __shared__ int wave_tot[4], n_valid;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *rows = boxes + (long long)b * N * 6;
    int base = 0;
    for (int r0 = 0; r0 < N; r0 += 256) {        // ordered compaction, 256 rows per sweep
        const int r = r0 + tid;
        const bool valid = r < N && rows[r * 6 + 4] != -1.f;
        const unsigned long long m = __ballot(valid);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (valid) src[before + __popcll(m & ((1ull << lane) - 1))] = r;
        base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) { n_valid = base; counts[b] = base; }
    __syncthreads();
    const int n = n_valid;
    float *ob = out_boxes + (long long)b * N * 6;
    for (int i = tid; i < N * 6; i += 256) {
        const int p = i / 6, f = i - p * 6;
        ob[i] = p < n ? rows[src[p] * 6 + f] : -1.f;
    }
    float *om = out_masks + (long long)b * N * hw;
    const float *im = masks + (long long)b * N * hw * C;
    for (long long i = tid; i < (long long)N * hw; i += 256) {
        const int p = (int)(i / hw), e = (int)(i - (long long)p * hw);
        float v = -1.f;
        if (p < n) {
            const int r = src[p];
            const int cls = (int)rows[r * 6 + 4];
            // tf.gather_nd on an out-of-range class index is an error in the reference; clamp instead of faulting
            const int cc = min(max(cls, 0), C - 1);
            v = im[((long long)r * hw + e) * C + cc];
        }
        om[i] = v;
    }
}

// ------------------------------------------------------------------ UpSampleOutput pieces
// (cx, cy, w, h, label, conf) -> int32: cx, w scaled by ratio0 (the HEIGHT ratio, misc.py:180-183 --
// the reference's own axis mix-up is preserved), cy, h by ratio1; conf * 100.  tf.cast truncates.
__global__ void upsample_boxes_kernel(const float *__restrict__ rows, int32_t *__restrict__ out, long long n_rows,
                                      float ratio0, float ratio1) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= n_rows * 6) return;
    const int f = (int)(idx % 6);
    const float v = rows[idx];
    float r;
    if (f == 0 || f == 2) r = v * ratio0;
    else if (f == 1 || f == 3) r = v * ratio1;
    else if (f == 4) r = v;
    else r = v * 100.f;
    out[idx] = (int32_t)r;
}

__global__ void threshold_kernel(const float *__restrict__ in, int32_t *__restrict__ out, float thr, long long n) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx < n) out[idx] = in[idx] > thr ? 1 : 0;
}

// ------------------------------------------------------------------ SemanticSmoothing
// tf.nn.erosion2d / dilation2d with an all-zero k x k x C element, stride 1, SAME: a min / max over the
// window rows y - (k-1)/2 .. y - (k-1)/2 + k-1 (positions outside the map are ignored), separable
// into a row pass and a column pass.  Each channel has its own k (0 = pass through) and weight.
struct MorphArgs {
    int k[ML_SMOOTH_MAX_CLASSES];
    float weight[ML_SMOOTH_MAX_CLASSES];
};

template <bool IS_MAX, bool ALONG_X>
__global__ void morph_pass_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W, int C,
                                  MorphArgs a, int apply_weight, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long pix = idx / C;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const int k = a.k[c];
    float v = in[idx];
    if (k > 0) {
        const int pad = (k - 1) / 2;
        const int len = ALONG_X ? W : H;
        const int pos = ALONG_X ? x : y;
        const long long step = ALONG_X ? C : (long long)W * C;
        const int lo = max(pos - pad, 0), hi = min(pos - pad + k - 1, len - 1);
        const float *p = in + idx + (long long)(lo - pos) * step;
        v = *p;
        for (int i = lo + 1; i <= hi; ++i) {
            p += step;
            v = IS_MAX ? fmaxf(v, *p) : fminf(v, *p);
        }
    }
    if (apply_weight) v = v * a.weight[c];
    out[idx] = v;
}

}  // namespace

extern "C" int ml_resize_image_ac(const void *in, int32_t in_is_u8, float *out_f32, int32_t *out_i32, float threshold,
                                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream) {
    ML_REQUIRE(in && (out_f32 || out_i32), "resize_image: null pointer");
    ML_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "resize_image: bad dims");
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * C;
    if (in_is_u8)
        hipLaunchKernelGGL(resize_any_kernel<true>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in, out_f32,
                           out_i32, threshold, H, W, C, Ho, Wo, sy, sx, total);
    else
        hipLaunchKernelGGL(resize_any_kernel<false>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in,
                           out_f32, out_i32, threshold, H, W, C, Ho, Wo, sy, sx, total);
    ML_CHECK_LAUNCH("resize_image");
    return ML_OK;
}

extern "C" int ml_trim_instances_f32(const float *roi_boxes, const float *roi_masks, float *out_boxes, float *out_masks,
                                     int32_t *counts, int32_t B, int32_t N, int32_t mh, int32_t mw, int32_t C,
                                     void *stream) {
    ML_REQUIRE(roi_boxes && roi_masks && out_boxes && out_masks && counts, "trim_instances: null pointer");
    ML_REQUIRE(B > 0 && N > 0 && N <= 8192 && mh > 0 && mw > 0 && C > 0, "trim_instances: bad dims (N <= 8192)");
    hipLaunchKernelGGL(trim_instances_kernel, dim3(B), dim3(256), (size_t)N * 4, (hipStream_t)stream, roi_boxes, roi_masks,
                       out_boxes, out_masks, counts, N, mh * mw, C);
    ML_CHECK_LAUNCH("trim_instances");
    return ML_OK;
}

extern "C" int ml_upsample_boxes_i32(const float *rows, int32_t *out, int64_t n_rows, float ratio0, float ratio1,
                                     void *stream) {
    ML_REQUIRE(rows && out && n_rows >= 0, "upsample_boxes: bad arguments");
    if (n_rows == 0) return ML_OK;
    hipLaunchKernelGGL(upsample_boxes_kernel, dim3(grid_for(n_rows * 6)), dim3(TPB), 0, (hipStream_t)stream, rows, out,
                       (long long)n_rows, ratio0, ratio1);
    ML_CHECK_LAUNCH("upsample_boxes");
    return ML_OK;
}

extern "C" int ml_threshold_i32(const float *in, int32_t *out, float threshold, int64_t n, void *stream) {
    ML_REQUIRE(in && out && n >= 0, "threshold: bad arguments");
    if (n == 0) return ML_OK;
    hipLaunchKernelGGL(threshold_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, in, out, threshold,
                       (long long)n);
    ML_CHECK_LAUNCH("threshold");
    return ML_OK;
}

extern "C" int ml_semantic_smoothing_f32(const float *in, float *out, float *tmp, int32_t B, int32_t H, int32_t W,
                                         int32_t C, const int32_t *kernel_sizes, const float *weights, void *stream) {
    ML_REQUIRE(in && out && tmp && kernel_sizes && weights, "semantic_smoothing: null pointer");
    ML_REQUIRE(in != out && in != tmp && out != tmp, "semantic_smoothing: in, out and tmp must be distinct buffers");
    ML_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= ML_SMOOTH_MAX_CLASSES, "semantic_smoothing: bad dims (C <= %d)",
               ML_SMOOTH_MAX_CLASSES);
    MorphArgs a;
    for (int c = 0; c < ML_SMOOTH_MAX_CLASSES; ++c) {
        a.k[c] = c < C ? kernel_sizes[c] : 0;
        a.weight[c] = c < C ? weights[c] : 1.f;
        ML_REQUIRE(a.k[c] >= 0, "semantic_smoothing: negative kernel size");
    }
    const long long total = (long long)B * H * W * C;
    const dim3 g(grid_for(total)), t(TPB);
    hipStream_t s = (hipStream_t)stream;
    // erosion (min over rows, then columns), dilation (max over rows, then columns), weight last
    hipLaunchKernelGGL((morph_pass_kernel<false, true>), g, t, 0, s, in, tmp, H, W, C, a, 0, total);
    hipLaunchKernelGGL((morph_pass_kernel<false, false>), g, t, 0, s, (const float *)tmp, out, H, W, C, a, 0, total);
    hipLaunchKernelGGL((morph_pass_kernel<true, true>), g, t, 0, s, (const float *)out, tmp, H, W, C, a, 0, total);
    hipLaunchKernelGGL((morph_pass_kernel<true, false>), g, t, 0, s, (const float *)tmp, out, H, W, C, a, 1, total);
    ML_CHECK_LAUNCH("semantic_smoothing");
    return ML_OK;
}
