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

// =====================================================================================================
// Serving post-processing (reference engine/layers/misc.py:358-401, 554-718,
// assembled in road_project/setup/serving.py:28-50): CropAndPadMask, CrackToInstance, CalculateInstanceSize,
// IncludeMyRoad -- the numeric half of the serving graph (JPEG decode / drawing / encode are not arithmetic
// on this path and stay outside).  All of it is byte / element streaming plus small reductions.
// =====================================================================================================
namespace {

// ---- CropAndPadMask: threshold = max(conf) > 50 ? 50 : -100 (misc.py:371-374), one block
__global__ void __launch_bounds__(256) conf_threshold_kernel(const int32_t *__restrict__ det, int rows, int32_t *thr) {
    __shared__ int red[256];
    int m = INT32_MIN;
    for (int i = threadIdx.x; i < rows; i += 256) m = max(m, det[i * 6 + 5]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *thr = red[0] > 50 ? 50 : -100;
}

// Where CropAndPadMask pastes an instance (misc.py:377-398): box = max(box, 1) elementwise, corners ceil(c -+ size/2)
// clipped to the canvas; rows below the confidence threshold and zero-sized boxes paste nothing (empty box).
struct PasteBox {
    int xmin, xmax, ymin, ymax;      // [xmin, xmax) x [ymin, ymax); empty when nothing is pasted
    float sy, sx;                    // mask rows / columns per canvas row / column (align_corners)
};
__device__ __forceinline__ PasteBox paste_box(const int32_t *d, int thr, int mh, int mw, int H, int W) {
    PasteBox p = {0, 0, 0, 0, 0.f, 0.f};
    if (d[5] < thr) return p;
    const float cx = (float)max(d[0], 1), cy = (float)max(d[1], 1), w = (float)max(d[2], 1), h = (float)max(d[3], 1);
    p.xmin = min(max((int)ceilf(cx - w / 2.f), 0), W); p.xmax = min(max((int)ceilf(cx + w / 2.f), 0), W);
    p.ymin = min(max((int)ceilf(cy - h / 2.f), 0), H); p.ymax = min(max((int)ceilf(cy + h / 2.f), 0), H);
    const int oh = p.ymax - p.ymin, ow = p.xmax - p.xmin;
    p.sy = oh > 1 ? (float)(mh - 1) / (float)(oh - 1) : 0.f;
    p.sx = ow > 1 ? (float)(mw - 1) / (float)(ow - 1) : 0.f;
    if (oh <= 0 || ow <= 0) p.xmax = p.xmin = p.ymax = p.ymin = 0;
    return p;
}
// value of canvas pixel (y, x): the mh x mw int mask resized bilinear (align_corners) to the box; 0 outside it
__device__ __forceinline__ float paste_value(const PasteBox &p, const int32_t *m, int mh, int mw, int y, int x) {
    if (!(y >= p.ymin && y < p.ymax && x >= p.xmin && x < p.xmax)) return 0.f;
    const float fy = (float)(y - p.ymin) * p.sy, fx = (float)(x - p.xmin) * p.sx;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), x0 = max((int)flx, 0);
    const int y1 = min((int)ceilf(fy), mh - 1), x1 = min((int)ceilf(fx), mw - 1);
    const float ty = fy - fly, tx = fx - flx;
    const float tl = (float)m[y0 * mw + x0], tr = (float)m[y0 * mw + x1];
    const float bl = (float)m[y1 * mw + x0], br = (float)m[y1 * mw + x1];
    const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
    return top + (bot - top) * ty;
}

// out[b,i,y,x]: the mh x mw mask of a selected instance resized (bilinear, align_corners) to its box and
// placed at (ymin, xmin) of an H x W canvas of zeros (misc.py:377-398).  One thread per canvas pixel.
__global__ void crop_pad_mask_kernel(const int32_t *__restrict__ det, const int32_t *__restrict__ masks,
                                     const int32_t *__restrict__ thr, float *__restrict__ out, int n, int mh, int mw,
                                     int H, int W, long long total) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H);
    const long long row = t / H;                 // b * n + i
    const PasteBox p = paste_box(det + row * 6, *thr, mh, mw, H, W);
    out[idx] = paste_value(p, masks + row * mh * mw, mh, mw, y, x);
}

// ---- CrackToInstance (misc.py:524-551): bounding box of the non-zero pixels of a [B,H,W] int32 map over the
// WHOLE batch (the reference reduces tf.where over all rows).  box = {ymin, xmin, ymax, xmax, any}
constexpr int BBOX_EPT = 8;                  // elements per thread
__global__ void __launch_bounds__(TPB) nonzero_bbox_kernel(const int32_t *__restrict__ map, int cstride, int coff, int H, int W,
                                                           long long total, int32_t *__restrict__ box) {
    // (a map that is mostly non-zero -- the crack channel of a random-weight model -- used to mean one atomic quartet per
    // pixel on four addresses: 2.8 ms at 8 x 1024^2; now one per block of 2048 pixels)
    __shared__ int red[4][TPB / 64];
    int y0 = 0x7fffffff, x0 = 0x7fffffff, y1 = -1, x1 = -1;
    const long long base = (long long)blockIdx.x * (TPB * BBOX_EPT) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < BBOX_EPT; ++k) {
        const long long idx = base + (long long)k * TPB;
        if (idx < total && map[idx * cstride + coff] != 0) {
            const int x = (int)(idx % W);
            const int y = (int)((idx / W) % H);
            y0 = min(y0, y); x0 = min(x0, x); y1 = max(y1, y); x1 = max(x1, x);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        y0 = min(y0, __shfl_down(y0, off, 64)); x0 = min(x0, __shfl_down(x0, off, 64));
        y1 = max(y1, __shfl_down(y1, off, 64)); x1 = max(x1, __shfl_down(x1, off, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = y0; red[1][wave] = x0; red[2][wave] = y1; red[3][wave] = x1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) {
            y0 = min(y0, red[0][w]); x0 = min(x0, red[1][w]); y1 = max(y1, red[2][w]); x1 = max(x1, red[3][w]);
        }
        if (y1 >= 0) {
            atomicMin(&box[0], y0);
            atomicMin(&box[1], x0);
            atomicMax(&box[2], y1);
            atomicMax(&box[3], x1);
            box[4] = 1;
        }
    }
}

// ---- CalculateInstanceSize (misc.py:632-718)
// per image row: min / max x of the road pixels (tf.segment_min / segment_max over tf.where, :683-684);
// rows without road pixels read 0 / 0 like an empty TF segment.  grid (H, B), block 64.
__global__ void __launch_bounds__(64) road_row_extent_kernel(const int32_t *__restrict__ seg, int cstride, int coff, int H,
                                                             int W, int32_t *__restrict__ xmin, int32_t *__restrict__ xmax,
                                                             int32_t *__restrict__ last_row) {
    const int y = blockIdx.x, b = blockIdx.y;
    const int32_t *row = seg + ((long long)(b * H + y) * W) * cstride + coff;
    int lo = INT32_MAX, hi = -1;
    for (int x = threadIdx.x; x < W; x += 64)
        if (row[(long long)x * cstride] > 0) { lo = min(lo, x); hi = max(hi, x); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_down(lo, off, 64));
        hi = max(hi, __shfl_down(hi, off, 64));
    }
    if (threadIdx.x == 0) {
        xmin[b * H + y] = hi >= 0 ? lo : 0;
        xmax[b * H + y] = hi >= 0 ? hi : 0;
        if (hi >= 0) atomicMax(&last_row[b], y);          // segment ids run 0 .. max(y) (:683-685)
    }
}

// 2x2 inverse through LU with partial pivoting in float32 (what Eigen / LAPACK do for tf.linalg.inv / det)
__device__ void theta_from_sums(double syy, double sy, double cnt, double syx, double sx, float *theta) {
    const float a = (float)syy, b = (float)sy, c = (float)sy, d = (float)cnt;     // X^T X = [[a b] [c d]]
    const float r0 = (float)syx, r1 = (float)sx;                                  // X^T y
    theta[0] = theta[1] = 0.f;
    // LU, pivot on the larger of |a|, |c|
    float p00, p01, p10, p11, q0, q1;
    float sign = 1.f;
    if (fabsf(c) > fabsf(a)) { p00 = c; p01 = d; p10 = a; p11 = b; q0 = r1; q1 = r0; sign = -1.f; }
    else { p00 = a; p01 = b; p10 = c; p11 = d; q0 = r0; q1 = r1; }
    if (p00 == 0.f) return;
    const float l = p10 / p00;
    const float u11 = p11 - l * p01;
    const float det = sign * p00 * u11;
    if (!(det > 0.f)) return;                                                     // tf.cond(det_x > 0, ..., zeros) (:708-711)
    const float z1 = q1 - l * q0;
    const float t1 = z1 / u11;
    const float t0 = (q0 - p01 * t1) / p00;
    theta[0] = t0;
    theta[1] = t1;
}

// one block per image: marginal points (rows with x_min != x_max), 15 % dropped at both ends, least squares of
// x on y for the left and the right edge, unit[b,y] = default_road_size / clip(right(y) - left(y), 1, inf)
__global__ void __launch_bounds__(256) road_unit_length_kernel(const int32_t *__restrict__ xmin, const int32_t *__restrict__ xmax,
                                                               const int32_t *__restrict__ last_row, int H,
                                                               float default_road_size, float *__restrict__ unit) {
    __shared__ int wtot[4], s_valid;
    __shared__ double red[6][4];
    __shared__ float th[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = last_row[b] + 1;            // 0 when the image has no road pixel
    const int32_t *lo = xmin + b * H, *hi = xmax + b * H;
    // pass 1: number of valid rows
    int cnt = 0;
    for (int y = tid; y < rows; y += 256) cnt += lo[y] != hi[y];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0) wtot[wave] = cnt;
    __syncthreads();
    if (tid == 0) s_valid = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    const int valid = s_valid;
    int drop = (int)((float)valid * 0.15f);      // tf.cast(valid_counts * 0.15, int32), clipped to >= 1 (:695-697)
    if (drop < 1) drop = 1;
    // pass 2: ordered rank of every valid row, sums over ranks [drop, valid - drop)
    double syy = 0, sy = 0, n = 0, sxl = 0, syxl = 0, sxr = 0, syxr = 0;
    int base = 0;
    for (int y0 = 0; y0 < rows; y0 += 256) {
        const int y = y0 + tid;
        const bool v = y < rows && lo[y] != hi[y];
        const unsigned long long m = __ballot(v);
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        const int rank = before + __popcll(m & ((1ull << lane) - 1));
        if (v && rank >= drop && rank < valid - drop) {
            const double yy = (double)y;
            syy += yy * yy; sy += yy; n += 1.0;
            sxl += (double)lo[y]; syxl += yy * (double)lo[y];
            sxr += (double)hi[y]; syxr += yy * (double)hi[y];
        }
        base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    double vals[7] = {syy, sy, n, sxl, syxl, sxr, syxr};
    double tot[7];
    for (int k = 0; k < 7; ++k) {
        double v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[k % 6][wave] = v;
        __syncthreads();
        tot[k] = red[k % 6][0] + red[k % 6][1] + red[k % 6][2] + red[k % 6][3];
        __syncthreads();
    }
    if (tid == 0) {
        theta_from_sums(tot[0], tot[1], tot[2], tot[4], tot[3], th);          // left edge
        theta_from_sums(tot[0], tot[1], tot[2], tot[6], tot[5], th + 2);      // right edge
    }
    __syncthreads();
    for (int y = tid; y < H; y += 256) {
        const float fy = (float)y;
        const float left = fy * th[0] + th[1], right = fy * th[2] + th[3];
        const float wdt = fmaxf(right - left, 1.f);
        unit[b * H + y] = default_road_size / wdt;
    }
}

// Mask source of the summary kernels: the padded canvas CropAndPadMask wrote (ROI = false: [B,n,H,W] floats, read in
// full), or the un-pasted instances themselves (ROI = true: detections + mh x mw int masks + the confidence threshold):
// the canvas value is recomputed on the fly (paste_value: the same arithmetic, bit for bit) and only the rows / column
// stripes that meet the box are visited -- every row keeps its wave and every pixel its lane, and what is skipped is
// exactly zero, so the sums are bit-identical while 3.4 GB (8 x 100 canvases of 1024^2) are never written or read.
struct RoiSrc {
    const int32_t *det, *masks, *thr;
    int mh, mw;
};

// per (instance, image): rows are dealt to the 4 waves, lanes sweep x.  out5 = {pixel sum, instance size,
// (horizontal: second kernel), vertical size, include_my_road}
template <bool ROI>
__global__ void __launch_bounds__(256) instance_rows_kernel(const float *__restrict__ masks, RoiSrc R,
                                                            const int32_t *__restrict__ seg,
                                                            int cstride, int road_coff, const float *__restrict__ unit, int n,
                                                            int H, int W, float ioi_threshold, float *__restrict__ out5,
                                                            int S, double *__restrict__ part) {
    __shared__ double red[5][4];
    const int i = blockIdx.x, b = blockIdx.y;
    // S > 1 (few instances: the crack pseudo-instance is ONE canvas per image): blockIdx.z owns a band of image rows --
    // a multiple of 4, so every row keeps its wave -- and leaves its five partial sums for instance_rows_finish_kernel
    const int band = ((H + S - 1) / S + 3) / 4 * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *m = ROI ? nullptr : masks + ((long long)(b * n + i) * H) * W;
    const int32_t *road = seg + ((long long)b * H * W) * cstride + road_coff;
    PasteBox pb = {0, W, 0, H, 0.f, 0.f};
    const int32_t *mi = nullptr;
    if (ROI) {
        pb = paste_box(R.det + (long long)(b * n + i) * 6, *R.thr, R.mh, R.mw, H, W);
        mi = R.masks + (long long)(b * n + i) * R.mh * R.mw;
    }
    // first row of this wave inside the box (and the band), first column stripe that meets it (ROI = false: everything)
    const int ylo = max(pb.ymin, (int)blockIdx.z * band), yhi = min(pb.ymax, ((int)blockIdx.z + 1) * band);
    const int y_first = ylo + ((wave - ylo) % 4 + 4) % 4;
    const int x_first = (pb.xmin / 64) * 64 + lane;
    double pix = 0, size = 0, vert = 0, inter = 0, area = 0;
    for (int y = y_first; y < yhi; y += 4) {
        const float u = unit[b * H + y];
        float rs = 0.f;
        int rin = 0, rar = 0;
        bool any = false;
        for (int x = x_first; x < pb.xmax; x += 64) {
            const float v = ROI ? paste_value(pb, mi, R.mh, R.mw, y, x) : m[(long long)y * W + x];
            rs += v;
            const bool on = v > 0.5f;
            any |= on;
            rar += on;
            rin += on && (float)road[((long long)y * W + x) * cstride] > 0.5f;
        }
        pix += rs;
        size += (double)(u * u) * rs;
        if (__ballot(any) != 0 && lane == 0) vert += u;
        inter += rin;
        area += rar;
    }
    double vals[5] = {pix, size, vert, inter, area};
    for (int k = 0; k < 5; ++k) {
        double v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
        for (int k = 0; k < 5; ++k) t[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        if (S > 1) {
            for (int k = 0; k < 5; ++k) part[((long long)(b * n + i) * S + blockIdx.z) * 5 + k] = t[k];
            return;
        }
        float *o = out5 + (long long)(b * n + i) * 5;
        o[0] = (float)t[0];
        o[1] = (float)t[1];
        o[3] = (float)t[2];
        const float ioi = (float)t[3] / ((float)t[4] + 1e-5f);                  // misc.py:617
        o[4] = ioi > ioi_threshold ? 1.f : 0.f;
    }
}

// S > 1: the bands' partial sums in band order, then the same five outputs; the column maxima of the column slices
__global__ void instance_finish_kernel(const double *__restrict__ part, const float *__restrict__ cmax, int S, int total,
                                       float ioi_threshold, float *__restrict__ out5) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total) return;
    double t[5] = {0, 0, 0, 0, 0};
    float best = -INFINITY;
    for (int z = 0; z < S; ++z) {
        for (int k = 0; k < 5; ++k) t[k] += part[((long long)r * S + z) * 5 + k];
        best = fmaxf(best, cmax[(long long)r * S + z]);
    }
    float *o = out5 + (long long)r * 5;
    o[0] = (float)t[0];
    o[1] = (float)t[1];
    o[2] = best;
    o[3] = (float)t[2];
    const float ioi = (float)t[3] / ((float)t[4] + 1e-5f);
    o[4] = ioi > ioi_threshold ? 1.f : 0.f;
}

// horizontal size = max over x of sum_y unit[y] * mask[y, x] (misc.py:656-658): threads own columns.  ROI: columns
// outside the box sum to exactly 0 (one of them is accounted for when the box leaves any), rows outside add exactly 0.
template <bool ROI>
__global__ void __launch_bounds__(256) instance_cols_kernel(const float *__restrict__ masks, RoiSrc R,
                                                            const float *__restrict__ unit,
                                                            int n, int H, int W, float *__restrict__ out5, int S,
                                                            float *__restrict__ cmax) {
    __shared__ float red[4];
    const int i = blockIdx.x, b = blockIdx.y;
    const int cband = ((W + S - 1) / S + 255) / 256 * 256;     // S > 1: blockIdx.z owns a band of columns (multiple of 256)
    const float *m = ROI ? nullptr : masks + ((long long)(b * n + i) * H) * W;
    PasteBox pb = {0, W, 0, H, 0.f, 0.f};
    const int32_t *mi = nullptr;
    if (ROI) {
        pb = paste_box(R.det + (long long)(b * n + i) * 6, *R.thr, R.mh, R.mw, H, W);
        mi = R.masks + (long long)(b * n + i) * R.mh * R.mw;
    }
    float best = -INFINITY;
    if (ROI && threadIdx.x == 0 && (pb.xmax - pb.xmin < W)) best = 0.f;      // a column the box does not reach
    const int xlo = max(pb.xmin, (int)blockIdx.z * cband), xhi = min(pb.xmax, ((int)blockIdx.z + 1) * cband);
    for (int x = (xlo / 256) * 256 + threadIdx.x; x < xhi; x += 256) {
        float s = 0.f;
        if (ROI) {
            if (x >= pb.xmin)
                for (int y = pb.ymin; y < pb.ymax; ++y) s += unit[b * H + y] * paste_value(pb, mi, R.mh, R.mw, y, x);
        } else {
            for (int y = 0; y < H; ++y) s += unit[b * H + y] * m[(long long)y * W + x];
        }
        best = fmaxf(best, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_down(best, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (S > 1) cmax[(long long)(b * n + i) * S + blockIdx.z] = v;
        else out5[(long long)(b * n + i) * 5 + 2] = v;
    }
}

// slices per instance for the summary kernels: one block per (instance, image) fills the chip from ~128 blocks on;
// below that the rows / columns of an instance are cut into up to 32 bands
static int summary_slices(int B, int n) {
    const int blocks = B * n;
    if (blocks >= 128) return 1;
    const int s = 256 / blocks;
    return s > 32 ? 32 : (s < 1 ? 1 : s);
}
constexpr long long SUMMARY_PART_BYTES = 128 * 32 * (5 * 8 + 4) + 64;      // partial sums + column maxima of < 128 x 32 slices

template <bool ROI>
static int launch_summary(const float *masks, const RoiSrc &R, const int32_t *seg, int seg_channels, int road_channel,
                          const float *unit, void *part_ws, int B, int n, int H, int W, float ioi_threshold, float *out5,
                          hipStream_t s) {
    const int S = summary_slices(B, n);
    double *part = reinterpret_cast<double *>(part_ws);
    float *cmax = reinterpret_cast<float *>(part + (size_t)B * n * S * 5);
    hipLaunchKernelGGL(instance_rows_kernel<ROI>, dim3(n, B, S), dim3(256), 0, s, masks, R, seg, seg_channels, road_channel, unit,
                       n, H, W, ioi_threshold, out5, S, part);
    hipLaunchKernelGGL(instance_cols_kernel<ROI>, dim3(n, B, S), dim3(256), 0, s, masks, R, unit, n, H, W, out5, S, cmax);
    if (S > 1)
        hipLaunchKernelGGL(instance_finish_kernel, dim3((B * n + 63) / 64), dim3(64), 0, s, part, cmax, S, B * n, ioi_threshold,
                           out5);
    return ML_OK;
}

}  // namespace

extern "C" int ml_crop_pad_mask_f32(const int32_t *det, const int32_t *masks, float *out, int32_t *threshold_ws, int32_t B,
                                    int32_t n, int32_t mh, int32_t mw, int32_t H, int32_t W, void *stream) {
    ML_REQUIRE(det && masks && out && threshold_ws, "crop_pad_mask: null pointer");
    ML_REQUIRE(B > 0 && n > 0 && mh > 0 && mw > 0 && H > 0 && W > 0, "crop_pad_mask: bad dims");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conf_threshold_kernel, dim3(1), dim3(256), 0, s, det, B * n, threshold_ws);
    const long long total = (long long)B * n * H * W;
    ML_REQUIRE(total / TPB < (1ll << 31), "crop_pad_mask: output too large for one launch");
    hipLaunchKernelGGL(crop_pad_mask_kernel, dim3(grid_for(total)), dim3(TPB), 0, s, det, masks, threshold_ws, out, n, mh, mw,
                       H, W, total);
    ML_CHECK_LAUNCH("crop_pad_mask");
    return ML_OK;
}

extern "C" int ml_nonzero_bbox_i32(const int32_t *map, int32_t B, int32_t H, int32_t W, int32_t cstride, int32_t coff,
                                   int32_t *box5, void *stream) {
    ML_REQUIRE(map && box5 && B > 0 && H > 0 && W > 0 && cstride > 0 && coff >= 0 && coff < cstride, "nonzero_bbox: bad arguments");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(nonzero_bbox_kernel, dim3(grid_for((total + BBOX_EPT - 1) / BBOX_EPT)), dim3(TPB), 0, (hipStream_t)stream,
                       map, cstride, coff, H, W, total, box5);
    ML_CHECK_LAUNCH("nonzero_bbox");
    return ML_OK;
}

extern "C" int64_t ml_instance_summary_workspace_bytes(int32_t B, int32_t H) {
    return ((int64_t)B * H * 2 + B) * (int64_t)sizeof(int32_t) + (int64_t)B * H * (int64_t)sizeof(float) + 256 +
           SUMMARY_PART_BYTES;
}

extern "C" int ml_instance_summary_f32(const int32_t *seg, int32_t seg_channels, int32_t road_channel, const float *masks,
                                       float *out5, int32_t B, int32_t n, int32_t H, int32_t W, float default_road_size,
                                       float ioi_threshold, void *workspace, void *stream) {
    ML_REQUIRE(seg && masks && out5 && workspace, "instance_summary: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && n > 0 && H > 0 && H < 65536 && W > 0, "instance_summary: bad dims");
    ML_REQUIRE(road_channel >= 0 && road_channel < seg_channels, "instance_summary: road channel out of range");
    hipStream_t s = (hipStream_t)stream;
    int32_t *xmin = reinterpret_cast<int32_t *>(workspace);
    int32_t *xmax = xmin + (size_t)B * H;
    int32_t *last_row = xmax + (size_t)B * H;
    float *unit = reinterpret_cast<float *>(last_row + B + ((B & 1) ? 1 : 0));
    ML_REQUIRE(hipMemsetAsync(last_row, 0xff, (size_t)B * sizeof(int32_t), s) == hipSuccess, "instance_summary: memset failed");
    hipLaunchKernelGGL(road_row_extent_kernel, dim3(H, B), dim3(64), 0, s, seg, seg_channels, road_channel, H, W, xmin, xmax,
                       last_row);
    hipLaunchKernelGGL(road_unit_length_kernel, dim3(B), dim3(256), 0, s, xmin, xmax, last_row, H, default_road_size, unit);
    const RoiSrc none = {nullptr, nullptr, nullptr, 0, 0};
    void *part_ws = reinterpret_cast<char *>(workspace) + (ml_instance_summary_workspace_bytes(B, H) - SUMMARY_PART_BYTES) / 8 * 8;
    launch_summary<false>(masks, none, seg, seg_channels, road_channel, unit, part_ws, B, n, H, W, ioi_threshold, out5, s);
    ML_CHECK_LAUNCH("instance_summary");
    return ML_OK;
}

extern "C" int ml_instance_summary_rois_f32(const int32_t *seg, int32_t seg_channels, int32_t road_channel, const int32_t *det,
                                            const int32_t *roi_masks, float *out5, int32_t B, int32_t n, int32_t mh, int32_t mw,
                                            int32_t H, int32_t W, float default_road_size, float ioi_threshold, void *workspace,
                                            void *stream) {
    ML_REQUIRE(seg && det && roi_masks && out5 && workspace, "instance_summary_rois: null pointer");
    ML_REQUIRE(B > 0 && B < 65536 && n > 0 && H > 0 && H < 65536 && W > 0 && mh > 0 && mw > 0, "instance_summary_rois: bad dims");
    ML_REQUIRE(road_channel >= 0 && road_channel < seg_channels, "instance_summary_rois: road channel out of range");
    hipStream_t s = (hipStream_t)stream;
    int32_t *xmin = reinterpret_cast<int32_t *>(workspace);
    int32_t *xmax = xmin + (size_t)B * H;
    int32_t *last_row = xmax + (size_t)B * H;
    float *unit = reinterpret_cast<float *>(last_row + B + ((B & 1) ? 1 : 0));
    int32_t *thr = reinterpret_cast<int32_t *>(unit + (size_t)B * H);        // (the workspace size leaves 256 spare bytes)
    ML_REQUIRE(hipMemsetAsync(last_row, 0xff, (size_t)B * sizeof(int32_t), s) == hipSuccess, "instance_summary_rois: memset failed");
    hipLaunchKernelGGL(conf_threshold_kernel, dim3(1), dim3(256), 0, s, det, B * n, thr);
    hipLaunchKernelGGL(road_row_extent_kernel, dim3(H, B), dim3(64), 0, s, seg, seg_channels, road_channel, H, W, xmin, xmax,
                       last_row);
    hipLaunchKernelGGL(road_unit_length_kernel, dim3(B), dim3(256), 0, s, xmin, xmax, last_row, H, default_road_size, unit);
    const RoiSrc R = {det, roi_masks, thr, mh, mw};
    void *part_ws = reinterpret_cast<char *>(workspace) + (ml_instance_summary_workspace_bytes(B, H) - SUMMARY_PART_BYTES) / 8 * 8;
    launch_summary<true>(nullptr, R, seg, seg_channels, road_channel, unit, part_ws, B, n, H, W, ioi_threshold, out5, s);
    ML_CHECK_LAUNCH("instance_summary_rois");
    return ML_OK;
}
