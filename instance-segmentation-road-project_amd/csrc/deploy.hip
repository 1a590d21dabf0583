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
    const int32_t *d = det + row * 6;
    float v = 0.f;
    if (d[5] >= *thr) {
        // box = max(box, 1) elementwise, then float (misc.py:378-382)
        const float cx = (float)max(d[0], 1), cy = (float)max(d[1], 1), w = (float)max(d[2], 1), h = (float)max(d[3], 1);
        const int xmin = min(max((int)ceilf(cx - w / 2.f), 0), W), xmax = min(max((int)ceilf(cx + w / 2.f), 0), W);
        const int ymin = min(max((int)ceilf(cy - h / 2.f), 0), H), ymax = min(max((int)ceilf(cy + h / 2.f), 0), H);
        const int oh = ymax - ymin, ow = xmax - xmin;
        if (y >= ymin && y < ymax && x >= xmin && x < xmax) {      // (a zero-sized box pastes nothing)
            const float sy = oh > 1 ? (float)(mh - 1) / (float)(oh - 1) : 0.f;
            const float sx = ow > 1 ? (float)(mw - 1) / (float)(ow - 1) : 0.f;
            const float fy = (float)(y - ymin) * sy, fx = (float)(x - xmin) * sx;
            const float fly = floorf(fy), flx = floorf(fx);
            const int y0 = max((int)fly, 0), x0 = max((int)flx, 0);
            const int y1 = min((int)ceilf(fy), mh - 1), x1 = min((int)ceilf(fx), mw - 1);
            const float ty = fy - fly, tx = fx - flx;
            const int32_t *m = masks + row * mh * mw;
            const float tl = (float)m[y0 * mw + x0], tr = (float)m[y0 * mw + x1];
            const float bl = (float)m[y1 * mw + x0], br = (float)m[y1 * mw + x1];
            const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
            v = top + (bot - top) * ty;
        }
    }
    out[idx] = v;
}

// ---- CrackToInstance (misc.py:524-551): bounding box of the non-zero pixels of a [B,H,W] int32 map over the
// WHOLE batch (the reference reduces tf.where over all rows).  box = {ymin, xmin, ymax, xmax, any}
__global__ void nonzero_bbox_kernel(const int32_t *__restrict__ map, int cstride, int coff, int H, int W, long long total,
                                    int32_t *__restrict__ box) {
    const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    if (map[idx * cstride + coff] != 0) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        atomicMin(&box[0], y);
        atomicMin(&box[1], x);
        atomicMax(&box[2], y);
        atomicMax(&box[3], x);
        box[4] = 1;
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

// per (instance, image): rows are dealt to the 4 waves, lanes sweep x.  out5 = {pixel sum, instance size,
// (horizontal: second kernel), vertical size, include_my_road}
__global__ void __launch_bounds__(256) instance_rows_kernel(const float *__restrict__ masks, const int32_t *__restrict__ seg,
                                                            int cstride, int road_coff, const float *__restrict__ unit, int n,
                                                            int H, int W, float ioi_threshold, float *__restrict__ out5) {
    __shared__ double red[5][4];
    const int i = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *m = masks + ((long long)(b * n + i) * H) * W;
    const int32_t *road = seg + ((long long)b * H * W) * cstride + road_coff;
    double pix = 0, size = 0, vert = 0, inter = 0, area = 0;
    for (int y = wave; y < H; y += 4) {
        const float u = unit[b * H + y];
        float rs = 0.f;
        int rin = 0, rar = 0;
        bool any = false;
        for (int x = lane; x < W; x += 64) {
            const float v = m[(long long)y * W + x];
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
        float *o = out5 + (long long)(b * n + i) * 5;
        o[0] = (float)t[0];
        o[1] = (float)t[1];
        o[3] = (float)t[2];
        const float ioi = (float)t[3] / ((float)t[4] + 1e-5f);                  // misc.py:617
        o[4] = ioi > ioi_threshold ? 1.f : 0.f;
    }
}

// horizontal size = max over x of sum_y unit[y] * mask[y, x] (misc.py:656-658): threads own columns
__global__ void __launch_bounds__(256) instance_cols_kernel(const float *__restrict__ masks, const float *__restrict__ unit,
                                                            int n, int H, int W, float *__restrict__ out5) {
    __shared__ float red[4];
    const int i = blockIdx.x, b = blockIdx.y;
    const float *m = masks + ((long long)(b * n + i) * H) * W;
    float best = -INFINITY;
    for (int x = threadIdx.x; x < W; x += 256) {
        float s = 0.f;
        for (int y = 0; y < H; ++y) s += unit[b * H + y] * m[(long long)y * W + x];
        best = fmaxf(best, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_down(best, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) out5[(long long)(b * n + i) * 5 + 2] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
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
    hipLaunchKernelGGL(nonzero_bbox_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, map, cstride, coff, H, W,
                       total, box5);
    ML_CHECK_LAUNCH("nonzero_bbox");
    return ML_OK;
}

extern "C" int64_t ml_instance_summary_workspace_bytes(int32_t B, int32_t H) {
    return ((int64_t)B * H * 2 + B) * (int64_t)sizeof(int32_t) + (int64_t)B * H * (int64_t)sizeof(float) + 256;
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
    hipLaunchKernelGGL(instance_rows_kernel, dim3(n, B), dim3(256), 0, s, masks, seg, seg_channels, road_channel, unit, n, H, W,
                       ioi_threshold, out5);
    hipLaunchKernelGGL(instance_cols_kernel, dim3(n, B), dim3(256), 0, s, masks, unit, n, H, W, out5);
    ML_CHECK_LAUNCH("instance_summary");
    return ML_OK;
}
