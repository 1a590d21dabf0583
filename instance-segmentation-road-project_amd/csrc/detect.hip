// Detection post-processing of the RetinaMask head, device-resident with fixed capacities
// (no host round trip inside):  RestoreBoxes -> DetectionProposal (threshold, per-(image,class)
// greedy NMS, per-image cross-class NMS, -1 padding) -> MaskDistribute + per-level slot lists
// -> crop_and_resize + MoldBatch.
//
// Index/byte work that must be BIT-EXACT against the oracle: every float expression that feeds a
// comparison (IoU > thr, in_y < 0, floor(log2)) is evaluated with FP contraction OFF so it
// rounds exactly like the reference's separately-rounded TF ops.
//
// NMS formulation: TF's greedy loop (pop best score; keep unless IoU with an already kept box
// > thr) is run as "select the best ALIVE candidate, then kill every alive candidate whose IoU
// with it exceeds thr" -- identical result, and each of the <= max_out rounds is a flat parallel
// sweep: a 64-bit key (score bits << 32 | ~index) makes the arg-max a plain max-reduction (wave
// shuffles + one LDS hop) and breaks score ties towards the lower index.  No sort is needed.
#include "common.h"
#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// max over the whole block, result broadcast to every thread; `red` has >= 17 slots
__device__ __forceinline__ u64 block_max_u64(u64 v, u64 *red) {
    v = wave_max_u64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous round's readers
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = 0;
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) m = red[w] > m ? red[w] : m;
        red[16] = m;
    }
    __syncthreads();
    return red[16];
}

// TF non_max_suppression_op.cc IOU(), float32 op for op (boxes are y1,x1,y2,x2)
__device__ __forceinline__ float iou_tf(const f32x4 a, const f32x4 b) {
    const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
    const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
    const float ymin_j = fminf(b[0], b[2]), xmin_j = fminf(b[1], b[3]);
    const float ymax_j = fmaxf(b[0], b[2]), xmax_j = fmaxf(b[1], b[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy1 = fmaxf(ymin_i, ymin_j), ix1 = fmaxf(xmin_i, xmin_j);
    const float iy2 = fminf(ymax_i, ymax_j), ix2 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
    return inter / ((area_i + area_j) - inter);
}

// NormalizeBoxes with the default shape (ones): pixel corners (detection.py:362,488)
__device__ __forceinline__ f32x4 corners(const f32x4 b) {
    const float hw = b[2] / 2.f, hh = b[3] / 2.f;
    f32x4 r = {b[1] - hh, b[0] - hw, b[1] + hh, b[0] + hw};
    return r;
}

// ------------------------------------------------------------------ RestoreBoxes
__global__ void restore_boxes_kernel(const float *__restrict__ loc, const int *__restrict__ pri,
                                     float *__restrict__ boxes, int A, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int a = (int)(i % A);
    const f32x4 l = *reinterpret_cast<const f32x4 *>(loc + i * 4);
    const int4 p = *reinterpret_cast<const int4 *>(pri + (long long)a * 4);
    const float pcx = (float)p.x, pcy = (float)p.y, pw = (float)p.z, ph = (float)p.w;
    f32x4 o;
    o[0] = l[0] * pw + pcx;
    o[1] = l[1] * ph + pcy;
    o[2] = expf(l[2]) * pw;
    o[3] = expf(l[3]) * ph;
    *reinterpret_cast<f32x4 *>(boxes + i * 4) = o;
}

// ------------------------------------------------------------------ DetectionProposal
struct DetWs {
    int *bucket_count;   // [B*C]
    int *bucket_first;   // [B*C]  min(a*C + c) over the bucket = tf.unique first-occurrence key
    int *s1_count;       // [B*C]
    int *s1_anchor;      // [B*C*max_out]
    u64 *keys;           // [B*C*A]
    f32x4 *cbox;         // [B*C*A]  (y1,x1,y2,x2)
};

__host__ __device__ inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

static DetWs det_ws_carve(void *ws, int B, int A, int C, int max_out, long long *bytes) {
    char *p = reinterpret_cast<char *>(ws);
    long long off = 0;
    DetWs w;
    const long long BC = (long long)B * C;
    w.bucket_count = reinterpret_cast<int *>(p + off); off = align_up(off + BC * 4, 256);
    w.bucket_first = reinterpret_cast<int *>(p + off); off = align_up(off + BC * 4, 256);
    w.s1_count = reinterpret_cast<int *>(p + off); off = align_up(off + BC * 4, 256);
    w.s1_anchor = reinterpret_cast<int *>(p + off); off = align_up(off + BC * max_out * 4, 256);
    w.keys = reinterpret_cast<u64 *>(p + off); off = align_up(off + BC * A * 8, 256);
    w.cbox = reinterpret_cast<f32x4 *>(p + off); off = align_up(off + BC * A * 16, 256);
    *bytes = off;
    return w;
}

__global__ void det_init_kernel(DetWs w, int BC) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < BC) { w.bucket_count[i] = 0; w.bucket_first[i] = 0x7fffffff; w.s1_count[i] = 0; }
}

// 1. filtering (detection.py:491-495).  grid (chunks, B): a block scans DET_EPB consecutive scores of
//    one image.  Survivors are ranked with LDS atomics (one counter per class), then ONE global
//    atomicAdd per (block, class) reserves the block's range in the bucket -- per-score global
//    atomics on 5 hot counters serialise (3.8 ms for 13 M scores; this form is HBM-bound).
//    Bucket order is irrelevant (keys carry the anchor index).
constexpr int DET_EPT = 8;                 // scores per thread
constexpr int DET_EPB = 256 * DET_EPT;     // scores per block

__global__ void __launch_bounds__(256)
det_threshold_kernel(const float *__restrict__ cls, const float *__restrict__ boxes, DetWs w, int A, int C, float thr) {
    __shared__ int cnt[64], gbase[64], first[64];
    const int b = blockIdx.y;
    const long long AC = (long long)A * C;
    const long long e0 = (long long)blockIdx.x * DET_EPB;
    if (threadIdx.x < C) { cnt[threadIdx.x] = 0; first[threadIdx.x] = 0x7fffffff; }
    __syncthreads();
    float sc[DET_EPT];
    int pos[DET_EPT];
#pragma unroll
    for (int k = 0; k < DET_EPT; ++k) {
        const long long e = e0 + k * 256 + threadIdx.x;       // (a*C + c) inside image b
        pos[k] = -1;
        sc[k] = 0.f;
        if (e < AC) {
            const float s = cls[(long long)b * AC + e];
            if (s >= thr) {
                const int c = (int)(e % C);
                sc[k] = s;
                pos[k] = atomicAdd(&cnt[c], 1);
                atomicMin(&first[c], (int)e);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < C && cnt[threadIdx.x] > 0) {
        gbase[threadIdx.x] = atomicAdd(&w.bucket_count[b * C + threadIdx.x], cnt[threadIdx.x]);
        atomicMin(&w.bucket_first[b * C + threadIdx.x], first[threadIdx.x]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DET_EPT; ++k) {
        if (pos[k] < 0) continue;
        const long long e = e0 + k * 256 + threadIdx.x;
        const int c = (int)(e % C);
        const int a = (int)(e / C);
        const long long slot = (long long)(b * C + c) * A + gbase[c] + pos[k];
        w.keys[slot] = ((u64)__float_as_uint(sc[k]) << 32) | (u64)(0xffffffffu - (unsigned)a);
        w.cbox[slot] = corners(*reinterpret_cast<const f32x4 *>(boxes + ((long long)b * A + a) * 4));
    }
}

// 2. per-(image,class) NMS (detection.py:499-524): one block per bucket.
//    Keys live in registers (NMS_R per thread, slot = r*512 + tid) when the bucket fits, so a
//    round is: register max -> block max -> owner publishes the winner's box -> kill sweep whose
//    box loads are issued in independent batches (the sweep is latency-, not bandwidth-bound).
//    Buckets larger than 1024*NMS_R fall back to keys in global memory.
constexpr int NMS_T = 512;
constexpr int NMS_R = 48;

__global__ void __launch_bounds__(NMS_T)
det_nms_bucket_kernel(DetWs w, int A, int max_out, float iou_thr) {
    __shared__ u64 red[17];
    __shared__ f32x4 sel_box;
    const int bucket = blockIdx.x;
    const int n = w.bucket_count[bucket];
    u64 *keys = w.keys + (long long)bucket * A;
    const f32x4 *cb = w.cbox + (long long)bucket * A;
    int picked = 0;
    if (n <= NMS_T * NMS_R) {
        u64 k[NMS_R];
#pragma unroll
        for (int r = 0; r < NMS_R; ++r) {
            const int i = r * NMS_T + threadIdx.x;
            k[r] = i < n ? keys[i] : 0;
        }
        for (; picked < max_out; ++picked) {
            u64 best = 0;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) best = k[r] > best ? k[r] : best;
            best = block_max_u64(best, red);
            if (best == 0) break;  // uniform: nothing alive
#pragma unroll
            for (int r = 0; r < NMS_R; ++r)
                if (k[r] == best) { sel_box = cb[r * NMS_T + threadIdx.x]; k[r] = 0; }
            __syncthreads();
            const f32x4 sb = sel_box;
            if (threadIdx.x == 0)
                w.s1_anchor[(long long)bucket * max_out + picked] = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
#pragma unroll
            for (int r0 = 0; r0 < NMS_R; r0 += 8) {
                if (r0 * NMS_T >= n) break;                       // uniform
                f32x4 bx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {                     // 8 independent loads in flight
                    const int i = (r0 + u) * NMS_T + threadIdx.x;
                    bx[u] = (k[r0 + u] != 0) ? cb[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k[r0 + u] != 0 && iou_tf(bx[u], sb) > iou_thr) k[r0 + u] = 0;
            }
        }
    } else {
        __shared__ int sel_slot;
        for (; picked < max_out; ++picked) {
            u64 best = 0;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const u64 kk = keys[i];
                best = kk > best ? kk : best;
            }
            best = block_max_u64(best, red);
            if (best == 0) break;
            for (int i = threadIdx.x; i < n; i += blockDim.x)
                if (keys[i] == best) sel_slot = i;
            __syncthreads();
            const f32x4 sb = cb[sel_slot];
            if (threadIdx.x == 0)
                w.s1_anchor[(long long)bucket * max_out + picked] = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const u64 kk = keys[i];
                if (kk == 0) continue;
                if (i == sel_slot || iou_tf(cb[i], sb) > iou_thr) keys[i] = 0;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) w.s1_count[bucket] = picked;
}

// 3+4. per-image cross-class NMS (detection.py:531-555) + result rows (:557-563) + -1 padding.
//      one block per image; candidates (<= C*max_out) live in LDS.
__global__ void __launch_bounds__(256)
det_nms_image_kernel(const float *__restrict__ cls, const float *__restrict__ boxes, DetWs w, float *__restrict__ proposed,
                     int *__restrict__ counts, int *__restrict__ kept, int A, int C, int max_out, float iou_thr) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ u64 red[17];
    __shared__ int order[64];     // bucket ranks (C <= 64)
    __shared__ int base[65];
    const int b = blockIdx.x;
    const int cap2 = C * max_out;
    f32x4 *cbx = reinterpret_cast<f32x4 *>(smem);
    u64 *keys = reinterpret_cast<u64 *>(smem + (size_t)cap2 * 16);
    int *anch = reinterpret_cast<int *>(smem + (size_t)cap2 * 24);
    int *clsid = reinterpret_cast<int *>(smem + (size_t)cap2 * 28);

    // tf.unique order of the image's (image,class) ids = ascending first-occurrence key
    if (threadIdx.x == 0) {
        int idx[64];
        for (int c = 0; c < C; ++c) idx[c] = c;
        for (int i = 1; i < C; ++i) {  // insertion sort by bucket_first
            const int v = idx[i];
            const int kv = w.bucket_first[b * C + v];
            int j = i - 1;
            while (j >= 0 && w.bucket_first[b * C + idx[j]] > kv) { idx[j + 1] = idx[j]; --j; }
            idx[j + 1] = v;
        }
        int off = 0;
        for (int r = 0; r < C; ++r) {
            order[r] = idx[r];
            base[r] = off;
            off += w.s1_count[b * C + idx[r]];
        }
        base[C] = off;
    }
    __syncthreads();
    const int n = base[C];
    for (int r = 0; r < C; ++r) {
        const int c = order[r];
        const int cnt = base[r + 1] - base[r];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int a = w.s1_anchor[((long long)b * C + c) * max_out + i];
            const int p = base[r] + i;
            const float s = cls[((long long)b * A + a) * C + c];
            keys[p] = ((u64)__float_as_uint(s) << 32) | (u64)(0xffffffffu - (unsigned)p);
            cbx[p] = corners(*reinterpret_cast<const f32x4 *>(boxes + ((long long)b * A + a) * 4));
            anch[p] = a;
            clsid[p] = c;
        }
    }
    __syncthreads();
    int picked = 0;
    for (; picked < max_out; ++picked) {
        u64 best = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) best = keys[i] > best ? keys[i] : best;
        best = block_max_u64(best, red);
        if (best == 0) break;
        const int p = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
        const f32x4 sb = cbx[p];
        if (threadIdx.x < 6) {
            const int a = anch[p], c = clsid[p];
            float v;
            if (threadIdx.x < 4) v = boxes[((long long)b * A + a) * 4 + threadIdx.x];
            else if (threadIdx.x == 4) v = (float)c;
            else v = __uint_as_float((unsigned)(best >> 32));
            proposed[((long long)b * max_out + picked) * 6 + threadIdx.x] = v;
            if (kept && threadIdx.x < 2)
                kept[((long long)b * max_out + picked) * 2 + threadIdx.x] = threadIdx.x == 0 ? a : c;
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (keys[i] == 0) continue;
            if (i == p || iou_tf(cbx[i], sb) > iou_thr) keys[i] = 0;
        }
        __syncthreads();
    }
    for (int i = picked * 6 + threadIdx.x; i < max_out * 6; i += blockDim.x)
        proposed[(long long)b * max_out * 6 + i] = -1.f;
    if (kept)
        for (int i = picked * 2 + threadIdx.x; i < max_out * 2; i += blockDim.x)
            kept[(long long)b * max_out * 2 + i] = -1;
    if (threadIdx.x == 0) counts[b] = picked;
}

// ------------------------------------------------------------------ MaskDistribute + level slots
__global__ void mask_distribute_kernel(const float *__restrict__ rows, int rs, int has_k, float *__restrict__ kvals,
                                       int *__restrict__ level_slots, int *__restrict__ level_counts, int cap, int max_k,
                                       float base_size) {
    extern __shared__ int kbuf[];  // [cap]
    const int b = blockIdx.x;
    const int L = max_k + 1;
    const float eps = 1e-7f;  // K.epsilon()
    for (int i = threadIdx.x; i < cap; i += blockDim.x) {
        const float *r = rows + ((long long)b * cap + i) * rs;
        float k;
        if (has_k) {
            k = r[0];
        } else {
            const float cx = r[0];
            const float size = sqrtf(r[2] * r[3]);                                  // instance.py:56-57
            const float dk = logf((size + eps) / (base_size + eps)) / logf(2.f);    // :58
            k = floorf(dk);                                                         // :59
            k = fminf(fmaxf(k, 0.f), (float)max_k);                                 // :60
            if (cx == -1.f) k = -1.f;                                               // :61-62
        }
        if (kvals) kvals[(long long)b * cap + i] = k;
        int ki = -2;
        if (k == k && k >= -1.f && k <= (float)max_k && k == floorf(k)) ki = (int)k;  // tf.equal(k, fmap_id)
        kbuf[i] = ki;
    }
    __syncthreads();
    if (threadIdx.x < L) {
        const int lvl = threadIdx.x;
        int n = 0;
        for (int i = 0; i < cap; ++i)
            if (kbuf[i] == lvl) level_slots[((long long)b * L + lvl) * cap + n++] = i;
        level_counts[b * L + lvl] = n;
        for (int i = n; i < cap; ++i) level_slots[((long long)b * L + lvl) * cap + i] = -1;
    }
}

// ------------------------------------------------------------------ crop_and_resize + MoldBatch(-1)
__global__ void __launch_bounds__(256)
roi_crop_resize_kernel(const float *__restrict__ fmap, const float *__restrict__ rows, int rs, int roff,
                       const int *__restrict__ level_slots, const int *__restrict__ level_counts,
                       float *__restrict__ roi_fmaps, float *__restrict__ roi_boxes, int Hf, int Wf, int C4, int cap,
                       int L, int level, int n_l, int ch, int cw, float img_h, float img_w, int box_off,
                       int box_rows) {
    const int j = blockIdx.x % n_l;
    const int b = blockIdx.x / n_l;
    const int cnt = level_counts[b * L + level];
    const int C = C4 * 4;
    float *dst = roi_fmaps + ((long long)b * n_l + j) * ch * cw * C;
    float *brow = roi_boxes + ((long long)b * box_rows + box_off + j) * 6;
    const int total = ch * cw * C4;
    if (j >= cnt) {  // MoldBatch padding (misc.py:276-282): -1 for features AND boxes
        const f32x4 m1 = {-1.f, -1.f, -1.f, -1.f};
        for (int i = threadIdx.x; i < total; i += blockDim.x) *reinterpret_cast<f32x4 *>(dst + (long long)i * 4) = m1;
        if (threadIdx.x < 6) brow[threadIdx.x] = -1.f;
        return;
    }
    const int slot = level_slots[((long long)b * L + level) * cap + j];
    const float *r = rows + ((long long)b * cap + slot) * rs + roff;
    if (threadIdx.x < 6) brow[threadIdx.x] = r[threadIdx.x];
    // NormalizeBoxes(shape=image H,W): instance.py:115-116, detection.py:364-374
    const float cx = r[0], cy = r[1], bw = r[2], bh = r[3];
    const float x1 = (cx - bw / 2.f) / img_w, y1 = (cy - bh / 2.f) / img_h;
    const float x2 = (cx + bw / 2.f) / img_w, y2 = (cy + bh / 2.f) / img_h;
    const float hs = ch > 1 ? (y2 - y1) * (float)(Hf - 1) / (float)(ch - 1) : 0.f;
    const float ws = cw > 1 ? (x2 - x1) * (float)(Wf - 1) / (float)(cw - 1) : 0.f;
    const float *img = fmap + (long long)b * Hf * Wf * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c = (i % C4) * 4;
        const int pos = i / C4;
        const int ox = pos % cw, oy = pos / cw;
        const float in_y = ch > 1 ? y1 * (float)(Hf - 1) + (float)oy * hs : 0.5f * (y1 + y2) * (float)(Hf - 1);
        const float in_x = cw > 1 ? x1 * (float)(Wf - 1) + (float)ox * ws : 0.5f * (x1 + x2) * (float)(Wf - 1);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(in_y < 0.f || in_y > (float)(Hf - 1) || in_x < 0.f || in_x > (float)(Wf - 1))) {
            const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
            const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
            const float fy = in_y - (float)ty, fx = in_x - (float)lx;
            const f32x4 tl = *reinterpret_cast<const f32x4 *>(img + ((long long)ty * Wf + lx) * C + c);
            const f32x4 tr = *reinterpret_cast<const f32x4 *>(img + ((long long)ty * Wf + rx) * C + c);
            const f32x4 bl = *reinterpret_cast<const f32x4 *>(img + ((long long)by * Wf + lx) * C + c);
            const f32x4 br = *reinterpret_cast<const f32x4 *>(img + ((long long)by * Wf + rx) * C + c);
            const f32x4 top = tl + (tr - tl) * fx;
            const f32x4 bot = bl + (br - bl) * fx;
            v = top + (bot - top) * fy;
        }
        *reinterpret_cast<f32x4 *>(dst + (long long)i * 4) = v;
    }
}

}  // namespace

extern "C" int ml_restore_boxes_f32(const float *loc, const int32_t *priors, float *boxes, int32_t B, int32_t A,
                                    void *stream) {
    ML_REQUIRE(loc && priors && boxes && B > 0 && A > 0, "restore_boxes: bad arguments");
    ML_REQUIRE(ml_aligned16(loc) && ml_aligned16(priors) && ml_aligned16(boxes), "restore_boxes: 16-byte alignment");
    const long long total = (long long)B * A;
    hipLaunchKernelGGL(restore_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, loc,
                       priors, boxes, A, total);
    ML_CHECK_LAUNCH("restore_boxes");
    return ML_OK;
}

extern "C" int64_t ml_detection_workspace_bytes(int32_t B, int32_t A, int32_t C, int32_t max_out) {
    long long bytes = 0;
    det_ws_carve(nullptr, B, A, C, max_out, &bytes);
    return bytes;
}

extern "C" int ml_detection_proposal_f32(const float *cls_pred, const float *boxes, float *proposed, int32_t *counts,
                                         int32_t *kept, int32_t B, int32_t A, int32_t C, float min_confidence,
                                         float nms_iou, float post_iou, int32_t max_out, void *workspace,
                                         void *stream) {
    ML_REQUIRE(cls_pred && boxes && proposed && counts && workspace, "detection_proposal: null pointer");
    ML_REQUIRE(B > 0 && A > 0 && C > 0 && C <= 64 && max_out > 0, "detection_proposal: bad dims (C <= 64)");
    ML_REQUIRE((long long)A * C < (1ll << 31), "detection_proposal: A*C overflows the first-occurrence key");
    ML_REQUIRE(ml_aligned16(boxes) && (((uintptr_t)workspace) & 255) == 0, "detection_proposal: alignment");
    const long long lds2 = (long long)C * max_out * 32;
    ML_REQUIRE(lds2 <= 64 * 1024, "detection_proposal: C*max_out = %d exceeds the stage-2 LDS capacity 2048", C * max_out);
    long long bytes = 0;
    DetWs w = det_ws_carve(workspace, B, A, C, max_out, &bytes);
    hipStream_t s = (hipStream_t)stream;
    const int BC = B * C;
    hipLaunchKernelGGL(det_init_kernel, dim3((BC + 255) / 256), dim3(256), 0, s, w, BC);
    const long long per_image = (long long)A * C;
    hipLaunchKernelGGL(det_threshold_kernel, dim3((unsigned)((per_image + DET_EPB - 1) / DET_EPB), B), dim3(256), 0, s,
                       cls_pred, boxes, w, A, C, min_confidence);
    hipLaunchKernelGGL(det_nms_bucket_kernel, dim3(BC), dim3(NMS_T), 0, s, w, A, max_out, nms_iou);
    hipLaunchKernelGGL(det_nms_image_kernel, dim3(B), dim3(256), (size_t)lds2, s, cls_pred, boxes, w, proposed, counts,
                       kept, A, C, max_out, post_iou);
    ML_CHECK_LAUNCH("detection_proposal");
    return ML_OK;
}

extern "C" int ml_mask_distribute_i32(const float *rows, int32_t row_stride, int32_t has_k, float *kvals,
                                      int32_t *level_slots, int32_t *level_counts, int32_t B, int32_t cap,
                                      int32_t max_k, float base_size, void *stream) {
    ML_REQUIRE(rows && level_slots && level_counts && B > 0 && cap > 0, "mask_distribute: bad arguments");
    ML_REQUIRE(row_stride >= (has_k ? 5 : 4), "mask_distribute: row_stride too small");
    ML_REQUIRE(max_k >= 0 && max_k < 64, "mask_distribute: max_k out of range");
    hipLaunchKernelGGL(mask_distribute_kernel, dim3(B), dim3(128), (size_t)cap * 4, (hipStream_t)stream, rows, row_stride,
                       has_k, kvals, level_slots, level_counts, cap, max_k, base_size);
    ML_CHECK_LAUNCH("mask_distribute");
    return ML_OK;
}

extern "C" int ml_roi_crop_resize_f32(const float *fmap, const float *rows, int32_t row_stride, int32_t row_off,
                                      const int32_t *level_slots,
                                      const int32_t *level_counts, float *roi_fmaps, float *roi_boxes, int32_t B,
                                      int32_t Hf, int32_t Wf, int32_t C, int32_t cap, int32_t L, int32_t level,
                                      int32_t n_l, int32_t ch, int32_t cw, float img_h, float img_w, int32_t box_off,
                                      int32_t box_rows, void *stream) {
    ML_REQUIRE(fmap && rows && level_slots && level_counts && roi_fmaps && roi_boxes, "roi_crop: null pointer");
    ML_REQUIRE(row_off >= 0 && row_off + 6 <= row_stride, "roi_crop: rows must hold 6 columns from row_off");
    ML_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && C % 4 == 0 && n_l > 0 && ch > 0 && cw > 0, "roi_crop: bad dims");
    ML_REQUIRE(level >= 0 && level < L && n_l <= cap && box_off >= 0 && box_off + n_l <= box_rows, "roi_crop: bad level/rows");
    ML_REQUIRE(ml_aligned16(fmap) && ml_aligned16(roi_fmaps), "roi_crop: 16-byte alignment");
    hipLaunchKernelGGL(roi_crop_resize_kernel, dim3((unsigned)(B * n_l)), dim3(256), 0, (hipStream_t)stream, fmap, rows,
                       row_stride, row_off, level_slots, level_counts, roi_fmaps, roi_boxes, Hf, Wf, C / 4, cap, L, level, n_l, ch, cw, img_h,
                       img_w, box_off, box_rows);
    ML_CHECK_LAUNCH("roi_crop");
    return ML_OK;
}
