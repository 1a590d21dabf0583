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
// > thr).  Candidates carry a 64-bit key (score bits << 32 | ~index): key order IS TF's pop order
// (score descending, ties towards the lower index).  A band of candidates is sorted by key in LDS
// (bitonic, whole block), then resolved 64 at a time by ONE wave with ballots and lane reads -- no
// block barrier per selected box: the chunk is checked against the boxes already selected, then
// the lowest alive lane is selected and kills its chunk-mates, until the chunk is empty.
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

// `iou_tf(a, b) > thr`, bit for bit, without paying for the IEEE division on every pair: the quotient
// is first estimated with v_rcp_f32 (about 1 ulp); only when the estimate lies within 1e-5 (relative)
// of the threshold -- or the union is so small that the reciprocal could overflow -- is the exact
// division evaluated.  Outside that margin the correctly rounded quotient is on the same side of thr.
__device__ __forceinline__ bool iou_gt(const f32x4 a, const f32x4 b, float thr) {
    const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
    const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
    const float ymin_j = fminf(b[0], b[2]), xmin_j = fminf(b[1], b[3]);
    const float ymax_j = fmaxf(b[0], b[2]), xmax_j = fmaxf(b[1], b[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f > thr;
    const float iy1 = fmaxf(ymin_i, ymin_j), ix1 = fmaxf(xmin_i, xmin_j);
    const float iy2 = fminf(ymax_i, ymax_j), ix2 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
    const float uni = (area_i + area_j) - inter;
    if (uni > 1e-30f && uni < 1e30f) {
        const float est = inter * __builtin_amdgcn_rcpf(uni);
        if (est > thr * 1.00001f) return true;
        if (est < thr * 0.99999f) return false;
    }
    return inter / uni > thr;
}

// Descending bitonic sort of P (power of two) keys in LDS by the whole block; `idx` (optional) is a
// 16-bit payload permuted with the keys.  Keys are distinct except for the 0 padding.
template <bool HAS_IDX>
__device__ __forceinline__ void bitonic_sort_desc(u64 *key, unsigned short *idx, int P) {
    const int tid = threadIdx.x, T = blockDim.x;
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const bool desc = (i & k) == 0;
                const u64 a = key[i], b = key[ixj];
                if ((a < b) == desc) {
                    key[i] = b;
                    key[ixj] = a;
                    if (HAS_IDX) {
                        const unsigned short ia = idx[i];
                        idx[i] = idx[ixj];
                        idx[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
}

struct GreedyShared {
    u64 supp[64];     // supp[i] bit j: candidate j of the chunk is suppressed by candidate i (j > i)
    int dead[64];     // candidate suppressed by a box selected before this chunk
    int picked;
};

// Greedy NMS over candidates SORTED by key (descending; 0 = dead/padding) in LDS.
//   skey[m]  keys                                             sidx[m]  box slot of each key (HAS_IDX)
//   bbox[]   corner boxes by slot (slot = sidx[j], or ~low32(key) when !HAS_IDX)
//   sel[]    boxes selected so far (`picked` of them on entry, from earlier bands)
// The list is consumed 64 candidates (one chunk) at a time:
//   (1) whole block: wave w tests the chunk against the selected boxes q = w, w+nw, ... and against
//       its share of the chunk's own 64x64 suppression matrix -- every IoU of the chunk in parallel;
//   (2) wave 0: walks the alive mask with scalar bit operations only: select the lowest alive lane,
//       clear the lanes it suppresses (its matrix row, fetched with a lane read), repeat.
// So the serial part per selected box is ~10 instructions, and the IoU arithmetic is spread over
// the block.  emit(picked, key, slot) runs on the selected candidate's lane only.
// Called by all threads of the block (blockDim.x = 64 * nw, nw | 64); returns the new `picked`.
template <bool HAS_IDX, class Emit>
__device__ __forceinline__ int greedy_sorted(const u64 *skey, const unsigned short *sidx, const f32x4 *bbox, int m,
                                             f32x4 *sel, int picked, int max_out, float thr, GreedyShared *gs,
                                             Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nw = blockDim.x >> 6;
    const int bpw = 64 / nw;                             // matrix columns handled per wave
    auto slot_of = [&](u64 k, int j) {
        return HAS_IDX ? (int)sidx[j] : (int)(0xffffffffu - (unsigned)(k & 0xffffffffu));
    };
    if (tid < 64) { gs->supp[tid] = 0; gs->dead[tid] = 0; }
    __syncthreads();
    for (int base = 0; base < m && picked < max_out; base += 64) {
        if (skey[base] == 0) break;                      // uniform: sorted, nothing alive from here on
        const int j = base + lane;
        const u64 k = j < m ? skey[j] : 0;
        const int slot = k != 0 ? slot_of(k, j) : 0;
        f32x4 box = {0.f, 0.f, 0.f, 0.f};
        if (k != 0) box = bbox[slot];
        // (1a) against the boxes selected before this chunk
        if (k != 0) {
            bool dead = false;
            for (int q = wv; q < picked; q += nw)
                if (iou_gt(box, sel[q], thr)) { dead = true; break; }
            if (dead) gs->dead[lane] = 1;
        }
        // (1b) this wave's columns of the chunk's suppression matrix
        {
            u64 bits = 0;
            for (int u = 0; u < bpw; ++u) {
                const int jj = wv * bpw + u;
                const int pj = base + jj;
                const u64 kj = pj < m ? skey[pj] : 0;     // wave-uniform
                if (kj == 0) break;                       // sorted: the rest of the chunk is padding
                const f32x4 bj = bbox[slot_of(kj, pj)];
                if (k != 0 && jj > lane && iou_gt(bj, box, thr)) bits |= 1ull << jj;
            }
            if (bits != 0) atomicOr(&gs->supp[lane], bits);
        }
        __syncthreads();
        // (2) serial resolve on wave 0
        if (tid < 64) {
            const u64 row = gs->supp[lane];
            const unsigned row_lo = (unsigned)row, row_hi = (unsigned)(row >> 32);
            u64 mask = __ballot(k != 0 && gs->dead[lane] == 0);
            while (mask != 0 && picked < max_out) {
                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
                if (lane == l) {
                    sel[picked] = box;
                    emit(picked, k, slot);
                }
                ++picked;
                const u64 kill = ((u64)(unsigned)__builtin_amdgcn_readlane((int)row_hi, l) << 32) |
                                 (u64)(unsigned)__builtin_amdgcn_readlane((int)row_lo, l);
                mask &= ~(kill | (1ull << l));
            }
            gs->supp[lane] = 0;
            gs->dead[lane] = 0;
            if (lane == 0) gs->picked = picked;
        }
        __syncthreads();
        picked = gs->picked;
    }
    return picked;
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
    // second stage through global memory (only when C*max_out candidates per image do not fit the LDS kernel)
    int *s2_n;           // [B]           candidates of the image
    int *s2_count;       // [B]           boxes selected
    int *s2_sel;         // [B*max_out]   selected candidate positions p
    int *s2_anchor;      // [B*C*max_out] anchor of candidate p
    int *s2_class;       // [B*C*max_out] class of candidate p
    u64 *s2_keys;        // [B*C*max_out]
    f32x4 *s2_cbox;      // [B*C*max_out]
};

// what one launch of det_nms_bucket_kernel works on: `count[bucket]` candidates at keys / cbox + bucket * stride
struct NmsIo {
    const int *count;
    u64 *keys;
    const f32x4 *cbox;
    int *out_sel;        // [buckets * max_out]  low 32 key bits (inverted) of the selected candidates, in pick order
    int *out_count;      // [buckets]
    long long stride;
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
    w.s2_n = reinterpret_cast<int *>(p + off); off = align_up(off + (long long)B * 4, 256);
    w.s2_count = reinterpret_cast<int *>(p + off); off = align_up(off + (long long)B * 4, 256);
    w.s2_sel = reinterpret_cast<int *>(p + off); off = align_up(off + (long long)B * max_out * 4, 256);
    w.s2_anchor = reinterpret_cast<int *>(p + off); off = align_up(off + BC * max_out * 4, 256);
    w.s2_class = reinterpret_cast<int *>(p + off); off = align_up(off + BC * max_out * 4, 256);
    w.s2_keys = reinterpret_cast<u64 *>(p + off); off = align_up(off + BC * max_out * 8, 256);
    w.s2_cbox = reinterpret_cast<f32x4 *>(p + off); off = align_up(off + BC * max_out * 16, 256);
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
    const unsigned AC = (unsigned)A * (unsigned)C;            // < 2^31 (checked by the launcher)
    const unsigned e0 = blockIdx.x * (unsigned)DET_EPB;
    if (threadIdx.x < C) { cnt[threadIdx.x] = 0; first[threadIdx.x] = 0x7fffffff; }
    __syncthreads();
    float sc[DET_EPT];
    int pos[DET_EPT];
#pragma unroll
    for (int k = 0; k < DET_EPT; ++k) {                       // all loads first, then the ranking
        const unsigned e = e0 + k * 256 + threadIdx.x;        // (a*C + c) inside image b
        sc[k] = e < AC ? cls[(long long)b * AC + e] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < DET_EPT; ++k) {
        const unsigned e = e0 + k * 256 + threadIdx.x;
        pos[k] = -1;
        if (e < AC && sc[k] >= thr) {
            const int c = (int)(e % (unsigned)C);
            pos[k] = atomicAdd(&cnt[c], 1);
            atomicMin(&first[c], (int)e);
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
        const unsigned e = e0 + k * 256 + threadIdx.x;
        const int a = (int)(e / (unsigned)C);
        const int c = (int)(e - (unsigned)a * (unsigned)C);
        const long long slot = (long long)(b * C + c) * A + gbase[c] + pos[k];
        w.keys[slot] = ((u64)__float_as_uint(sc[k]) << 32) | (u64)(0xffffffffu - (unsigned)a);
        w.cbox[slot] = corners(*reinterpret_cast<const f32x4 *>(boxes + ((long long)b * A + a) * 4));
    }
}

// 2. per-(image,class) NMS (detection.py:499-524): one block per bucket, processed in SCORE BANDS.
//    Greedy NMS consumes candidates in descending score order and stops after max_out picks, so a
//    bucket of n candidates (17 k per bucket in the benchmark) rarely needs more than its top few
//    hundred.  A 2048-bin histogram of the score bits (monotone in the score) cuts the bucket into
//    bands of NMS_BAND_FIRST (x4 per band) .. NMS_BAND candidates, highest scores first; a band is gathered into LDS
//    (keys + boxes), sorted, and resolved by greedy_sorted(); the next band is only touched when
//    fewer than max_out boxes have been selected.
//    Exact: every candidate of a later band scores below every candidate of an earlier one, and ties
//    inside a band are broken by the index carried in the key.  A single bin with more than NMS_BAND
//    candidates (many identical scores) falls back to rounds over global memory for that bin.
constexpr int NMS_T = 512;
constexpr int NMS_BINS = 2048;
constexpr int NMS_BAND = 4096;
constexpr int NMS_BAND_FIRST = 256;    // target size of the first band; single-band buckets: n <= NMS_BAND_MIN
constexpr int NMS_BAND_MIN = 1024;
constexpr int NMS_U = 8;        // keys loaded per thread before any is consumed (memory-level parallelism)

struct NmsShared {
    int hist[NMS_BINS + 1];   // histogram, then in place its suffix sums: hist[b] = #candidates in bins >= b
    int wsum[NMS_T / 64];
    u64 red[17];
    GreedyShared greedy;
    int lo, band_n, sel_slot;
};

__global__ void __launch_bounds__(NMS_T)
det_nms_bucket_kernel(NmsIo io, int max_out, float iou_thr, float min_conf) {
    extern __shared__ __align__(16) unsigned char nms_smem[];
    f32x4 *bbox = reinterpret_cast<f32x4 *>(nms_smem);                                   // [NMS_BAND]
    f32x4 *sel = bbox + NMS_BAND;                                                        // [max_out]
    u64 *bkey = reinterpret_cast<u64 *>(sel + max_out);                                  // [NMS_BAND]
    unsigned short *bidx = reinterpret_cast<unsigned short *>(bkey + NMS_BAND);          // [NMS_BAND]
    NmsShared *sh = reinterpret_cast<NmsShared *>(bidx + NMS_BAND);
    const int bucket = blockIdx.x;
    const int n = io.count[bucket];
    u64 *keys = io.keys + (long long)bucket * io.stride;
    const f32x4 *cb = io.cbox + (long long)bucket * io.stride;
    const int tid = threadIdx.x;

    // monotone score -> bin map over [min_conf, 1]
    const unsigned lo_bits = min_conf > 0.f ? __float_as_uint(min_conf) : 0u;
    const unsigned range = 0x3F800000u > lo_bits ? 0x3F800000u - lo_bits + 1u : 1u;
    int shift = 0;
    while ((range >> shift) >= (unsigned)NMS_BINS) ++shift;
    auto bin_of = [&](u64 k) {
        const unsigned sb = (unsigned)(k >> 32);
        const unsigned d = sb > lo_bits ? sb - lo_bits : 0u;
        const unsigned bn = d >> shift;
        return (int)(bn < (unsigned)NMS_BINS ? bn : NMS_BINS - 1);
    };

    const bool single_band = n <= NMS_BAND_MIN;
    if (!single_band) {
        for (int i = tid; i < NMS_BINS; i += NMS_T) sh->hist[i] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += NMS_T * NMS_U) {          // NMS_U independent loads in flight per thread
            u64 kk[NMS_U];
#pragma unroll
            for (int u = 0; u < NMS_U; ++u) {
                const int i = i0 + u * NMS_T + tid;
                kk[u] = i < n ? keys[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < NMS_U; ++u)
                if (kk[u] != 0) atomicAdd(&sh->hist[bin_of(kk[u])], 1);
        }
        __syncthreads();
        // suffix sums in place (thread t owns bins 4t..4t+3): the band search below is then one parallel
        // compare per bin instead of a serial walk over (mostly empty) bins
        static_assert(NMS_BINS == 4 * NMS_T, "one thread per 4 bins");
        const int lane = tid & 63, wave = tid >> 6;
        int loc[4], own = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { loc[u] = sh->hist[4 * tid + u]; own += loc[u]; }
        int v = own;                                     // inclusive suffix over the wave's threads
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_down(v, off, 64);
            if (lane + off < 64) v += o;
        }
        if (lane == 0) sh->wsum[wave] = v;
        __syncthreads();
        for (int w2 = wave + 1; w2 < NMS_T / 64; ++w2) v += sh->wsum[w2];
        int run = v - own;
#pragma unroll
        for (int u = 3; u >= 0; --u) { run += loc[u]; sh->hist[4 * tid + u] = run; }
        if (tid == 0) sh->hist[NMS_BINS] = 0;
    }
    __syncthreads();

    int picked = 0;
    int hi = NMS_BINS;                                   // exclusive upper bin of the next band
    int band_target = NMS_BAND_FIRST;                    // grows x4 per band: most buckets finish in the first
    while (picked < max_out && hi > 0 && n > 0) {
        // ---- choose the band [lo, hi): top bins until it holds >= band_target, never more than NMS_BAND
        int lo = 0, bn = n;
        if (!single_band) {
            if (tid == 0) sh->lo = 0;
            __syncthreads();
            const int below = sh->hist[hi];              // candidates in the bands already processed
            const int want = below + band_target;
#pragma unroll
            for (int u = 0; u < 4; ++u) {                // the unique bin where the suffix count crosses `want`
                const int b = 4 * tid + u;
                if (b < hi && sh->hist[b] >= want && sh->hist[b + 1] < want) sh->lo = b;
            }
            __syncthreads();
            lo = sh->lo;                                 // 0 when fewer than band_target candidates are left
            bn = sh->hist[lo] - below;
            // never more than NMS_BAND: give up the crossing bin unless it is the band's only non-empty one
            if (bn > NMS_BAND && lo + 1 < hi && sh->hist[lo + 1] - below > 0) {
                ++lo;
                bn = sh->hist[lo] - below;
            }
            __syncthreads();
        }
        if (bn == 0) { hi = lo; continue; }
        if (bn <= NMS_BAND) {
            // ---- gather the band into LDS (any order), pad to a power of two, sort by key
            int P = 64;
            while (P < bn) P <<= 1;
            if (tid == 0) sh->band_n = 0;
            for (int j = bn + tid; j < P; j += NMS_T) { bkey[j] = 0; bidx[j] = 0; }
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += NMS_T * NMS_U) {
                u64 kk[NMS_U];
#pragma unroll
                for (int u = 0; u < NMS_U; ++u) {
                    const int i = i0 + u * NMS_T + tid;
                    kk[u] = i < n ? keys[i] : 0;
                }
#pragma unroll
                for (int u = 0; u < NMS_U; ++u) {
                    const u64 k = kk[u];
                    const int b = single_band ? lo : bin_of(k);
                    const bool hit = k != 0 && b >= lo && b < hi;
                    const u64 hits = __ballot(hit);      // one LDS atomic per wave, not per candidate
                    if (hits == 0) continue;
                    const int leader = __ffsll((long long)hits) - 1;
                    int wbase = 0;
                    if ((tid & 63) == leader) wbase = atomicAdd(&sh->band_n, __popcll(hits));
                    wbase = __builtin_amdgcn_readlane(wbase, __builtin_amdgcn_readfirstlane(leader));
                    if (hit) {
                        const int p = wbase + __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0));
                        bkey[p] = k;
                        bidx[p] = (unsigned short)p;
                        bbox[p] = cb[i0 + u * NMS_T + tid];
                    }
                }
            }
            __syncthreads();
            bitonic_sort_desc<true>(bkey, bidx, P);
            picked = greedy_sorted<true>(bkey, bidx, bbox, bn, sel, picked, max_out, iou_thr, &sh->greedy,
                                         [&](int slot_out, u64 key, int) {
                                             io.out_sel[(long long)bucket * max_out + slot_out] =
                                                 (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
                                         });
        } else {
            // ---- one bin larger than the LDS band (ties en masse): rounds over global memory for that bin
            for (int i = tid; i < n; i += NMS_T) {       // suppression by earlier selections first
                const u64 k = keys[i];
                if (k == 0) continue;
                const int b = bin_of(k);
                if (b < lo || b >= hi) continue;
                const f32x4 bi = cb[i];
                for (int q = 0; q < picked; ++q)
                    if (iou_gt(bi, sel[q], iou_thr)) { keys[i] = 0; break; }
            }
            __syncthreads();
            while (picked < max_out) {
                u64 best = 0;
                for (int i = tid; i < n; i += NMS_T) {
                    const u64 k = keys[i];
                    if (k == 0) continue;
                    const int b = bin_of(k);
                    if (b >= lo && b < hi && k > best) best = k;
                }
                best = block_max_u64(best, sh->red);
                if (best == 0) break;
                for (int i = tid; i < n; i += NMS_T)
                    if (keys[i] == best) { sh->sel_slot = i; keys[i] = 0; }
                __syncthreads();
                const f32x4 sb = cb[sh->sel_slot];
                if (tid == 0) {
                    sel[picked] = sb;
                    io.out_sel[(long long)bucket * max_out + picked] = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
                }
                for (int i = tid; i < n; i += NMS_T) {
                    const u64 k = keys[i];
                    if (k == 0) continue;
                    const int b = bin_of(k);
                    if (b >= lo && b < hi && iou_gt(cb[i], sb, iou_thr)) keys[i] = 0;
                }
                ++picked;
                __syncthreads();
            }
        }
        hi = lo;
        band_target = band_target * 4 < NMS_BAND ? band_target * 4 : NMS_BAND;
    }
    if (tid == 0) io.out_count[bucket] = picked;
}

// 3+4. per-image cross-class NMS (detection.py:531-555) + result rows (:557-563) + -1 padding.

// tf.unique order of the image's (image,class) ids = ascending first-occurrence key (detection.py:519-520).
// order[r] = class of rank r, base[r] = first candidate position of rank r, base[C] = candidates of the image.
// Called by the whole block; bfirst / bcount are 64-entry scratch arrays.
__device__ __forceinline__ void image_bucket_order(const DetWs &w, int b, int C, int *order, int *base, int *bfirst,
                                                   int *bcount) {
    if (threadIdx.x < C) {                                  // one parallel round trip for the bucket headers
        bfirst[threadIdx.x] = w.bucket_first[b * C + threadIdx.x];
        bcount[threadIdx.x] = w.s1_count[b * C + threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int idx[64];
        for (int c = 0; c < C; ++c) idx[c] = c;
        for (int i = 1; i < C; ++i) {  // insertion sort by bucket_first
            const int v = idx[i];
            const int kv = bfirst[v];
            int j = i - 1;
            while (j >= 0 && bfirst[idx[j]] > kv) { idx[j + 1] = idx[j]; --j; }
            idx[j + 1] = v;
        }
        int off = 0;
        for (int r = 0; r < C; ++r) {
            order[r] = idx[r];
            base[r] = off;
            off += bcount[idx[r]];
        }
        base[C] = off;
    }
    __syncthreads();
}

// Result rows (cx,cy,w,h,class,conf) of image b in pick order, -1 padding (MoldBatch, misc.py:276-282), the count,
// the optional (anchor, class) list, and the optional all-gather payload row
// [max_out*6 floats of `proposed` | count bit-cast to float] (masklab_hip/parallel.py: ONE collective per batch).
// ac(r) -> (anchor, class) of pick r.  Called by the whole block.
template <class AC>
__device__ __forceinline__ void write_result_rows(const float *__restrict__ cls, const float *__restrict__ boxes,
                                                  float *__restrict__ proposed, int *__restrict__ counts,
                                                  int *__restrict__ kept, float *__restrict__ payload, int b, int A, int C,
                                                  int max_out, int picked, AC ac) {
    float *pay = payload ? payload + (long long)b * (max_out * 6 + 1) : nullptr;
    for (int i = threadIdx.x; i < picked * 6; i += blockDim.x) {
        const int r = i / 6, f = i - r * 6;
        int a, c;
        ac(r, a, c);
        float v;
        if (f < 4) v = boxes[((long long)b * A + a) * 4 + f];
        else if (f == 4) v = (float)c;
        else v = cls[((long long)b * A + a) * C + c];
        proposed[((long long)b * max_out + r) * 6 + f] = v;
        if (pay) pay[i] = v;
        if (kept && f < 2) kept[((long long)b * max_out + r) * 2 + f] = f == 0 ? a : c;
    }
    for (int i = picked * 6 + threadIdx.x; i < max_out * 6; i += blockDim.x) {
        proposed[(long long)b * max_out * 6 + i] = -1.f;
        if (pay) pay[i] = -1.f;
    }
    if (kept)
        for (int i = picked * 2 + threadIdx.x; i < max_out * 2; i += blockDim.x)
            kept[(long long)b * max_out * 2 + i] = -1;
    if (threadIdx.x == 0) {
        counts[b] = picked;
        if (pay) pay[max_out * 6] = __int_as_float(picked);
    }
}

//      LDS form: one block per image; candidates (<= C*max_out <= 2048) live in LDS.
__global__ void __launch_bounds__(512)
det_nms_image_kernel(const float *__restrict__ cls, const float *__restrict__ boxes, DetWs w, float *__restrict__ proposed,
                     int *__restrict__ counts, int *__restrict__ kept, float *__restrict__ payload, int A, int C,
                     int max_out, float iou_thr) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int order[64];     // bucket ranks (C <= 64)
    __shared__ int base[65];
    __shared__ GreedyShared greedy;
    __shared__ int bfirst[64], bcount[64];
    const int b = blockIdx.x;
    const int cap2 = C * max_out;
    int P2 = 64;
    while (P2 < cap2) P2 <<= 1;
    f32x4 *cbx = reinterpret_cast<f32x4 *>(smem);                          // [cap2]
    f32x4 *sel = cbx + cap2;                                               // [max_out]
    u64 *keys = reinterpret_cast<u64 *>(sel + max_out);                    // [P2]
    int *anch = reinterpret_cast<int *>(keys + P2);                        // [cap2]
    int *clsid = anch + cap2;                                              // [cap2]
    int *pick = clsid + cap2;                                              // [max_out]

    image_bucket_order(w, b, C, order, base, bfirst, bcount);
    const int n = base[C];
    for (int p = threadIdx.x; p < n; p += blockDim.x) {      // flat over the image's candidates: one load chain
        int r = 0;
        while (base[r + 1] <= p) ++r;
        const int c = order[r];
        const int a = w.s1_anchor[((long long)b * C + c) * max_out + (p - base[r])];
        const float s = cls[((long long)b * A + a) * C + c];
        keys[p] = ((u64)__float_as_uint(s) << 32) | (u64)(0xffffffffu - (unsigned)p);
        cbx[p] = corners(*reinterpret_cast<const f32x4 *>(boxes + ((long long)b * A + a) * 4));
        anch[p] = a;
        clsid[p] = c;
    }
    __syncthreads();
    int P = 64;
    while (P < n) P <<= 1;
    for (int i = n + threadIdx.x; i < P; i += blockDim.x) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc<false>(keys, nullptr, P);
    // picks are recorded in LDS (re-using the dead tail of `sel` is not possible: it is live) ...
    const int picked = greedy_sorted<false>(
        keys, nullptr, cbx, n, sel, 0, max_out, iou_thr, &greedy, [&](int slot_out, u64, int p) { pick[slot_out] = p; });
    // ... and the result rows (:557-563) are written by the whole block afterwards
    write_result_rows(cls, boxes, proposed, counts, kept, payload, b, A, C, max_out, picked,
                      [&](int r, int &a, int &c) { const int p = pick[r]; a = anch[p]; c = clsid[p]; });
}

//      Global-memory form for C*max_out > 2048 (e.g. the constructor default nms_max_output_size = 1000,
//      detection.py:472): (i) the image's stage-1 survivors are laid out as one more bucket -- key = score bits and
//      the inverted POSITION p in tf.unique order, so ties break like the concatenated per-class list (:541) --,
//      (ii) det_nms_bucket_kernel runs over these B buckets with post_iou_threshold, (iii) rows are written.
__global__ void __launch_bounds__(256)
det_stage2_gather_kernel(const float *__restrict__ cls, const float *__restrict__ boxes, DetWs w, int A, int C, int max_out) {
    __shared__ int order[64], base[65], bfirst[64], bcount[64];
    const int b = blockIdx.x;
    const long long cap2 = (long long)C * max_out;
    image_bucket_order(w, b, C, order, base, bfirst, bcount);
    const int n = base[C];
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
        int r = 0;
        while (base[r + 1] <= p) ++r;
        const int c = order[r];
        const int a = w.s1_anchor[((long long)b * C + c) * max_out + (p - base[r])];
        const float s = cls[((long long)b * A + a) * C + c];
        w.s2_keys[b * cap2 + p] = ((u64)__float_as_uint(s) << 32) | (u64)(0xffffffffu - (unsigned)p);
        w.s2_cbox[b * cap2 + p] = corners(*reinterpret_cast<const f32x4 *>(boxes + ((long long)b * A + a) * 4));
        w.s2_anchor[b * cap2 + p] = a;
        w.s2_class[b * cap2 + p] = c;
    }
    if (threadIdx.x == 0) w.s2_n[b] = n;
}

__global__ void __launch_bounds__(256)
det_rows_kernel(const float *__restrict__ cls, const float *__restrict__ boxes, DetWs w, float *__restrict__ proposed,
                int *__restrict__ counts, int *__restrict__ kept, float *__restrict__ payload, int A, int C, int max_out) {
    const int b = blockIdx.x;
    const long long cap2 = (long long)C * max_out;
    write_result_rows(cls, boxes, proposed, counts, kept, payload, b, A, C, max_out, w.s2_count[b],
                      [&](int r, int &a, int &c) {
                          const int p = w.s2_sel[(long long)b * max_out + r];
                          a = w.s2_anchor[b * cap2 + p];
                          c = w.s2_class[b * cap2 + p];
                      });
}

// ------------------------------------------------------------------ MaskDistribute + level slots
__global__ void mask_distribute_kernel(const float *__restrict__ rows, int rs, int has_k, float *__restrict__ kvals,
                                       int *__restrict__ level_slots, int *__restrict__ level_counts,
                                       int *__restrict__ level_max, int cap, int max_k, float base_size) {
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
        if (level_max) atomicMax(&level_max[lvl], n);    // MoldBatch's per-level second axis (misc.py:235-236); zeroed by the launcher
        for (int i = n; i < cap; ++i) level_slots[((long long)b * L + lvl) * cap + i] = -1;
    }
}

// ------------------------------------------------------------------ crop_and_resize + MoldBatch(-1)
// T = float, or _Float16 (feature maps and crops of the fp16-storage heads: same fp32 arithmetic on the stored values,
// one rounding at the store; the RoI BOXES stay fp32 either way)
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
template <class T>
__global__ void __launch_bounds__(256)
roi_crop_resize_kernel(const T *__restrict__ fmap, const float *__restrict__ rows, int rs, int roff,
                       const int *__restrict__ level_slots, const int *__restrict__ level_counts,
                       T *__restrict__ roi_fmaps, float *__restrict__ roi_boxes, int Hf, int Wf, int CV, int cap,
                       int L, int level, int n_l, int ch, int cw, float img_h, float img_w, int box_off,
                       int box_rows, const int *__restrict__ live) {
    constexpr int W = 16 / (int)sizeof(T);            // elements per 16-byte access
    const int j = blockIdx.x % n_l;
    const int b = blockIdx.x / n_l;
    // fixed-capacity launch (n_l = cap, no host read of the RoI counts): only the first max(1, *live) slots of a level
    // exist in the molded tensor (MoldBatch, misc.py:235-236); the others are never read by anyone
    if (live && j >= max(1, *live)) return;
    const int cnt = level_counts[b * L + level];
    const int C = CV * W;
    T *dst = roi_fmaps + ((long long)b * n_l + j) * ch * cw * C;
    float *brow = roi_boxes + ((long long)b * box_rows + box_off + j) * 6;
    const int total = ch * cw * CV;
    auto load = [&](const T *p, float (&v)[W]) {
        if constexpr (W == 4) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(p);
            v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        } else {
            const f16x8d x = *reinterpret_cast<const f16x8d *>(p);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
        }
    };
    auto store = [&](T *p, const float (&v)[W]) {
        if constexpr (W == 4) {
            const f32x4 x = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(p) = x;
        } else {
            const f16x8d x = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3],
                              (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            *reinterpret_cast<f16x8d *>(p) = x;
        }
    };
    if (j >= cnt) {  // MoldBatch padding (misc.py:276-282): -1 for features AND boxes
        float m1[W];
#pragma unroll
        for (int e = 0; e < W; ++e) m1[e] = -1.f;
        for (int i = threadIdx.x; i < total; i += blockDim.x) store(dst + (long long)i * W, m1);
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
    const T *img = fmap + (long long)b * Hf * Wf * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c = (i % CV) * W;
        const int pos = i / CV;
        const int ox = pos % cw, oy = pos / cw;
        const float in_y = ch > 1 ? y1 * (float)(Hf - 1) + (float)oy * hs : 0.5f * (y1 + y2) * (float)(Hf - 1);
        const float in_x = cw > 1 ? x1 * (float)(Wf - 1) + (float)ox * ws : 0.5f * (x1 + x2) * (float)(Wf - 1);
        float v[W];
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = 0.f;
        if (!(in_y < 0.f || in_y > (float)(Hf - 1) || in_x < 0.f || in_x > (float)(Wf - 1))) {
            const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
            const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
            const float fy = in_y - (float)ty, fx = in_x - (float)lx;
            float tl[W], tr[W], bl[W], br[W];
            load(img + ((long long)ty * Wf + lx) * C + c, tl);
            load(img + ((long long)ty * Wf + rx) * C + c, tr);
            load(img + ((long long)by * Wf + lx) * C + c, bl);
            load(img + ((long long)by * Wf + rx) * C + c, br);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float top = tl[e] + (tr[e] - tl[e]) * fx;
                const float bot = bl[e] + (br[e] - bl[e]) * fx;
                v[e] = top + (bot - top) * fy;
            }
        }
        store(dst + (long long)i * W, v);
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
                                         int32_t *kept, float *gather_payload, int32_t B, int32_t A, int32_t C,
                                         float min_confidence, float nms_iou, float post_iou, int32_t max_out,
                                         void *workspace, void *stream) {
    ML_REQUIRE(cls_pred && boxes && proposed && counts && workspace, "detection_proposal: null pointer");
    ML_REQUIRE(B > 0 && A > 0 && C > 0 && C <= 64 && max_out > 0, "detection_proposal: bad dims (C <= 64)");
    ML_REQUIRE((long long)A * C < (1ll << 31), "detection_proposal: A*C overflows the first-occurrence key");
    ML_REQUIRE((long long)B * C * max_out < (1ll << 31), "detection_proposal: B*C*max_out overflows int32");
    ML_REQUIRE(ml_aligned16(boxes) && (((uintptr_t)workspace) & 255) == 0, "detection_proposal: alignment");
    const size_t nms_lds = (size_t)NMS_BAND * 16 + (size_t)max_out * 16 + (size_t)NMS_BAND * 10 + sizeof(NmsShared);
    ML_REQUIRE(nms_lds <= 160 * 1024, "detection_proposal: max_out %d too large for the NMS LDS budget", max_out);
    long long bytes = 0;
    DetWs w = det_ws_carve(workspace, B, A, C, max_out, &bytes);
    hipStream_t s = (hipStream_t)stream;
    const int BC = B * C;
    hipLaunchKernelGGL(det_init_kernel, dim3((BC + 255) / 256), dim3(256), 0, s, w, BC);
    const long long per_image = (long long)A * C;
    hipLaunchKernelGGL(det_threshold_kernel, dim3((unsigned)((per_image + DET_EPB - 1) / DET_EPB), B), dim3(256), 0, s,
                       cls_pred, boxes, w, A, C, min_confidence);
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = ml_ensure_dynamic_lds(reinterpret_cast<const void *>(det_nms_bucket_kernel), 160 * 1024, lds_ok,
                                       "detection_proposal"))
        return rc;
    const NmsIo io1 = {w.bucket_count, w.keys, w.cbox, w.s1_anchor, w.s1_count, (long long)A};
    hipLaunchKernelGGL(det_nms_bucket_kernel, dim3(BC), dim3(NMS_T), nms_lds, s, io1, max_out, nms_iou, min_confidence);
    if (C * max_out <= 2048) {
        long long p2 = 64;
        while (p2 < (long long)C * max_out) p2 <<= 1;
        const long long lds2 = (long long)C * max_out * 24 + (long long)max_out * 20 + p2 * 8;
        hipLaunchKernelGGL(det_nms_image_kernel, dim3(B), dim3(512), (size_t)lds2, s, cls_pred, boxes, w, proposed, counts,
                           kept, gather_payload, A, C, max_out, post_iou);
    } else {
        hipLaunchKernelGGL(det_stage2_gather_kernel, dim3(B), dim3(256), 0, s, cls_pred, boxes, w, A, C, max_out);
        const NmsIo io2 = {w.s2_n, w.s2_keys, w.s2_cbox, w.s2_sel, w.s2_count, (long long)C * max_out};
        hipLaunchKernelGGL(det_nms_bucket_kernel, dim3(B), dim3(NMS_T), nms_lds, s, io2, max_out, post_iou, min_confidence);
        hipLaunchKernelGGL(det_rows_kernel, dim3(B), dim3(256), 0, s, cls_pred, boxes, w, proposed, counts, kept,
                           gather_payload, A, C, max_out);
    }
    ML_CHECK_LAUNCH("detection_proposal");
    return ML_OK;
}

extern "C" int ml_mask_distribute_i32(const float *rows, int32_t row_stride, int32_t has_k, float *kvals,
                                      int32_t *level_slots, int32_t *level_counts, int32_t *level_max, int32_t B,
                                      int32_t cap, int32_t max_k, float base_size, void *stream) {
    ML_REQUIRE(rows && level_slots && level_counts && B > 0 && cap > 0, "mask_distribute: bad arguments");
    ML_REQUIRE(row_stride >= (has_k ? 5 : 4), "mask_distribute: row_stride too small");
    ML_REQUIRE(max_k >= 0 && max_k < 64, "mask_distribute: max_k out of range");
    if (level_max && hipMemsetAsync(level_max, 0, (size_t)(max_k + 1) * 4, (hipStream_t)stream) != hipSuccess) {
        ml_set_error("mask_distribute: hipMemsetAsync failed");
        return ML_E_LAUNCH;
    }
    hipLaunchKernelGGL(mask_distribute_kernel, dim3(B), dim3(128), (size_t)cap * 4, (hipStream_t)stream, rows, row_stride,
                       has_k, kvals, level_slots, level_counts, level_max, cap, max_k, base_size);
    ML_CHECK_LAUNCH("mask_distribute");
    return ML_OK;
}

template <class T>
static int roi_crop_launch(const T *fmap, const float *rows, int32_t row_stride, int32_t row_off, const int32_t *level_slots,
                           const int32_t *level_counts, T *roi_fmaps, float *roi_boxes, int32_t B, int32_t Hf, int32_t Wf,
                           int32_t C, int32_t cap, int32_t L, int32_t level, int32_t n_l, int32_t ch, int32_t cw, float img_h,
                           float img_w, int32_t box_off, int32_t box_rows, const int32_t *live, void *stream) {
    constexpr int W = 16 / (int)sizeof(T);
    ML_REQUIRE(fmap && rows && level_slots && level_counts && roi_fmaps && roi_boxes, "roi_crop: null pointer");
    ML_REQUIRE(row_off >= 0 && row_off + 6 <= row_stride, "roi_crop: rows must hold 6 columns from row_off");
    ML_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && C % W == 0 && n_l > 0 && ch > 0 && cw > 0, "roi_crop: bad dims (C %% %d)", W);
    ML_REQUIRE(level >= 0 && level < L && n_l <= cap && box_off >= 0 && box_off + n_l <= box_rows, "roi_crop: bad level/rows");
    ML_REQUIRE(ml_aligned16(fmap) && ml_aligned16(roi_fmaps), "roi_crop: 16-byte alignment");
    hipLaunchKernelGGL(roi_crop_resize_kernel<T>, dim3((unsigned)(B * n_l)), dim3(256), 0, (hipStream_t)stream, fmap, rows,
                       row_stride, row_off, level_slots, level_counts, roi_fmaps, roi_boxes, Hf, Wf, C / W, cap, L, level, n_l, ch, cw, img_h,
                       img_w, box_off, box_rows, live);
    ML_CHECK_LAUNCH("roi_crop");
    return ML_OK;
}

extern "C" int ml_roi_crop_resize_f32(const float *fmap, const float *rows, int32_t row_stride, int32_t row_off,
                                      const int32_t *level_slots,
                                      const int32_t *level_counts, float *roi_fmaps, float *roi_boxes, int32_t B,
                                      int32_t Hf, int32_t Wf, int32_t C, int32_t cap, int32_t L, int32_t level,
                                      int32_t n_l, int32_t ch, int32_t cw, float img_h, float img_w, int32_t box_off,
                                      int32_t box_rows, const int32_t *live, void *stream) {
    return roi_crop_launch<float>(fmap, rows, row_stride, row_off, level_slots, level_counts, roi_fmaps, roi_boxes, B, Hf, Wf, C,
                                  cap, L, level, n_l, ch, cw, img_h, img_w, box_off, box_rows, live, stream);
}

extern "C" int ml_roi_crop_resize_f16(const void *fmap, const float *rows, int32_t row_stride, int32_t row_off,
                                      const int32_t *level_slots, const int32_t *level_counts, void *roi_fmaps,
                                      float *roi_boxes, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t cap, int32_t L,
                                      int32_t level, int32_t n_l, int32_t ch, int32_t cw, float img_h, float img_w,
                                      int32_t box_off, int32_t box_rows, const int32_t *live, void *stream) {
    return roi_crop_launch<_Float16>(reinterpret_cast<const _Float16 *>(fmap), rows, row_stride, row_off, level_slots,
                                     level_counts, reinterpret_cast<_Float16 *>(roi_fmaps), roi_boxes, B, Hf, Wf, C, cap, L,
                                     level, n_l, ch, cw, img_h, img_w, box_off, box_rows, live, stream);
}

// ------------------------------------------------------------------ MoldBatch of a fixed-capacity stage 2
// The forward without a host read keeps every RoI level at capacity: src [B, L*cap, E] (level l's RoIs at rows
// l*cap ..).  Once the host knows the per-level maxima n_l (ONE read, after the whole forward has been enqueued) this
// copies rows [l*cap, l*cap + n_l) of every image next to each other: dst [B, sum n_l, E] -- the reference's
// Concatenate(axis=1) of the molded levels (engine/layers/instance.py:222-225).  E % 4 == 0: 16 bytes per lane.
namespace {
struct MoldArgs { int n_l[8], dst_off[8], L, cap, total; };
__global__ void __launch_bounds__(256) mold_levels_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int E4, MoldArgs A,
                                                           long long work) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= work) return;
    const int e = (int)(idx % E4);
    long long r = idx / E4;
    const int row = (int)(r % A.total);
    const int b = (int)(r / A.total);
    int l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < A.L && row >= A.dst_off[k]) l = k;
    const int j = row - A.dst_off[l];
    dst[((long long)b * A.total + row) * E4 + e] = src[((long long)b * A.L * A.cap + (long long)l * A.cap + j) * E4 + e];
}
// The same concatenation with the level sizes read ON THE DEVICE (the fixed-capacity forward under a hipGraph: no host value
// may enter the launch): n_l = max(1, lmax[l]) as the host path computes them; every source slot is visited, slots past
// its level's n_l write nothing.  dst holds B * total * E floats at its front (total = sum n_l <= L * cap).
template <int V>
__global__ void __launch_bounds__(256) mold_levels_dev_kernel(const float *__restrict__ src, float *__restrict__ dst, int EV, int B, int L,
                                                               int cap, const int *__restrict__ lmax, long long work) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= work) return;
    const int e = (int)(idx % EV);
    long long r = idx / EV;
    const int j = (int)(r % cap);
    r /= cap;
    const int l = (int)(r % L), b = (int)(r / L);
    int off = 0, total = 0, n_mine = 0;
    for (int k = 0; k < L; ++k) {                     // (L <= 8 uniform loads)
        const int n = min(max(lmax[k], 1), cap);
        if (k == l) { off = total; n_mine = n; }
        total += n;
    }
    if (j >= n_mine) return;
    reinterpret_cast<vec *>(dst)[((long long)b * total + off + j) * EV + e] =
        reinterpret_cast<const vec *>(src)[(((long long)b * L + l) * cap + j) * EV + e];
}
}  // namespace

extern "C" int ml_mold_levels_dev_f32(const float *src, float *dst, int32_t B, int32_t L, int32_t cap, int64_t E,
                                      const int32_t *lmax_dev, void *stream) {
    ML_REQUIRE(src && dst && lmax_dev && B > 0 && L >= 1 && L <= 8 && cap > 0 && E > 0, "mold_levels_dev: bad arguments (1..8 levels)");
    const bool v4 = E % 4 == 0 && ml_aligned16(src) && ml_aligned16(dst);
    const long long EV = v4 ? E / 4 : E;
    ML_REQUIRE(EV < (1ll << 31), "mold_levels_dev: rows too long");
    const long long work = (long long)B * L * cap * EV;
    ML_REQUIRE((work + 255) / 256 < (1ll << 31), "mold_levels_dev: too many elements");
    const dim3 grid((unsigned)((work + 255) / 256));
    if (v4) hipLaunchKernelGGL(mold_levels_dev_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (int)EV, B, L, cap, lmax_dev, work);
    else hipLaunchKernelGGL(mold_levels_dev_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (int)EV, B, L, cap, lmax_dev, work);
    ML_CHECK_LAUNCH("mold_levels_dev");
    return ML_OK;
}

extern "C" int ml_mold_levels_f32(const float *src, float *dst, int32_t B, int32_t L, int32_t cap, int64_t E,
                                  const int32_t *n_l, void *stream) {
    ML_REQUIRE(src && dst && n_l && B > 0 && L >= 1 && L <= 8 && cap > 0 && E > 0 && E % 4 == 0,
               "mold_levels: bad arguments (1..8 levels, E %% 4 == 0)");
    ML_REQUIRE(ml_aligned16(src) && ml_aligned16(dst), "mold_levels: 16-byte alignment");
    MoldArgs A;
    A.L = L; A.cap = cap; A.total = 0;
    for (int l = 0; l < 8; ++l) { A.n_l[l] = 0; A.dst_off[l] = 0; }
    for (int l = 0; l < L; ++l) {
        ML_REQUIRE(n_l[l] >= 1 && n_l[l] <= cap, "mold_levels: level %d keeps %d of %d slots", l, n_l[l], cap);
        A.n_l[l] = n_l[l];
        A.dst_off[l] = A.total;
        A.total += n_l[l];
    }
    ML_REQUIRE(E / 4 < (1ll << 31), "mold_levels: rows too long");
    const long long work = (long long)B * A.total * (E / 4);
    hipLaunchKernelGGL(mold_levels_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4 *>(src), reinterpret_cast<f32x4 *>(dst), (int)(E / 4), A, work);
    ML_CHECK_LAUNCH("mold_levels");
    return ML_OK;
}
