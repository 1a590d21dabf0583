// Shared helpers for the gfx950 kernels of libmasklab_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "masklab_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

void ml_set_error(const char *fmt, ...);

#define ML_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ml_set_error(__VA_ARGS__);        \
            return ML_E_BADARG;               \
        }                                     \
    } while (0)

#define ML_CHECK_LAUNCH(what)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            ml_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));   \
            return ML_E_LAUNCH;                                                   \
        }                                                                         \
    } while (0)

__device__ __forceinline__ float ml_apply_act(float v, int act) {
    if (act == ML_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ML_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    if (act == ML_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// ML_MATH_F32X3: 8 fp32 values -> 8 halves `hi` (round to nearest: v_cvt_pk_f16_f32) and 8 halves `lo` = (x - hi) * 2^11,
// rounded to nearest.  x - hi is exact in fp32 (at most 13 significant bits), |lo| <= |x|, and the scaling keeps lo a NORMAL
// half wherever x is one, so |x - (hi + 2^-11 lo)| <= 2^-22 |x| for 2^-14 <= |x| < 65520 (smaller: 2^-36; beyond: hi overflows, the result is
// Inf / NaN -- loud, as in the fp16-storage mode).  16 VALU instructions: 4 packed multiplies (x * 2^11), 4 packed converts,
// 8 v_fma_mix (f16 hi * -2^11 + the scaled x, result stored as f16).
__device__ __forceinline__ void split_hi_lo(const f32x4 x0, const f32x4 x1, const float neg_scale, f16x8 &hi, f16x8 &lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x4 s0 = x0 * 2048.f, s1 = x1 * 2048.f;
    unsigned hw[4], lw[4];
    hw[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0[0], x0[1]}, f16x2));
    hw[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0[2], x0[3]}, f16x2));
    hw[2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x1[0], x1[1]}, f16x2));
    hw[3] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x1[2], x1[3]}, f16x2));
    const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lw[i]) : "v"(hw[i]), "s"(neg_scale), "v"(sv[2 * i]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "+v"(lw[i]) : "v"(hw[i]), "s"(neg_scale), "v"(sv[2 * i + 1]));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 H = {hw[0], hw[1], hw[2], hw[3]}, L = {lw[0], lw[1], lw[2], lw[3]};
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
}

// the same split for two values (packed halves in one register each)
__device__ __forceinline__ void split_hi_lo_pair(const float xa, const float xb, const float neg_scale, unsigned &hi, unsigned &lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 x = {xa, xb};
    const f32x2 sc = x * 2048.f;
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "s"(neg_scale), "v"(sc[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(neg_scale), "v"(sc[1]));
}

static inline bool ml_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: `done` (one per kernel) remembers
// in bit d that device d has it.  Host threads that race here both set the same value (idempotent), so an
// atomic mask is all the synchronisation needed; there is no other mutable state in the library.
#include <atomic>
static inline int ml_ensure_dynamic_lds(const void *fn, int bytes, std::atomic<unsigned long long> &done,
                                        const char *what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        ml_set_error("%s: hipGetDevice failed or device index %d out of range", what, dev);
        return ML_E_LAUNCH;
    }
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return ML_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        ml_set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, bytes, hipGetErrorString(e));
        return ML_E_LAUNCH;
    }
    done.fetch_or(bit, std::memory_order_release);
    return ML_OK;
}

// Blocks a persistent kernel keeps resident: `per_cu` per compute unit of the CURRENT device, rounded down to a
// multiple of 32 (the persistent kernels map work units to blocks modulo small powers of two), at least 32.  The
// CU count is read once per device (256 on MI355X in SPX mode; fewer in partitioned modes).  Only scheduling depends on
// it, never which values are computed or in which order.
static inline int ml_resident_blocks(int per_cu) {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256 * per_cu;
    int n = cus[dev].load(std::memory_order_acquire);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_release);
    }
    const int r = n * per_cu / 32 * 32;
    return r < 32 ? 32 : r;
}

