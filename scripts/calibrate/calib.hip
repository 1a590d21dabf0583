// Calibration program (measurement infrastructure, not part of libmasklab_hip.so):
//   1. what rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts, in the access patterns the product kernels
//      use -- (a) wide coalesced 16-B-per-lane global loads, (b) LDS-direct `buffer_load_dwordx4 ... lds` staging in the
//      8-rows-x-128-B pattern of conv_mfma.hip / conv1x1_pipe.hip (rows `pitch` bytes apart, a panel read once over its
//      K chunks), (c) the same addresses through register loads.  Every buffer is 1 GiB (> the 256 MiB Infinity Cache),
//      each byte read once and written once: no reuse to argue about.
//   2. the peaks this box really sustains: HBM copy bandwidth of those kernels, and back-to-back MFMA issue on non-zero
//      operands for v_mfma_f32_32x32x2_f32, v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16.
// Build: hipcc -O3 --offload-arch=gfx950 calib.hip -o calib     Run: ./calib            (prints one JSON object)
//        rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./calib pmc      (one launch of each copy kernel, no MFMA loops)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

// ---- (a) wide coalesced copy: a wave-instruction moves 1 KiB contiguous
__global__ void __launch_bounds__(256) copy_global_x4(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) out[i] = in[i];
}

// ---- (b) LDS-direct staging in the conv kernels' pattern.  A block owns panels of 128 rows x `pitch` bytes; per K chunk
// (128 B of every row) each of its 4 waves issues 4 loads of 8 rows x 128 B (lane -> row lane >> 3, 16-byte group lane & 7);
// the staged chunk is read back with ds_read_b128 and stored to `out` at the same address (16 B per lane).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
#endif
}

template <bool DMA>
__global__ void __launch_bounds__(256) copy_rows128(const char *__restrict__ in, char *__restrict__ out, int panels, int pitch) {
    __shared__ __align__(16) char lds[128 * 128];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld_row = tid >> 3, ld_g = tid & 7;
    const int nk = pitch / 128;
    for (int p = blockIdx.x; p < panels; p += gridDim.x) {
        const char *base = in + (long long)p * 128 * pitch;
        char *obase = out + (long long)p * 128 * pitch;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 128 * pitch, 0x00020000);
        for (int kc = 0; kc < nk; ++kc) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int voff = (ld_row + 32 * i) * pitch + ld_g * 16;
                if constexpr (DMA) {
                    lds_dma16(rs, lds + (32 * i + 8 * wave) * 128, voff, kc * 128);
                } else {
                    v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, kc * 128, 0));
                }
            }
            if constexpr (DMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4 *>(lds + (ld_row + 32 * i) * 128 + ld_g * 16);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4 *>(obase + (long long)(ld_row + 32 * i) * pitch + kc * 128 + ld_g * 16) = v[i];
        }
    }
}

// ---- MFMA issue peaks: 4 independent accumulators per wave, operands in registers, non-zero data
template <int KIND>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters, float seed) {
    const float a0 = seed + 0.001f * (float)(threadIdx.x & 63), b0 = 1.0f - 0.002f * (float)(threadIdx.x & 31);
    if constexpr (KIND == 0) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0 + k, b0, acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < 16; ++e) s += acc[k][e];
        out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(a0 * 0.1f + 0.01f * e); b[e] = (_Float16)(b0 * 0.1f - 0.01f * e); }
        if constexpr (KIND == 1) {
            f32x16 acc[4];
            for (int k = 0; k < 4; ++k)
                for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
            }
            float s = 0.f;
            for (int k = 0; k < 4; ++k)
                for (int e = 0; e < 16; ++e) s += acc[k][e];
            out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
        } else {
            f32x4 acc[8];
            for (int k = 0; k < 8; ++k)
                for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
            }
            float s = 0.f;
            for (int k = 0; k < 8; ++k)
                for (int e = 0; e < 4; ++e) s += acc[k][e];
            out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
        }
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv) {
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const long long BYTES = 1ll << 30;
    char *in, *out;
    CHECK(hipMalloc(&in, BYTES));
    CHECK(hipMalloc(&out, BYTES));
    {   // non-trivial contents
        std::vector<float> h(1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f - 0.5f;
        for (long long off = 0; off < BYTES; off += (long long)h.size() * 4) CHECK(hipMemcpy(in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = pmc ? 1 : 5;
    printf("{\"compute_units\": %d, \"buffer_bytes\": %lld", cus, BYTES);
    auto run_copy = [&](const char *name, auto launch) {
        launch();                                   // warm-up (also the one PMC-counted launch in pmc mode when reps == 1)
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < (pmc ? 0 : reps); ++r) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            const float ms = time_ms(e0, e1);
            if (ms < best) best = ms;
        }
        if (!pmc) printf(", \"%s\": {\"ms\": %.4f, \"read_plus_write_GBs\": %.1f}", name, best, 2.0 * BYTES / 1e6 / best);
    };
    run_copy("copy_global_x4", [&] {
        hipLaunchKernelGGL(copy_global_x4, dim3(cus * 8), dim3(256), 0, 0, (const f32x4 *)in, (f32x4 *)out, BYTES / 16);
    });
    const int pitches[3] = {128, 512, 2048};        // contiguous rows / 128 fp32 channels (512 B) / 512 fp32 channels
    for (int pi = 0; pi < 3; ++pi) {
        const int pitch = pitches[pi];
        const int panels = (int)(BYTES / (128ll * pitch));
        char name[64];
        snprintf(name, sizeof name, "copy_lds_dma_pitch%d", pitch);
        run_copy(name, [&] { hipLaunchKernelGGL(copy_rows128<true>, dim3(cus * 4), dim3(256), 0, 0, in, out, panels, pitch); });
        snprintf(name, sizeof name, "copy_regs_pitch%d", pitch);
        run_copy(name, [&] { hipLaunchKernelGGL(copy_rows128<false>, dim3(cus * 4), dim3(256), 0, 0, in, out, panels, pitch); });
    }
    if (!pmc) {
        float *sink;
        CHECK(hipMalloc(&sink, (size_t)cus * 2 * 256 * 4));
        const int iters = 20000;
        auto run_mfma = [&](const char *name, auto launch, double flop_per_wave_iter) {
            launch(100);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipEventRecord(e0));
                launch(iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                const float ms = time_ms(e0, e1);
                if (ms < best) best = ms;
            }
            const double flop = flop_per_wave_iter * iters * 4.0 * cus * 2;       // 4 waves per block, 2 blocks per CU
            printf(", \"%s\": {\"ms\": %.3f, \"TFLOPs\": %.1f}", name, best, flop / 1e9 / best);
        };
        run_mfma("mfma_f32_32x32x2_f32", [&](int n) { hipLaunchKernelGGL(mfma_loop<0>, dim3(cus * 2), dim3(256), 0, 0, sink, n, 0.5f); },
                 4.0 * 2.0 * 32 * 32 * 2);
        run_mfma("mfma_f32_32x32x16_f16", [&](int n) { hipLaunchKernelGGL(mfma_loop<1>, dim3(cus * 2), dim3(256), 0, 0, sink, n, 0.5f); },
                 4.0 * 2.0 * 32 * 32 * 16);
        run_mfma("mfma_f32_16x16x32_f16", [&](int n) { hipLaunchKernelGGL(mfma_loop<2>, dim3(cus * 2), dim3(256), 0, 0, sink, n, 0.5f); },
                 8.0 * 2.0 * 16 * 16 * 32);
    }
    printf("}\n");
    return 0;
}
