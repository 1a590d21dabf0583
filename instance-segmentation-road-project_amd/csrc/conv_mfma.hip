// Implicit-GEMM NHWC convolution on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   M = B*Ho*Wo output pixels, N = output channels, K = taps x span_pad.
//   A (pixels x K) is gathered straight from the NHWC activation: for one tap the `span`
//   input channels of a pixel are contiguous, so a K-chunk of 32 floats is one 128-byte line
//   per output pixel -- no im2col buffer.  Out-of-image taps are zero-filled in registers.
//   B (N x K) is the host-packed weight matrix, k contiguous.
//   Block = 256 threads = 4 waves; wave tile = TM x TN MFMA tiles of 32x32; K-chunk = 32 floats,
//   staged global -> registers -> LDS (row stride 36 floats => conflict-free ds_read_b128),
//   double buffered so the next chunk's global loads fly under the current chunk's 64 MFMAs.
//   Lane (r = lane&31, h = lane>>5) reads 4 consecutive k per ds_read_b128; MFMA step j pairs
//   k = 8*ks + j (h=0) with k = 8*ks + 4 + j (h=1) -- A and B use the same pairing, so the sum
//   over K is unchanged.  Result = a k-ordered fp32 fma chain (exact fp32, no reduced precision).
//   Epilogue: + bias (folded BatchNorm) + optional residual + activation, written to a
//   channel slice of the destination buffer (concat fusion) or 2x2 pixel-shuffled
//   (Conv2DTranspose).
//   Block -> tile map is XCD aware: the NB column tiles that share one 128-pixel A panel get
//   ids congruent mod 8, i.e. the same XCD / L2.
#include "common.h"

namespace {

constexpr int LDS_LD = 36;  // floats per staged row (32 + 4 pad)

template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ void __launch_bounds__(256)
conv_mfma_kernel(const ml_conv2d_desc p, int MB, int NB, int M, int ncpt, int ktot) {
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    constexpr int A_LD = BM / 32;  // float4 loads per thread per chunk
    constexpr int B_LD = BN / 32;
    constexpr int BUF = (BM + BN) * LDS_LD;
    extern __shared__ __align__(16) float lds[];

    // ---- XCD-aware tile assignment
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int jj = id >> 3;
    const int mt = (jj / NB) * 8 + xcd;
    const int nt = jj % NB;
    if (mt >= MB) return;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int ld_row = tid >> 3;
    const int ld_c = (tid & 7) * 4;

    // ---- per-thread pixel coordinates of the A rows it stages
    int a_iy0[A_LD], a_ix0[A_LD], a_pix[A_LD];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + ld_row + 32 * i;
        if (m < M) {
            const int b = m / HoWo;
            const int r = m - b * HoWo;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
            a_pix[i] = b * p.H * p.W;
        } else {
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
            a_pix[i] = 0;
        }
    }
    const int gofs = p.in_coff + nt * p.group_cin_step;
    const float *wrow = p.wgt + (size_t)(n0 + ld_row) * ktot + ld_c;

    f32x4 areg[A_LD], breg[B_LD];
    int kh = 0, kw = 0, cc = 0;  // state of the NEXT chunk to load
    const int nchunks = p.KH * p.KW * ncpt;

    auto load_chunk = [&](int kc) {
        const int c = cc * 32 + ld_c;
        const int px = c >> p.cpp_shift;
        const int dy = kh * p.dil, dx = kw * p.dil;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_iy0[i] + dy;
            const int ix = a_ix0[i] + dx;
            const bool ok = ((unsigned)iy < (unsigned)p.H) && ((unsigned)(ix + px) < (unsigned)p.W) &&
                            (c < p.span);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const long long off = (long long)(a_pix[i] + iy * p.W + ix) * (long long)p.in_cstride + gofs + c;
                v = *reinterpret_cast<const f32x4 *>(p.in + off);
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            breg[i] = *reinterpret_cast<const f32x4 *>(wrow + (size_t)(32 * i) * ktot + (size_t)kc * 32);
        // advance (kh,kw,cc)
        if (++cc == ncpt) {
            cc = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    auto store_chunk = [&](int buf) {
        float *As = lds + buf * BUF;
        float *Bs = As + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<f32x4 *>(As + (ld_row + 32 * i) * LDS_LD + ld_c) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (ld_row + 32 * i) * LDS_LD + ld_c) = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    const int r = lane & 31;
    const int h = lane >> 5;
    const int a_off = (wm * TM * 32 + r) * LDS_LD + h * 4;
    const int b_off = BM * LDS_LD + (wn * TN * 32 + r) * LDS_LD + h * 4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_chunk(kc + 1);
        const float *base = lds + buf * BUF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[mi] = *reinterpret_cast<const f32x4 *>(base + a_off + mi * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                b[ni] = *reinterpret_cast<const f32x4 *>(base + b_off + ni * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        if (kc + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int co = p.shuffle2x2 ? (p.cout >> 2) : p.cout;  // channels per output pixel
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn * TN * 32 + ni * 32 + r;
        if (n >= p.cout) continue;
        int ab = 0, o = n;
        if (p.shuffle2x2) { ab = n / co; o = n - ab * co; }
        const float bv = p.bias ? p.bias[o] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = m0 + wm * TM * 32 + mi * 32 + row;
                if (m >= M) continue;
                float v = acc[mi][ni][e] + bv;
                if (p.residual) v += p.residual[(size_t)m * p.res_cstride + p.res_coff + n];
                v = ml_apply_act(v, p.act);
                if (p.shuffle2x2) {
                    const int b = m / HoWo;
                    const int rr = m - b * HoWo;
                    const int oy = rr / p.Wo;
                    const int ox = rr - oy * p.Wo;
                    const size_t opix = ((size_t)b * (2 * p.Ho) + 2 * oy + (ab >> 1)) * (size_t)(2 * p.Wo) + 2 * ox + (ab & 1);
                    p.out[opix * p.out_cstride + p.out_coff + o] = v;
                } else if (p.out_bstride) {
                    const int b = m / HoWo;
                    p.out[(size_t)b * (size_t)p.out_bstride + (size_t)(m - b * HoWo) * p.out_cstride + p.out_coff + o] = v;
                } else {
                    p.out[(size_t)m * p.out_cstride + p.out_coff + o] = v;
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
int launch_conv(const ml_conv2d_desc &d, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    constexpr int LDS_BYTES = 2 * (BM + BN) * LDS_LD * 4;
    auto kern = conv_mfma_kernel<WAVES_M, WAVES_N, TM, TN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            ml_set_error("conv2d: hipFuncSetAttribute(%d B LDS) failed: %s", LDS_BYTES, hipGetErrorString(e));
            return ML_E_LAUNCH;
        }
        attr_set = true;
    }
    const long long M = (long long)d.B * d.Ho * d.Wo;
    const int MB = (int)((M + BM - 1) / BM);
    const int NB = d.n_pad / BN;
    const int ncpt = d.span_pad / 32;
    const int ktot = d.KH * d.KW * d.span_pad;
    const long long grid = (long long)((MB + 7) / 8) * 8 * NB;
    ML_REQUIRE(grid > 0 && grid < (1ll << 31), "conv2d: grid %lld out of range", grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, d, MB, NB, (int)M, ncpt, ktot);
    ML_CHECK_LAUNCH("conv2d");
    return ML_OK;
}

int pick_tile(int cout, int tile) {
    if (tile >= 1 && tile <= 3) return tile;
    if (cout <= 32) return 3;
    if (cout <= 64) return 2;
    if (cout <= 96) return 3;
    return 1;
}

}  // namespace

extern "C" int ml_conv2d_ntile(int32_t cout, int32_t tile) {
    const int t = pick_tile(cout, tile);
    return t == 1 ? 128 : (t == 2 ? 64 : 32);
}

extern "C" int ml_conv2d_f32(const ml_conv2d_desc *dp, void *stream) {
    ML_REQUIRE(dp != nullptr, "conv2d: null descriptor");
    const ml_conv2d_desc &d = *dp;
    ML_REQUIRE(d.in && d.wgt && d.out, "conv2d: null tensor pointer");
    ML_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0, "conv2d: bad spatial dims");
    ML_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0, "conv2d: bad kernel geometry");
    ML_REQUIRE(d.span > 0 && d.span % 4 == 0, "conv2d: span %d must be a positive multiple of 4", d.span);
    ML_REQUIRE(d.span_pad == (d.span + 31) / 32 * 32, "conv2d: span_pad %d != ceil32(span %d)", d.span_pad, d.span);
    ML_REQUIRE(d.in_cstride % 4 == 0 && d.in_coff % 4 == 0 && d.group_cin_step % 4 == 0,
               "conv2d: input channel stride/offset must be multiples of 4 (16-byte loads)");
    ML_REQUIRE(ml_aligned16(d.in) && ml_aligned16(d.wgt), "conv2d: in/wgt must be 16-byte aligned");
    ML_REQUIRE(d.cpp_shift >= 0 && d.cpp_shift <= 30, "conv2d: bad cpp_shift");
    ML_REQUIRE(d.cout > 0 && d.out_cstride > 0 && d.out_coff >= 0, "conv2d: bad output channels");
    ML_REQUIRE((long long)d.B * d.H * d.W < (1ll << 31) / 2, "conv2d: too many input pixels for int32 indexing");
    ML_REQUIRE((long long)d.B * d.Ho * d.Wo < (1ll << 31) - 256, "conv2d: too many output pixels");
    if (d.shuffle2x2) {
        ML_REQUIRE(d.cout % 4 == 0 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.Ho == d.H && d.Wo == d.W,
                   "conv2d: shuffle2x2 needs a 1x1 stride-1 problem with cout = 4*Cout");
        ML_REQUIRE(d.out_coff + d.cout / 4 <= d.out_cstride, "conv2d: output slice exceeds buffer channels");
        ML_REQUIRE(d.residual == nullptr, "conv2d: shuffle2x2 does not take a residual");
    } else {
        ML_REQUIRE(d.out_coff + d.cout <= d.out_cstride, "conv2d: output slice exceeds buffer channels");
    }
    if (d.residual) ML_REQUIRE(d.res_coff + d.cout <= d.res_cstride, "conv2d: residual slice exceeds buffer");
    if (d.cpp_shift == 30) {
        ML_REQUIRE(d.in_coff + d.span <= d.in_cstride || d.group_cin_step > 0,
                   "conv2d: input slice exceeds buffer channels");
    }
    const int t = pick_tile(d.cout, d.tile);
    const int bn = t == 1 ? 128 : (t == 2 ? 64 : 32);
    ML_REQUIRE(d.n_pad >= d.cout && d.n_pad % bn == 0, "conv2d: n_pad %d must be a multiple of the N tile %d", d.n_pad, bn);
    if (d.group_cin_step) {
        ML_REQUIRE(t == 3, "conv2d: grouped mode needs the 32-wide N tile");
        ML_REQUIRE(d.in_coff + (d.n_pad / 32 - 1) * d.group_cin_step + d.span <= d.in_cstride,
                   "conv2d: grouped input slices exceed buffer channels");
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (t) {
        case 1: return launch_conv<2, 2, 2, 2>(d, s);
        case 2: return launch_conv<2, 2, 2, 1>(d, s);
        default: return launch_conv<4, 1, 1, 1>(d, s);
    }
}
