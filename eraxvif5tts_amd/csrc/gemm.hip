// gemm.hip -- reference tile kernel (any shape, bf16 or fp32-input MFMA) + dispatch.
// The tuned 256x256 LDS-DMA kernel lives in gemm_fast.hip.
#include "gemm.h"
#include "gemm_epilogue.h"

// MFMA wrapper: weights on the MFMA row index (operand "A"), tokens on the column index (operand "B").
template <typename T> struct TileMma;
template <> struct TileMma<bf16_t> {
    // v_mfma_f32_16x16x32_bf16: lane l holds rows[l&15][k = 8*(l>>4) .. +7] of both operands
    template <int STRIDE>
    static __device__ __forceinline__ void step(const bf16_t* w_rows, const bf16_t* a_rows, int lane, f32x4 (&acc)[2][2]) {
        const int r = lane & 15, kq = (lane >> 4) * 8;
        bf16x8 wf[2], af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wf[i] = *reinterpret_cast<const bf16x8*>(w_rows + (i * 16 + r) * STRIDE + kq);
            af[i] = *reinterpret_cast<const bf16x8*>(a_rows + (i * 16 + r) * STRIDE + kq);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
};
template <> struct TileMma<float> {
    // v_mfma_f32_16x16x4_f32 (exact fp32 fma chain): lane l holds rows[l&15][k = l>>4]
    template <int STRIDE>
    static __device__ __forceinline__ void step(const float* w_rows, const float* a_rows, int lane, f32x4 (&acc)[2][2]) {
        const int r = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = w_rows[(i * 16 + r) * STRIDE + 4 * s + kq];
                af[i] = a_rows[(i * 16 + r) * STRIDE + 4 * s + kq];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
};

template <typename T> struct Vec8 {
    T v[8];
};

template <typename T> __device__ __forceinline__ void load8(Vec8<T>& dst, const T* src, bool valid) {
    if (valid) {
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<bf16x8*>(dst.v) = *reinterpret_cast<const bf16x8*>(src);
        } else {
            *reinterpret_cast<f32x4*>(dst.v) = *reinterpret_cast<const f32x4*>(src);
            *reinterpret_cast<f32x4*>(dst.v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst.v[i] = (T)0.0f;
    }
}
template <typename T> __device__ __forceinline__ void store8_lds(T* dst, const Vec8<T>& src) {
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src.v);
    } else {
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src.v);
        *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src.v + 4);
    }
}

// 64x64 output tile, 4 waves (2x2), K-chunks of 32 staged through registers into a single LDS buffer.
template <typename T, int MODE, int EPI>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmParams p) {
    constexpr int BM = 64, BN = 64, BK = 32;
    constexpr int STRIDE = sizeof(T) == 2 ? 40 : 36;  // elements; keeps 16-byte alignment of 8-element chunks
    __shared__ __attribute__((aligned(16))) T As[BM * STRIDE];
    __shared__ __attribute__((aligned(16))) T Ws[BN * STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* W = reinterpret_cast<const T*>(p.W);

    const int lr = tid >> 2, lc = (tid & 3) * 8;  // this thread stages row lr, elements lc..lc+7 of each chunk
    const int am = m0 + lr, wn_row = n0 + lr;
    const int cslices = MODE == GEMM_CONV31 ? p.conv_win / BK : 1;  // 32-channel slices of the input-channel window
    const int nk = MODE == GEMM_CONV31 ? 31 * cslices : p.K / BK;
    const int win0 = MODE == GEMM_CONV31 ? (n0 / p.conv_cg) * p.conv_cg : 0;  // first input channel this tile can see

    // per-thread source description
    int a_row = am;
    if (MODE == GEMM_DENSE && p.a_row_mod > 0) a_row = am % p.a_row_mod;
    const int L = p.rows_per_batch;
    const int a_b = MODE == GEMM_CONV31 ? am / L : 0, a_pos = MODE == GEMM_CONV31 ? am % L : 0;

    auto fetch = [&](int kk, Vec8<T>& ra, Vec8<T>& rw) {
        if constexpr (MODE == GEMM_DENSE) {
            load8(ra, A + (size_t)a_row * p.lda + kk * BK + lc, am < p.M);
            load8(rw, W + (size_t)wn_row * p.ldw + kk * BK + lc, wn_row < p.N);
        } else {
            const int tap = kk / cslices, sl = kk - tap * cslices;
            const int sp = a_pos + tap - 15;  // Conv1d(padding=15): zero outside [0, L) of this utterance
            const int ch = win0 + sl * 32 + lc;
            const bool ok = am < p.M && sp >= 0 && sp < L && ch < p.N;
            load8(ra, A + (size_t)(a_b * L + sp) * p.lda + ch, ok);
            load8(rw, W + ((size_t)tap * p.N + wn_row) * p.conv_win + sl * 32 + lc, wn_row < p.N);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Vec8<T> ra, rw;
    fetch(0, ra, rw);
    for (int kk = 0; kk < nk; ++kk) {
        store8_lds(As + lr * STRIDE + lc, ra);
        store8_lds(Ws + lr * STRIDE + lc, rw);
        __syncthreads();
        if (kk + 1 < nk) fetch(kk + 1, ra, rw);
        TileMma<T>::template step<STRIDE>(Ws + wn * 32 * STRIDE, As + wm * 32 * STRIDE, lane, acc);
        __syncthreads();
    }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int n = n0 + wn * 32 + ni * 16 + 4 * (lane >> 4);
            const int m = m0 + wm * 32 + mi * 16 + (lane & 15);
            gemm_epilogue4<T, EPI>(p, m, n, acc[ni][mi]);
        }
}

template <typename T, int MODE, int EPI> static int launch_tile(const GemmParams& p, hipStream_t stream) {
    dim3 grid(cdiv(p.M, 64), cdiv(p.N, 64));
    hipLaunchKernelGGL((gemm_tile_kernel<T, MODE, EPI>), grid, dim3(256), 0, stream, p);
    F5_LAUNCH_CHECK();
    return 0;
}

template <typename T> static int dispatch_tile(const GemmParams& p, int mode, int epi, hipStream_t stream) {
    if (mode == GEMM_DENSE) {
        switch (epi) {
            case EPI_STORE_T: return launch_tile<T, GEMM_DENSE, EPI_STORE_T>(p, stream);
            case EPI_STORE_F32: return launch_tile<T, GEMM_DENSE, EPI_STORE_F32>(p, stream);
            case EPI_RESID: return launch_tile<T, GEMM_DENSE, EPI_RESID>(p, stream);
            case EPI_ADD2: return launch_tile<T, GEMM_DENSE, EPI_ADD2>(p, stream);
            case EPI_ROPE_T: return launch_tile<T, GEMM_DENSE, EPI_ROPE_T>(p, stream);
            case EPI_GATE_T: return launch_tile<T, GEMM_DENSE, EPI_GATE_T>(p, stream);
        }
    } else {
        switch (epi) {
            case EPI_STORE_T: return launch_tile<T, GEMM_CONV31, EPI_STORE_T>(p, stream);
            case EPI_RESID: return launch_tile<T, GEMM_CONV31, EPI_RESID>(p, stream);
            case EPI_GATE_T: return launch_tile<T, GEMM_CONV31, EPI_GATE_T>(p, stream);
        }
    }
    return f5_fail(F5_EINVAL, "gemm: unsupported mode/epilogue %d/%d", mode, epi);
}

int launch_gemm_fast(const GemmParams& p, int mode, int epi, hipStream_t stream);  // gemm_fast.hip

int launch_gemm(const GemmParams& p, int precision, int mode, int epi, int kernel_kind, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (mode == GEMM_DENSE && (p.K <= 0 || p.K % 32 != 0)) return f5_fail(F5_EINVAL, "gemm: K=%d must be a positive multiple of 32", p.K);
    if (mode == GEMM_DENSE && ((p.lda & 7) || (p.ldw & 7))) return f5_fail(F5_EINVAL, "gemm: lda/ldw must be multiples of 8");
    if (mode == GEMM_CONV31) {
        if (p.N % 64 != 0 || p.rows_per_batch <= 0 || p.M % p.rows_per_batch != 0 || (p.lda & 7) || p.conv_cg <= 0 ||
            (p.conv_cg & 7) || p.conv_win % 32 != 0)
            return f5_fail(F5_EINVAL, "gemm(conv31): N must be a multiple of 64, M a whole number of sequences, group size a multiple of 8");
    }
    if ((epi == EPI_RESID || epi == EPI_ROPE_T || epi == EPI_GATE_T) && p.rows_per_batch <= 0) return f5_fail(F5_EINVAL, "gemm: rows_per_batch missing");
    if ((p.lnf_stats || p.stats_out) && kernel_kind != 1) return f5_fail(F5_ESTATE, "gemm: the LayerNorm fold exists in the tuned kernel only");
    if (kernel_kind == 1) {
        if (!gemm_fast_supported(p, precision, mode, epi)) return f5_fail(F5_EINVAL, "gemm: tuned kernel does not support this problem");
        return launch_gemm_fast(p, mode, epi, stream);
    }
    if (precision == F5_PREC_BF16) return dispatch_tile<bf16_t>(p, mode, epi, stream);
    if (precision == F5_PREC_FP32) return dispatch_tile<float>(p, mode, epi, stream);
    return f5_fail(F5_EINVAL, "gemm: bad precision %d", precision);
}
