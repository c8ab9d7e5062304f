// elementwise.hip -- HBM-bound kernels of the DiT path: LayerNorm/modulate, ConvNeXt depthwise conv + LN, GRN,
// embedding gather, timestep embedding, small-M fp32 linears (AdaLN modulation), packing, CFG + ODE step.
// All row-wise kernels use one 64-lane wavefront per row with 16-byte vector accesses and wave-shuffle reductions.
#include "kernels.h"

// ----------------------------------------------------------------------------- LayerNorm (+ modulation)
// x[r] (+= y[r], written back) -> out[r] = LN(x[r]) * (add_one + mul[b]) + add[b].  The optional y is the previous
// residual branch (gate * (attention | feed-forward) output in the activation dtype): folding the fp32 residual add into
// this streaming pass keeps the GEMM epilogues store-only and the read-modify-write fully coalesced.
// YMODE: 0 = x only; 1 = x += y, written back; 2 = (x + y) normalised but x NOT written back (the add is repeated by the next pass);
//        3 = x = (x + y) + y2, written back.  Modes 2 + 3 alternate inside a DiT block: the residual stream is written once per
//        block instead of twice, with bit-identical sums (same operands, same order).
// FULL: dim == MAXV * 256 (every lane owns MAXV whole vectors): no per-vector bounds branch, so ALL loads of a row (x, y, y2 and the
// modulation vectors) are issued before the first use -- with the branches the compiler emitted one dependent memory round trip
// per vector column (4 per row and pass).
// XI / XO: storage type of the residual stream read / written (float, or _Float16 in the bf16 production mode: the reference's own GPU
// path keeps the whole residual stream in fp16, utils_infer.py:184-193; fp32 arithmetic here either way, the fp16 store saturates).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <typename XT> __device__ __forceinline__ f32x4 load_res4(const XT* p) {
    if constexpr (sizeof(XT) == 2) {
        const f16x4 r = *reinterpret_cast<const f16x4*>(p);
        return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(p);
    }
}
template <typename XT> __device__ __forceinline__ void store_res4(XT* p, const f32x4& v) {
    if constexpr (sizeof(XT) == 2) {
        f16x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        *reinterpret_cast<f16x4*>(p) = r;
    } else {
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

// Range guard of the fp16 residual stream (the stores above SATURATE at +-65504): a lane that read, formed or stored an element at or beyond
// fp16's largest finite value -- or a NaN, which the running sum carries -- raises the plan's flag word; f5_sample reads it after the loop
// and repeats the call with fp32 residual storage (model.hip).  One compare per row on the common path.
__device__ __forceinline__ void res_range_guard(unsigned* sat, float amax, float partial_sum, int tag = 0, int row = 0) {
    const bool bad = !(amax < 65504.0f) || !(partial_sum == partial_sum);
    if (sat && __builtin_amdgcn_ballot_w64(bad) != 0ull) {  // rare: record what was seen (diagnostics read back by f5_plan_get_option)
        if (bad) {
            if (amax == amax && amax < 3.0e38f) atomicMax(sat + 1, __float_as_uint(amax));  // word 1: largest finite |element| (float bits)
            if (!(partial_sum == partial_sum)) __hip_atomic_store(sat + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // word 2: NaN seen
        }
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store(sat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(sat + 3, 1u << (tag & 15));         // word 3: which pass (bit 0 hoisted embedding, 1 first / 2 second LayerNorm of a block, 3 final)
            atomicOr(sat + 4, 1u << ((tag >> 4) & 31));  // word 4: DiT blocks (bit = index mod 32)
            atomicMax(sat + 5, 0x7fffffffu - (unsigned)row);  // word 5: 0x7fffffff - smallest offending row
        }
    }
}

template <typename TO, int MAXV, int YMODE, bool FULL, typename XI = float, typename XO = float>
__global__ __launch_bounds__(256) void layernorm_kernel(const XI* x, XO* xo /* may be x itself */, int ldx, int rows, int dim, const TO* __restrict__ y,
                                                        int ldy, const TO* __restrict__ y2, const float* __restrict__ mul, const float* __restrict__ add,
                                                        int mod_bstride, int rows_per_batch, float add_one, TO* __restrict__ out, int ldo,
                                                        unsigned* sat, int sat_tag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const XI* xr = x + (size_t)row * ldx;
    [[maybe_unused]] XO* xw = xo + (size_t)row * ldx;
    const int nvec = dim >> 2;  // dim % 4 == 0
    typedef typename std::conditional<sizeof(TO) == 2, bf16x4, f32x4>::type yvec_t;
    auto widen = [](const yvec_t& r) {
        if constexpr (sizeof(TO) == 2)
            return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        else
            return r;
    };
    const size_t moff = (size_t)(row / rows_per_batch) * mod_bstride;
    f32x4 v[MAXV], m4[MAXV], a4[MAXV];
    [[maybe_unused]] yvec_t yr[MAXV], yr2[MAXV];
    // ---- every load of the row first
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nvec) {
            v[i] = load_res4<XI>(xr + c * 4);
            if constexpr (YMODE != 0) yr[i] = *reinterpret_cast<const yvec_t*>(y + (size_t)row * ldy + c * 4);
            if constexpr (YMODE == 3) yr2[i] = *reinterpret_cast<const yvec_t*>(y2 + (size_t)row * ldy + c * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nvec) {
            m4[i] = *reinterpret_cast<const f32x4*>(mul + moff + c * 4);
            a4[i] = *reinterpret_cast<const f32x4*>(add + moff + c * 4);
        }
    }
    float s = 0.f;
    [[maybe_unused]] float amax = 0.f;  // fp16 residual storage: largest |element| read or formed in this row (range guard below)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nvec) {
            if constexpr (sizeof(XI) == 2) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
            if constexpr (YMODE != 0) {
                v[i] += widen(yr[i]);
                if constexpr (YMODE == 3) v[i] += widen(yr2[i]);
                if constexpr (YMODE != 2) store_res4<XO>(xw + c * 4, v[i]);
            }
            if constexpr (sizeof(XI) == 2 || sizeof(XO) == 2) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    if constexpr (sizeof(XI) == 2 || sizeof(XO) == 2) res_range_guard(sat, amax, s, sat_tag, row);
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + 1e-6f);
    TO* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (FULL || c < nvec) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * (add_one + m4[i][e]) + a4[i][e];
            if constexpr (sizeof(TO) == 2) {
                *reinterpret_cast<bf16x4*>(orow + c * 4) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            } else {
                *reinterpret_cast<f32x4*>(orow + c * 4) = f32x4{o[0], o[1], o[2], o[3]};
            }
        }
    }
}

// The production shape of the bf16 mode (dim = 1024, fp16 residual stream, bf16 branches and output) with 16-byte accesses: a lane owns
// two runs of 8 consecutive features.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int YMODE>
__global__ __launch_bounds__(256) void layernorm1024_h_kernel(const _Float16* x, _Float16* xo, int ldx, int rows, const bf16_t* __restrict__ y, int ldy,
                                                              const bf16_t* __restrict__ y2, const float* __restrict__ mul, const float* __restrict__ add,
                                                              int mod_bstride, int rows_per_batch, float add_one, bf16_t* __restrict__ out, int ldo,
                                                              PrefetchSet pf, unsigned* sat, int sat_tag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    // weight prefetch: one dword per 128-byte line is enough to pull the line in; lane l of the wave of row w touches line 64 w + l of every
    // range (8 KiB of lines per wave and range; the loads are issued first and waited for last)
    unsigned pfv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pfv[r] = 0u;
        const unsigned line = (unsigned)row * 64u + lane;
        if (pf.p[r] && line * 128u < pf.n[r]) pfv[r] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(pf.p[r]) + (size_t)line * 128);
    }
    if (row >= rows) return;
    const size_t moff = (size_t)(row / rows_per_batch) * mod_bstride;
    float v[2][8];
    f16x8 xr[2];
    [[maybe_unused]] bf16x8 yr[2], yr2[2];
    f32x4 m4[2][2], a4[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + i * 64) * 8;
        xr[i] = *reinterpret_cast<const f16x8*>(x + (size_t)row * ldx + c);
        if constexpr (YMODE != 0) yr[i] = *reinterpret_cast<const bf16x8*>(y + (size_t)row * ldy + c);
        if constexpr (YMODE == 3) yr2[i] = *reinterpret_cast<const bf16x8*>(y2 + (size_t)row * ldy + c);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + i * 64) * 8;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            m4[i][hh] = *reinterpret_cast<const f32x4*>(mul + moff + c + 4 * hh);
            a4[i][hh] = *reinterpret_cast<const f32x4*>(add + moff + c + 4 * hh);
        }
    }
    float s = 0.f, amax = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (float)xr[i][e];
            if constexpr (YMODE != 0) {
                amax = fmaxf(amax, fabsf(t));  // an element an earlier pass already clamped
                t += (float)yr[i][e];
            }
            if constexpr (YMODE == 3) t += (float)yr2[i][e];
            v[i][e] = t;
            amax = fmaxf(amax, fabsf(t));
        }
        if constexpr (YMODE == 1 || YMODE == 3) {
            f16x8 w;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (_Float16)__builtin_amdgcn_fmed3f(v[i][e], -65504.0f, 65504.0f);
            *reinterpret_cast<f16x8*>(xo + (size_t)row * ldx + (lane + i * 64) * 8) = w;
        }
        // the same pairing of the sum as the 4-wide kernel: ((a0 + a1) + (a2 + a3)) per group of four
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        s += (v[i][4] + v[i][5]) + (v[i][6] + v[i][7]);
    }
    res_range_guard(sat, amax, s, sat_tag, row);
    const float mean = wave_sum(s) / 1024.0f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) / 1024.0f + 1e-6f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[i][e] - mean) * rstd * (add_one + m4[i][e >> 2][e & 3]) + a4[i][e >> 2][e & 3]);
        *reinterpret_cast<bf16x8*>(out + (size_t)row * ldo + (lane + i * 64) * 8) = o;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(pfv[r]));  // keeps the prefetch loads alive
}
// The read-only pass of the in-place-residual mode (YMODE 0) with ONE modulation row for every token (the sampler: one time per evaluation):
// a wave normalises RPW consecutive rows and keeps the 2 x 1024 modulation values in registers across them -- per row the kernel above issues
// 2 loads of the stream and 8 of the modulation rows (L1 hits, but 4/5 of its load instructions).  Same arithmetic per row, bit for bit.
template <int RPW>
__global__ __launch_bounds__(256) void layernorm1024_h_rows_kernel(const _Float16* __restrict__ x, int ldx, int rows, const float* __restrict__ mul,
                                                                   const float* __restrict__ add, float add_one, bf16_t* __restrict__ out, int ldo,
                                                                   PrefetchSet pf, unsigned* sat, int sat_tag) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    unsigned pfv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pfv[r] = 0u;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const unsigned line = (unsigned)(row0 + k) * 64u + lane;
            if (pf.p[r] && line * 128u < pf.n[r]) pfv[r] ^= *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(pf.p[r]) + (size_t)line * 128);
        }
    }
    if (row0 >= rows) return;
    f16x8 xr[RPW][2];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int row = row0 + k < rows ? row0 + k : rows - 1;  // (rows past the end are computed on the last row and not stored)
#pragma unroll
        for (int i = 0; i < 2; ++i) xr[k][i] = *reinterpret_cast<const f16x8*>(x + (size_t)row * ldx + (lane + i * 64) * 8);
    }
    f32x4 m4[2][2], a4[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (lane + i * 64) * 8;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            m4[i][hh] = *reinterpret_cast<const f32x4*>(mul + c + 4 * hh);
            a4[i][hh] = *reinterpret_cast<const f32x4*>(add + c + 4 * hh);
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int row = row0 + k;
        float v[2][8];
        float s = 0.f, amax = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)xr[k][i][e];
                amax = fmaxf(amax, fabsf(v[i][e]));
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            s += (v[i][4] + v[i][5]) + (v[i][6] + v[i][7]);
        }
        res_range_guard(sat, amax, s, sat_tag, row < rows ? row : rows - 1);
        const float mean = wave_sum(s) / 1024.0f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        const float rstd = rsqrtf(wave_sum(q) / 1024.0f + 1e-6f);
        if (row < rows) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[i][e] - mean) * rstd * (add_one + m4[i][e >> 2][e & 3]) + a4[i][e >> 2][e & 3]);
                *reinterpret_cast<bf16x8*>(out + (size_t)row * ldo + (lane + i * 64) * 8) = o;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(pfv[r]));  // keeps the prefetch loads alive
}
int g_ln_rows_min = 16384;  // tuning knob ("ln_rows_min"): smallest launch (token rows) that takes the multi-row kernel
int g_ln_rows = 2;  // tuning knob ("ln_rows"): rows per wave of the read-only LayerNorm pass (1 = the one-row kernel; same-box A/B at C2, pair of passes: 96.4 us -> 90.8 with 2, 93.3 with 4)
int g_ln_wide = 1;  // tuning knob ("ln_wide"): 16-byte form of the LayerNorm pass at its production shape

template <typename TO, int MAXV, typename XI = float, typename XO = float>
static void ln_launch(const void* x, void* xo, int ldx, int rows, int dim, const void* y, int ldy, const void* y2, int ymode, const float* mul,
                      const float* add, int mod_bstride, int rows_per_batch, float one, void* out, int ldo, hipStream_t stream, unsigned* sat, int sat_tag) {
    dim3 grid(cdiv(rows, 4)), block(256);
#define F5_LN_CASE(M)                                                                                                                        \
    do {                                                                                                                                     \
        if (dim == MAXV * 256)                                                                                                               \
            hipLaunchKernelGGL((layernorm_kernel<TO, MAXV, M, true, XI, XO>), grid, block, 0, stream, (const XI*)x, (XO*)xo, ldx, rows, dim, \
                               (const TO*)y, ldy, (const TO*)y2, mul, add, mod_bstride, rows_per_batch, one, (TO*)out, ldo, sat, sat_tag);   \
        else                                                                                                                                 \
            hipLaunchKernelGGL((layernorm_kernel<TO, MAXV, M, false, XI, XO>), grid, block, 0, stream, (const XI*)x, (XO*)xo, ldx, rows, dim, \
                               (const TO*)y, ldy, (const TO*)y2, mul, add, mod_bstride, rows_per_batch, one, (TO*)out, ldo, sat, sat_tag);   \
    } while (0)
    if (ymode == 0) F5_LN_CASE(0);
    else if (ymode == 1) F5_LN_CASE(1);
    else if (ymode == 2) F5_LN_CASE(2);
    else F5_LN_CASE(3);
#undef F5_LN_CASE
}

// ymode (see layernorm_kernel): 1 = x += y (written back), 2 = normalise x + y without writing x, 3 = x = (x + y) + y2 (written back).
// The residual stream is read from `xin` (fp32, or fp16 when xin_f16) and -- modes 1 and 3 -- written to `xout` (fp32 / fp16 by xout_f16;
// the same buffer as xin or another one of the same leading dimension).  fp16 residual storage exists for the bf16 output type only.
int launch_layernorm_res(int precision_out, const void* xin, int xin_f16, void* xout, int xout_f16, int ldx, int rows, int dim, const void* y, int ldy,
                         const void* y2, int ymode, const float* mul, const float* add, int mod_bstride, int rows_per_batch, int add_one, void* out,
                         int ldo, hipStream_t stream, const PrefetchSet* prefetch, unsigned* sat, int sat_tag) {
    if (rows <= 0) return 0;
    if (dim % 4 != 0 || dim > 2048 || (ldx & 3) || (ldo & 3) || (mod_bstride & 3) || (y && (ldy & 3)))
        return f5_fail(F5_EINVAL, "layernorm: dim=%d unsupported", dim);
    if (!y) ymode = 0;
    if (ymode < 0 || ymode > 3 || (ymode == 3 && !y2)) return f5_fail(F5_EINVAL, "layernorm: bad residual mode %d", ymode);
    if ((xin_f16 || xout_f16) && precision_out != F5_PREC_BF16) return f5_fail(F5_EINVAL, "layernorm: fp16 residual storage needs the bf16 output type");
    if (xin_f16 && !xout_f16 && (ymode == 1 || ymode == 3)) return f5_fail(F5_EINVAL, "layernorm: fp16 -> fp32 residual write-back is not built");
    if (rows_per_batch <= 0) rows_per_batch = rows;
    const float one = add_one ? 1.0f : 0.0f;
#define F5_LN_DIM(TO, XI, XO)                                                                                                                  \
    do {                                                                                                                                       \
        if (dim <= 1024)                                                                                                                       \
            ln_launch<TO, 4, XI, XO>(xin, xout, ldx, rows, dim, y, ldy, y2, ymode, mul, add, mod_bstride, rows_per_batch, one, out, ldo, stream, sat, sat_tag); \
        else                                                                                                                                   \
            ln_launch<TO, 8, XI, XO>(xin, xout, ldx, rows, dim, y, ldy, y2, ymode, mul, add, mod_bstride, rows_per_batch, one, out, ldo, stream, sat, sat_tag); \
    } while (0)
    if (precision_out == F5_PREC_BF16 && xin_f16 && xout_f16 && dim == 1024 && g_ln_wide && !(ldx & 7) && !(ldo & 7) && !(y && (ldy & 7))) {
        dim3 grid(cdiv(rows, 4)), block(256);
        PrefetchSet pfs{{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}};
        if (prefetch) pfs = *prefetch;
#define F5_LN_W(M)                                                                                                                              \
    hipLaunchKernelGGL((layernorm1024_h_kernel<M>), grid, block, 0, stream, (const _Float16*)xin, (_Float16*)xout, ldx, rows, (const bf16_t*)y, ldy, \
                       (const bf16_t*)y2, mul, add, mod_bstride, rows_per_batch, one, (bf16_t*)out, ldo, pfs, sat, sat_tag)
        if (ymode == 0 && mod_bstride == 0 && xin == xout && (g_ln_rows == 2 || g_ln_rows == 4) && rows >= g_ln_rows_min) {
            if (g_ln_rows == 4)
                hipLaunchKernelGGL((layernorm1024_h_rows_kernel<4>), dim3(cdiv(rows, 16)), block, 0, stream, (const _Float16*)xin, ldx, rows, mul, add, one,
                                   (bf16_t*)out, ldo, pfs, sat, sat_tag);
            else
                hipLaunchKernelGGL((layernorm1024_h_rows_kernel<2>), dim3(cdiv(rows, 8)), block, 0, stream, (const _Float16*)xin, ldx, rows, mul, add, one,
                                   (bf16_t*)out, ldo, pfs, sat, sat_tag);
        } else if (ymode == 0) F5_LN_W(0);
        else if (ymode == 1) F5_LN_W(1);
        else if (ymode == 2) F5_LN_W(2);
        else F5_LN_W(3);
#undef F5_LN_W
        F5_LAUNCH_CHECK();
        return 0;
    }
    if (precision_out == F5_PREC_BF16) {
        if (xin_f16)
            F5_LN_DIM(bf16_t, _Float16, _Float16);
        else if (xout_f16)
            F5_LN_DIM(bf16_t, float, _Float16);
        else
            F5_LN_DIM(bf16_t, float, float);
    } else {
        F5_LN_DIM(float, float, float);
    }
#undef F5_LN_DIM
    F5_LAUNCH_CHECK();
    return 0;
}

// fp32 -> fp16 (saturating) copy of n elements (n % 4 == 0): the hoisted part of the input embedding, once per sample() and branch
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t nvec, unsigned* sat) {
    float amax = 0.f, sum = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const f32x4 v = load_res4<float>(src + i * 4);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        sum += (v[0] + v[1]) + (v[2] + v[3]);
        store_res4<_Float16>(dst + i * 4, v);
    }
    res_range_guard(sat, amax, sum);
}
int launch_f32_to_f16(const float* src, void* dst, size_t n, hipStream_t stream, unsigned* sat) {
    if (n == 0) return 0;
    if (n & 3) return f5_fail(F5_EINVAL, "f32_to_f16: n %% 4 != 0");
    const size_t nvec = n >> 2;
    const int grid = (int)(nvec / 256 + 1 < 4096 ? nvec / 256 + 1 : 4096);
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid), dim3(256), 0, stream, src, (_Float16*)dst, nvec, sat);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void f16_to_f32_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, size_t nvec) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) *reinterpret_cast<f32x4*>(dst + i * 4) = load_res4<_Float16>(src + i * 4);
}
int launch_f16_to_f32(const void* src, float* dst, size_t n, hipStream_t stream) {
    if (n == 0) return 0;
    if (n & 3) return f5_fail(F5_EINVAL, "f16_to_f32: n %% 4 != 0");
    const size_t nvec = n >> 2;
    const int grid = (int)(nvec / 256 + 1 < 4096 ? nvec / 256 + 1 : 4096);
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(grid), dim3(256), 0, stream, (const _Float16*)src, dst, nvec);
    F5_LAUNCH_CHECK();
    return 0;
}

int launch_layernorm_add2(int precision_out, float* x, int ldx, int rows, int dim, const void* y, int ldy, const void* y2, int ymode,
                          const float* mul, const float* add, int mod_bstride, int rows_per_batch, int add_one, void* out, int ldo,
                          hipStream_t stream) {
    return launch_layernorm_res(precision_out, x, 0, x, 0, ldx, rows, dim, y, ldy, y2, ymode, mul, add, mod_bstride, rows_per_batch, add_one, out, ldo,
                                stream, nullptr, nullptr, 0);
}

int launch_layernorm_add(int precision_out, float* x, int ldx, int rows, int dim, const void* y, int ldy, const float* mul, const float* add,
                         int mod_bstride, int rows_per_batch, int add_one, void* out, int ldo, hipStream_t stream) {
    return launch_layernorm_add2(precision_out, x, ldx, rows, dim, y, ldy, nullptr, 1, mul, add, mod_bstride, rows_per_batch, add_one, out, ldo,
                                 stream);
}

int launch_layernorm(int precision_out, const float* x, int ldx, int rows, int dim, const float* mul, const float* add, int mod_bstride,
                     int rows_per_batch, int add_one, void* out, int ldo, hipStream_t stream) {
    return launch_layernorm_add(precision_out, const_cast<float*>(x), ldx, rows, dim, nullptr, 0, mul, add, mod_bstride, rows_per_batch, add_one,
                                out, ldo, stream);
}

// ----------------------------------------------------------------------------- UNetT (reference model/backbones/unett.py): RMSNorm and the time-token plumbing
// x_transformers.RMSNorm as unett.py:146,156,175 uses it: out = x / max(||x||_2, 1e-12) * sqrt(dim) * g   (one wavefront per row, fp32 math)
template <typename TO>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int ldx, int rows, int dim, const float* __restrict__ g, TO* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float q = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float nrm = sqrtf(wave_sum(q));
    const float sc = sqrtf((float)dim) / fmaxf(nrm, 1e-12f);
    TO* orow = out + (size_t)row * ldo;
    for (int c = lane * 4; c < dim; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * sc * gv[e];
        if constexpr (sizeof(TO) == 2)
            *reinterpret_cast<bf16x4*>(orow + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        else
            *reinterpret_cast<f32x4*>(orow + c) = f32x4{o[0], o[1], o[2], o[3]};
    }
}
int launch_rmsnorm(int precision_out, const float* x, int ldx, int rows, int dim, const float* g, void* out, int ldo, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (dim % 4 != 0 || (ldx & 3) || (ldo & 3)) return f5_fail(F5_EINVAL, "rmsnorm: dim=%d unsupported", dim);
    dim3 grid(cdiv(rows, 4)), block(256);
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), grid, block, 0, stream, x, ldx, rows, dim, g, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL((rmsnorm_kernel<float>), grid, block, 0, stream, x, ldx, rows, dim, g, (float*)out, ldo);
    F5_LAUNCH_CHECK();
    return 0;
}

// unett.py:211-213: the time embedding is prepended as one more token.  dst[b][0] = temb[b]; dst[b][1 + n] = h[b][n] + branch[b][n]
// (h = input projection, branch = the position-conv branch in the activation dtype: InputEmbedding's x + conv_pos_embed(x), unett.py:96-97)
template <typename TB>
__global__ __launch_bounds__(256) void pack_time_token_kernel(const float* __restrict__ h, const TB* __restrict__ branch, const float* __restrict__ temb,
                                                              int temb_bstride, int B, int N, int D, float* __restrict__ dst) {
    const int row = blockIdx.x;  // b * (N + 1) + s
    const int b = row / (N + 1), sidx = row - b * (N + 1);
    float* d = dst + (size_t)row * D;
    if (sidx == 0) {
        for (int c = threadIdx.x; c < D; c += 256) d[c] = temb[(size_t)b * temb_bstride + c];
    } else {
        const size_t src = (size_t)b * N + (sidx - 1);
        for (int c = threadIdx.x; c < D; c += 256) d[c] = h[src * D + c] + to_f32(branch[src * D + c]);
    }
}
int launch_pack_time_token(int precision, const float* h, const void* branch, const float* temb, int temb_bstride, int B, int N, int D, float* dst,
                           hipStream_t stream) {
    if (B * (N + 1) <= 0) return 0;
    if (precision == F5_PREC_BF16)
        hipLaunchKernelGGL((pack_time_token_kernel<bf16_t>), dim3(B * (N + 1)), dim3(256), 0, stream, h, (const bf16_t*)branch, temb, temb_bstride, B, N, D, dst);
    else
        hipLaunchKernelGGL((pack_time_token_kernel<float>), dim3(B * (N + 1)), dim3(256), 0, stream, h, (const float*)branch, temb, temb_bstride, B, N, D, dst);
    F5_LAUNCH_CHECK();
    return 0;
}
// unett.py:214: mask = F.pad(mask, (1, 0), value=1)
__global__ __launch_bounds__(256) void pad_mask_kernel(const uint8_t* __restrict__ mask, int B, int N, uint8_t* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * (N + 1)) return;
    const int b = i / (N + 1), sidx = i - b * (N + 1);
    dst[i] = sidx == 0 ? (uint8_t)1 : mask[(size_t)b * N + (sidx - 1)];
}
int launch_pad_mask(const uint8_t* mask, int B, int N, uint8_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(pad_mask_kernel, dim3(cdiv(B * (N + 1), 256)), dim3(256), 0, stream, mask, B, N, dst);
    F5_LAUNCH_CHECK();
    return 0;
}
// unett.py:246: [:, 1:, :] -- drop the time token.  src [B * (N + 1), cols] -> dst [B * N, cols]
__global__ __launch_bounds__(256) void drop_time_token_kernel(const float* __restrict__ src, int B, int N, int cols, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N * cols) return;
    const size_t row = i / cols;
    const int c = (int)(i - row * cols);
    const size_t b = row / N, n = row - b * N;
    dst[i] = src[(b * (N + 1) + n + 1) * cols + c];
}
int launch_drop_time_token(const float* src, int B, int N, int cols, float* dst, hipStream_t stream) {
    const size_t total = (size_t)B * N * cols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(drop_time_token_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, B, N, cols, dst);
    F5_LAUNCH_CHECK();
    return 0;
}
// ---- MMDiT (reference model/backbones/mmdit.py, JointAttnProcessor modules.py:509-606)
// `nb` segments of seg_bytes each (multiple of 16, 16-byte aligned) from src + b * src_bstride to dst + b * dst_bstride: the x / text rows of
// every utterance into (or out of) the joint [frames | text] sequence the attention kernel reads
__global__ __launch_bounds__(256) void copy_segments_kernel(const uint4* __restrict__ src, size_t src_bstride, uint4* __restrict__ dst, size_t dst_bstride,
                                                            size_t seg_vec) {
    const uint4* s = src + (size_t)blockIdx.y * src_bstride;
    uint4* d = dst + (size_t)blockIdx.y * dst_bstride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < seg_vec; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
int launch_copy_segments(const void* src, size_t src_bstride_bytes, void* dst, size_t dst_bstride_bytes, size_t seg_bytes, int nb, hipStream_t stream) {
    if (nb <= 0 || seg_bytes == 0) return 0;
    if ((seg_bytes | src_bstride_bytes | dst_bstride_bytes | (size_t)(uintptr_t)src | (size_t)(uintptr_t)dst) & 15)
        return f5_fail(F5_EINVAL, "copy_segments: sizes and addresses must be multiples of 16 bytes");
    const size_t vec = seg_bytes / 16;
    const unsigned gx = (unsigned)std::min<size_t>((vec + 255) / 256, 4096);
    hipLaunchKernelGGL(copy_segments_kernel, dim3(gx, nb), dim3(256), 0, stream, (const uint4*)src, src_bstride_bytes / 16, (uint4*)dst, dst_bstride_bytes / 16, vec);
    F5_LAUNCH_CHECK();
    return 0;
}
// modules.py:573: attn_mask = F.pad(mask, (0, nt), value=True) -- every text key is visible
__global__ __launch_bounds__(256) void joint_mask_kernel(const uint8_t* __restrict__ mask, int B, int N, int nt, uint8_t* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x, S = N + nt;
    if (i >= B * S) return;
    const int b = i / S, sidx = i - b * S;
    dst[i] = sidx < N ? mask[(size_t)b * N + sidx] : (uint8_t)1;
}
int launch_joint_mask(const uint8_t* mask, int B, int N, int nt, uint8_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(joint_mask_kernel, dim3(cdiv(B * (N + nt), 256)), dim3(256), 0, stream, mask, B, N, nt, dst);
    F5_LAUNCH_CHECK();
    return 0;
}
// ---- qk_norm = "rms_norm" (reference model/modules.py:275-294, 391-396, 463-467): RMSNorm(dim_head = 64, eps 1e-6, learned weight) on the q
// and k features of every head, then the rotary embedding on the first rope_heads heads -- in place on the q|k|v rows the projection stored
// (the fused-RoPE GEMM epilogue is not used with this switch).  One wave per (token row, q|k part, head): lane = feature.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(T* __restrict__ qkv, int ldq, int inner, int heads, int rope_heads, const float* __restrict__ wq,
                                                          const float* __restrict__ wk, const float* __restrict__ rope, int rows_per_batch, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int pos = row % rows_per_batch;
    for (int item = wave; item < 2 * heads; item += 4) {
        const int part = item / heads, h = item - part * heads;
        T* px = qkv + (size_t)row * ldq + (size_t)part * inner + h * 64 + lane;
        const float x = to_f32(*px);
        float ss = x * x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        float y = x * rsqrtf(ss * (1.0f / 64.0f) + 1e-6f) * (part == 0 ? wq : wk)[lane];
        const float other = __shfl_xor(y, 1, 64);
        if (h < rope_heads) {  // x_transformers apply_rotary_pos_emb, adjacent pairs (2j, 2j+1)
            const float c = rope[((size_t)pos * 32 + (lane >> 1)) * 2], sn = rope[((size_t)pos * 32 + (lane >> 1)) * 2 + 1];
            y = (lane & 1) ? __builtin_fmaf(y, c, other * sn) : __builtin_fmaf(y, c, -(other * sn));
        }
        *px = from_f32<T>(y);
    }
}
int launch_qknorm_rope(int precision, void* qkv, int ldq, int rows, int inner, int heads, int rope_heads, const float* wq, const float* wk,
                       const float* rope, int rows_per_batch, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (precision == F5_PREC_BF16)
        hipLaunchKernelGGL((qknorm_rope_kernel<bf16_t>), dim3(rows), dim3(256), 0, stream, (bf16_t*)qkv, ldq, inner, heads, rope_heads, wq, wk, rope, rows_per_batch, rows);
    else
        hipLaunchKernelGGL((qknorm_rope_kernel<float>), dim3(rows), dim3(256), 0, stream, (float*)qkv, ldq, inner, heads, rope_heads, wq, wk, rope, rows_per_batch, rows);
    F5_LAUNCH_CHECK();
    return 0;
}
// ---- ragged sampler (f5_sample_ragged): rows flagged 1 are the zero gaps between utterances, which stand for the position conv's zero padding
__global__ __launch_bounds__(256) void zero_rows_kernel(uint4* __restrict__ x, int row_vec, const uint8_t* __restrict__ flags) {
    const int row = blockIdx.x;
    if (!flags[row]) return;
    for (int c = threadIdx.x; c < row_vec; c += 256) x[(size_t)row * row_vec + c] = make_uint4(0u, 0u, 0u, 0u);
}
int launch_zero_rows(void* x, size_t row_bytes, int rows, const uint8_t* flags, hipStream_t stream) {
    if (rows <= 0) return 0;
    if ((row_bytes & 15) || ((size_t)(uintptr_t)x & 15)) return f5_fail(F5_EINVAL, "zero_rows: rows must be multiples of 16 bytes");
    hipLaunchKernelGGL(zero_rows_kernel, dim3(rows), dim3(256), 0, stream, (uint4*)x, (int)(row_bytes / 16), flags);
    F5_LAUNCH_CHECK();
    return 0;
}
// x += y (skip_connect_type "add", unett.py:237-238); n % 4 == 0
__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ x, const float* __restrict__ y, size_t nvec) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 4);
        a += *reinterpret_cast<const f32x4*>(y + i * 4);
        *reinterpret_cast<f32x4*>(x + i * 4) = a;
    }
}
int launch_add_f32(float* x, const float* y, size_t n, hipStream_t stream) {
    if (n == 0) return 0;
    if (n & 3) return f5_fail(F5_EINVAL, "add_f32: n %% 4 != 0");
    const size_t nvec = n >> 2;
    hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)(nvec / 256 + 1 < 4096 ? nvec / 256 + 1 : 4096)), dim3(256), 0, stream, x, y, nvec);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- depthwise conv k=7 + LayerNorm(affine)
template <typename TO, int MAXV>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, int B, int N, int C, const float* __restrict__ wt,
                                                         const float* __restrict__ cbias, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, TO* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * N) return;
    const int pos = row % N;
    const int nvec = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            f32x4 a = *reinterpret_cast<const f32x4*>(cbias + c * 4);
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const int sp = pos + tap - 3;
                if (sp >= 0 && sp < N) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(row + tap - 3) * C + c * 4);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + (size_t)tap * C + c * 4);
                    a += xv * wv;
                }
            }
            v[i] = a;
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-6f);
    TO* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(ln_w + c * 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(ln_b + c * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * w4[e] + b4[e];
            if constexpr (sizeof(TO) == 2) {
                *reinterpret_cast<bf16x4*>(orow + c * 4) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            } else {
                *reinterpret_cast<f32x4*>(orow + c * 4) = f32x4{o[0], o[1], o[2], o[3]};
            }
        }
    }
}

int launch_dwconv7_ln(int precision_out, const float* x, int B, int N, int C, const float* wt, const float* cbias, const float* ln_w,
                      const float* ln_b, void* out, int ldo, hipStream_t stream) {
    if (B * N <= 0) return 0;
    if (C % 4 != 0 || C > 1024 || (ldo & 3)) return f5_fail(F5_EINVAL, "dwconv7_ln: C=%d unsupported", C);
    dim3 grid(cdiv(B * N, 4)), block(256);
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((dwconv7_ln_kernel<bf16_t, 4>), grid, block, 0, stream, x, B, N, C, wt, cbias, ln_w, ln_b, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL((dwconv7_ln_kernel<float, 4>), grid, block, 0, stream, x, B, N, C, wt, cbias, ln_w, ln_b, (float*)out, ldo);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- GRN
// pass 1: sumsq[b][c] over the sequence; pass 2: Nx = G / (mean_c G + 1e-6); pass 3: apply in place.
template <typename T>
__global__ __launch_bounds__(256) void grn_sumsq_kernel(const T* __restrict__ h, int N, int C, float* __restrict__ sumsq) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int n = rg; n < N; n += 4) {
            const float v = to_f32(h[((size_t)b * N + n) * C + c]);
            s += v * v;
        }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < C) sumsq[(size_t)b * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void grn_norm_kernel(float* __restrict__ g, int C, float* __restrict__ meanbuf) {
    // one block per batch element: g[b][c] = sqrt(sumsq); then divide by (mean_c + 1e-6)
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = sqrtf(g[(size_t)b * C + c]);
        g[(size_t)b * C + c] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float denom = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C + 1e-6f;
    if (threadIdx.x == 0) meanbuf[b] = denom;
    for (int c = threadIdx.x; c < C; c += 256) g[(size_t)b * C + c] = g[(size_t)b * C + c] / denom;
}
template <typename T>
__global__ __launch_bounds__(256) void grn_apply_kernel(T* __restrict__ h, int N, int C, const float* __restrict__ nx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int b = (int)(i / ((size_t)N * C));
    const float v = to_f32(h[i]);
    h[i] = from_f32<T>(gamma[c] * (v * nx[(size_t)b * C + c]) + beta[c] + v);
}

int launch_grn(int precision, void* h, int B, int N, int C, const float* gamma, const float* beta, float* scratch, hipStream_t stream) {
    if (B * N <= 0) return 0;
    const size_t total = (size_t)B * N * C;
    if (precision == F5_PREC_BF16)
        hipLaunchKernelGGL((grn_sumsq_kernel<bf16_t>), dim3(cdiv(C, 64), B), dim3(256), 0, stream, (const bf16_t*)h, N, C, scratch);
    else
        hipLaunchKernelGGL((grn_sumsq_kernel<float>), dim3(cdiv(C, 64), B), dim3(256), 0, stream, (const float*)h, N, C, scratch);
    F5_LAUNCH_CHECK();
    hipLaunchKernelGGL(grn_norm_kernel, dim3(B), dim3(256), 0, stream, scratch, C, scratch + (size_t)B * C);
    F5_LAUNCH_CHECK();
    if (precision == F5_PREC_BF16)
        hipLaunchKernelGGL((grn_apply_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (bf16_t*)h, N, C, scratch, gamma, beta, total);
    else
        hipLaunchKernelGGL((grn_apply_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (float*)h, N, C, scratch, gamma, beta, total);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- text embedding gather (dit.py:49-68)
__global__ __launch_bounds__(256) void text_gather_kernel(const int32_t* __restrict__ text, int nt, int B, int N, int td,
                                                          const float* __restrict__ table, const float* __restrict__ pos_table, int pos_rows,
                                                          int drop_text, float* __restrict__ out, uint8_t* __restrict__ filler) {
    const int row = blockIdx.x;  // b * N + p
    const int b = row / N, p = row % N;
    int tok = 0;
    if (p < nt) tok = text[(size_t)b * nt + p] + 1;  // +1: 0 is the filler token; batch padding -1 -> 0
    if (threadIdx.x == 0 && filler) filler[row] = tok == 0;
    if (drop_text) tok = 0;
    const int pp = p < pos_rows ? p : pos_rows - 1;  // get_pos_embed_indices clamps at precompute_max_pos (modules.py:218; 4096 dit.py:41, 1024 mmdit.py:37)
    for (int c = threadIdx.x; c < td; c += 256) {
        float v = table[(size_t)tok * td + c];
        if (pos_table) v += pos_table[(size_t)pp * td + c];
        out[(size_t)row * td + c] = v;
    }
}
int launch_text_gather(const int32_t* text, int nt, int B, int N, int td, const float* table, const float* pos_table, int pos_rows, int drop_text,
                       float* out, uint8_t* filler, hipStream_t stream) {
    if (B * N <= 0) return 0;
    hipLaunchKernelGGL(text_gather_kernel, dim3(B * N), dim3(256), 0, stream, text, nt, B, N, td, table, pos_table, pos_rows, drop_text, out, filler);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, int rows, int cols, const uint8_t* __restrict__ flags) {
    const int row = blockIdx.x;
    if (!flags[row]) return;
    for (int c = threadIdx.x; c < cols; c += 256) x[(size_t)row * cols + c] = 0.f;
}
int launch_mask_rows(float* x, int rows, int cols, const uint8_t* zero_flags, hipStream_t stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(rows), dim3(256), 0, stream, x, rows, cols, zero_flags);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- timestep embedding (modules.py:149-161)
__global__ void time_sinus_kernel(const float* __restrict__ t, int n, float* __restrict__ out) {
    const int i = blockIdx.x, j = threadIdx.x;  // 128 threads
    const float k = logf(10000.0f) / 127.0f;
    const float f = expf((float)j * -k);
    const float arg = 1000.0f * t[i] * f;
    out[(size_t)i * 256 + j] = sinf(arg);
    out[(size_t)i * 256 + 128 + j] = cosf(arg);
}
int launch_time_sinus(const float* t, int n, float* out, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(time_sinus_kernel, dim3(n), dim3(128), 0, stream, t, n, out);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- small-M fp32 linear (time MLP, AdaLN modulation)
// one wavefront per output feature; the weight row stays in registers while the (few) input rows stream from L2.
template <int KV>  // K <= KV * 256
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ in, int ldi, int rows, const float* __restrict__ W,
                                                        const float* __restrict__ b, int N, int K, int pre_silu, int post_silu,
                                                        float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int nvec = K >> 2;
    f32x4 w[KV];
#pragma unroll
    for (int i = 0; i < KV; ++i) {
        const int c = lane + i * 64;
        w[i] = c < nvec ? *reinterpret_cast<const f32x4*>(W + (size_t)n * K + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bias = b ? b[n] : 0.f;
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)r * ldi + c * 4);
                if (pre_silu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = act_silu(x[e]);
                }
                s += (x[0] * w[i][0] + x[1] * w[i][1]) + (x[2] * w[i][2] + x[3] * w[i][3]);
            }
        }
        s = wave_sum(s) + bias;
        if (post_silu) s = act_silu(s);
        if (lane == 0) out[(size_t)r * ldo + n] = s;
    }
}
int launch_gemv_rows(const float* in, int ldi, int rows, const float* W, const float* b, int N, int K, int pre_silu, int post_silu,
                     float* out, int ldo, hipStream_t stream) {
    if (rows <= 0 || N <= 0) return 0;
    if (K % 4 != 0 || K > 2048 || (ldi & 3)) return f5_fail(F5_EINVAL, "gemv_rows: K=%d unsupported", K);
    dim3 grid(cdiv(N, 4)), block(256);
    if (K <= 256)
        hipLaunchKernelGGL((gemv_rows_kernel<1>), grid, block, 0, stream, in, ldi, rows, W, b, N, K, pre_silu, post_silu, out, ldo);
    else if (K <= 1024)
        hipLaunchKernelGGL((gemv_rows_kernel<4>), grid, block, 0, stream, in, ldi, rows, W, b, N, K, pre_silu, post_silu, out, ldo);
    else
        hipLaunchKernelGGL((gemv_rows_kernel<8>), grid, block, 0, stream, in, ldi, rows, W, b, N, K, pre_silu, post_silu, out, ldo);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- conversions / packing
template <typename TO>
__global__ __launch_bounds__(256) void convert_pad_kernel(const float* __restrict__ src, int lds, int rows, int cols, int padcols, TO* __restrict__ dst, int ldo) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * padcols) return;
    const int r = (int)(i / padcols), c = (int)(i % padcols);
    dst[(size_t)r * ldo + c] = from_f32<TO>(c < cols ? src[(size_t)r * lds + c] : 0.f);
}
int launch_convert_pad(int precision_out, const float* src, int lds, int rows, int cols, int padcols, void* dst, int ldo, hipStream_t stream) {
    if (rows <= 0) return 0;
    const size_t total = (size_t)rows * padcols;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((convert_pad_kernel<bf16_t>), grid, block, 0, stream, src, lds, rows, cols, padcols, (bf16_t*)dst, ldo);
    else
        hipLaunchKernelGGL((convert_pad_kernel<float>), grid, block, 0, stream, src, lds, rows, cols, padcols, (float*)dst, ldo);
    F5_LAUNCH_CHECK();
    return 0;
}
template <typename TI>
__global__ __launch_bounds__(256) void convert_back_kernel(const TI* __restrict__ src, int lds, int rows, int cols, float* __restrict__ dst, int ldd) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)r * ldd + c] = to_f32(src[(size_t)r * lds + c]);
}
int launch_convert_back(int precision_in, const void* src, int lds, int rows, int cols, float* dst, int ldd, hipStream_t stream) {
    if (rows <= 0) return 0;
    const size_t total = (size_t)rows * cols;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (precision_in == F5_PREC_BF16)
        hipLaunchKernelGGL((convert_back_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, lds, rows, cols, dst, ldd);
    else
        hipLaunchKernelGGL((convert_back_kernel<float>), grid, block, 0, stream, (const float*)src, lds, rows, cols, dst, ldd);
    F5_LAUNCH_CHECK();
    return 0;
}

// A_base row m = [ cond[m][0..mel) masked by lens | 0 pad to melp | text_embed[m][0..td) ]
template <typename TO>
__global__ __launch_bounds__(256) void pack_base_kernel(const float* __restrict__ cond, const int32_t* __restrict__ lens, const float* __restrict__ te,
                                                        int B, int N, int mel, int melp, int td, int zero_cond, TO* __restrict__ dst, int ldd) {
    const int row = blockIdx.x;
    const int b = row / N, p = row % N;
    const bool keep = !zero_cond && (lens == nullptr || p < lens[b]);  // step_cond = where(cond_mask, cond, 0) (cfm.py:148-150)
    for (int c = threadIdx.x; c < melp + td; c += 256) {
        float v;
        if (c < melp)
            v = (keep && c < mel) ? cond[(size_t)row * mel + c] : 0.f;
        else
            v = te[(size_t)row * td + (c - melp)];
        dst[(size_t)row * ldd + c] = from_f32<TO>(v);
    }
}
int launch_pack_base(int precision_out, const float* cond, const int32_t* lens, const float* text_embed, int B, int N, int mel, int melp,
                     int td, int zero_cond, void* dst, int ldd, hipStream_t stream) {
    if (B * N <= 0) return 0;
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((pack_base_kernel<bf16_t>), dim3(B * N), dim3(256), 0, stream, cond, lens, text_embed, B, N, mel, melp, td, zero_cond, (bf16_t*)dst, ldd);
    else
        hipLaunchKernelGGL((pack_base_kernel<float>), dim3(B * N), dim3(256), 0, stream, cond, lens, text_embed, B, N, mel, melp, td, zero_cond, (float*)dst, ldd);
    F5_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- CFG combine + fixed-grid ODE step
__global__ __launch_bounds__(256) void cfg_step_kernel(const float* __restrict__ x_base, const float* __restrict__ vc, const float* __restrict__ vu, int ldv,
                                                       int rows, int mel, float cfg, const float* __restrict__ coef, float* __restrict__ x_out,
                                                       float* __restrict__ x_out2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * mel) return;
    const int r = (int)(i / mel), c = (int)(i % mel);
    const float pc = vc[(size_t)r * ldv + c];
    float f = pc;
    if (vu) f = pc + (pc - vu[(size_t)r * ldv + c]) * cfg;  // pred + (pred - null_pred) * cfg_strength (cfm.py:173)
    const float y = x_base[i] + coef[0] * f;
    x_out[i] = y;
    if (x_out2) x_out2[i] = y;
}
int launch_cfg_step(const float* x_base, const float* vc, const float* vu, int ldv, int rows, int mel, float cfg, const float* coef,
                    float* x_out, float* x_out2, hipStream_t stream) {
    if (rows <= 0) return 0;
    const size_t total = (size_t)rows * mel;
    hipLaunchKernelGGL(cfg_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x_base, vc, vu, ldv, rows, mel, cfg, coef, x_out, x_out2);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void final_where_kernel(const float* __restrict__ cond, const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                          int B, int N, int mel, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N * mel) return;
    const int row = (int)(i / mel);
    const int b = row / N, p = row % N;
    out[i] = p < lens[b] ? cond[i] : x[i];
}
int launch_final_where(const float* cond, const float* x, const int32_t* lens, int B, int N, int mel, float* out, hipStream_t stream) {
    const size_t total = (size_t)B * N * mel;
    if (total == 0) return 0;
    hipLaunchKernelGGL(final_where_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, cond, x, lens, B, N, mel, out);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void len_mask_kernel(const int32_t* __restrict__ durations, int B, int N, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    mask[i] = (i % N) < durations[i / N];
}
int launch_len_mask(const int32_t* durations, int B, int N, uint8_t* mask, hipStream_t stream) {
    if (B * N <= 0) return 0;
    hipLaunchKernelGGL(len_mask_kernel, dim3(cdiv(B * N, 256)), dim3(256), 0, stream, durations, B, N, mask);
    F5_LAUNCH_CHECK();
    return 0;
}

// rowbits[(m / 128) * 16 + (m % 16)] bit ((m % 128) / 16) = mask[m] != 0: the eight rows one lane of the tuned GEMM's epilogue owns
// inside a 128-row wave tile, in one byte (gemm.h: GemmParams::rowbits)
__global__ __launch_bounds__(256) void rowbits_kernel(const uint8_t* __restrict__ mask, int rows, uint8_t* __restrict__ bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ((rows + 127) >> 7) * 16) return;
    const int base = (i >> 4) * 128 + (i & 15);
    unsigned b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = base + 16 * j;
        if (m < rows && mask[m]) b |= 1u << j;
    }
    bits[i] = (uint8_t)b;
}
int launch_rowbits(const uint8_t* mask, int rows, uint8_t* bits, hipStream_t stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowbits_kernel, dim3(cdiv(cdiv(rows, 128) * 16, 256)), dim3(256), 0, stream, mask, rows, bits);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ dst, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v;
}
int launch_fill_f32(float* dst, size_t n, float v, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, n, v);
    F5_LAUNCH_CHECK();
    return 0;
}

// small host -> device float transfers as KERNEL ARGUMENTS (copied by the runtime at launch: no host-buffer lifetime or
// pageable-memcpy staging question, and capturable)
struct FloatPack {
    float v[64];
};
__global__ void set_floats_kernel(float* __restrict__ dst, FloatPack pk, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = pk.v[threadIdx.x];
}
int launch_set_floats(float* dst, const float* host_vals, int n, hipStream_t stream) {
    for (int off = 0; off < n; off += 64) {
        FloatPack pk;
        const int m = n - off < 64 ? n - off : 64;
        for (int i = 0; i < 64; ++i) pk.v[i] = i < m ? host_vals[off + i] : 0.f;
        hipLaunchKernelGGL(set_floats_kernel, dim3(1), dim3(64), 0, stream, dst + off, pk, m);
        F5_LAUNCH_CHECK();
    }
    return 0;
}
