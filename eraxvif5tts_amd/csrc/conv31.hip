// conv31.hip -- ConvPositionEmbedding's grouped Conv1d(dim, dim, k = 31, groups = 16, padding = 15) + Mish for dim = 1024
// (reference model/modules.py:167-190: two of them per network evaluation), bf16 in / bf16 out, fp32 accumulate.
//
// The generic path (gemm_fast.hip, GEMM_CONV31) treats the conv as an implicit GEMM and re-fetches the 256-token activation slice
// for every one of the 31 taps: 20 KiB of LDS-DMA per MFLOP-sized K-step, 2.5x the traffic of the dense tiles (570 TFLOP/s).
// With 64 channels per group a token row of one group is exactly one 128-byte line, so this kernel keeps the whole halo tile
// resident instead:
//   * workgroup = 256 tokens of ONE utterance x the 64 output channels of one group; 4 waves, each 64 tokens x 64 channels;
//   * the (256 + 30)-row x 128-byte halo tile (rows outside the utterance read a zero page = Conv1d's zero padding) is fetched
//     ONCE by LDS-DMA in whole-line pieces (8 rows x 128 B); a tap only shifts the fragment row index;
//   * per tap only the 64 x 64 weight slice (8 KiB, identical for every token tile of the group -> L2-resident) is streamed
//     through a 2-slot LDS buffer, one barrier per tap;
//   * 4 weight slots (3 taps fetched ahead): 68 KiB of LDS per workgroup -> 2 workgroups per CU cover each other's barriers;
//   * rows are 128 B in LDS for both operands: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7), conflict-free for
//     ds_read_b128 at any tap shift.
// Weights are the tap-major image the model already builds for the generic path: [31][dim][64] (GemmParams::conv_win == 64).
#include "gemm_tile.h"
#include "runtime.h"

__device__ __attribute__((aligned(256))) unsigned char g_conv_zero_page[256];  // zero-initialised: Conv1d zero padding / rows outside the utterance

namespace {

constexpr int TAPS = 31, HALO = 15, W_BYTES = 64 * 128;
constexpr int WSLOTS = 4, WAHEAD = WSLOTS - 1;  // weight slices in LDS / taps fetched ahead (4 slots: 68 KiB -> 2 workgroups per CU)

// TOK = 256 tokens per workgroup (64 per wave); TOK = 128 (32 per wave) for launches that would otherwise leave CUs without a workgroup
// (single utterance: 4 token tiles x 16 groups x 2 branches = 128 workgroups of 256 tokens).  Same sums per output either way.
template <int TOK>
__global__ __launch_bounds__(256, 2) void conv31_kernel(GemmParams p, int tiles_per_seq) {
    constexpr int ROWS = TOK + 2 * HALO;            // 286 / 158 halo rows
    constexpr int A_PIECES = (ROWS + 7) / 8;        // 36 / 20 DMA pieces of 8 rows x 128 B
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int TW = TOK / 4, TJ = TW / 16;       // tokens per wave, 16-token tiles per wave
    static_assert(A_PIECES % 4 == 0, "pieces divide over the 4 waves");
    __shared__ __attribute__((aligned(16))) char smem[A_BYTES + WSLOTS * W_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, grp = blockIdx.y, b = blockIdx.z;
    const int L = p.rows_per_batch, n0 = tile * TOK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (size_t)b * L * p.lda + grp * 64;
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W) + (size_t)grp * 64 * 64;  // + tap * N * 64

    // ---- DMA: a piece is 8 rows x 128 B; lane -> (row = lane >> 3, physical chunk = lane & 7)
    const int drow = lane >> 3, dchunk = lane & 7;
    auto dma_weights = [&](int tap, int buf) {  // 8 pieces, 2 per wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pc = wave * 2 + j, row = pc * 8 + drow;  // output channel inside the group
            const int lc = dchunk ^ ((row >> 1) & 7);
            dma16(W + (size_t)tap * p.N * 64 + (size_t)row * 64 + lc * 8, smem + A_BYTES + buf * W_BYTES + pc * 1024);
        }
    };
#pragma unroll
    for (int j = 0; j < A_PIECES / 4; ++j) {  // 36 pieces, 9 per wave
        const int pc = wave * (A_PIECES / 4) + j, row = pc * 8 + drow;  // halo row: token n0 - 15 + row
        const int lc = dchunk ^ ((row >> 1) & 7);
        const int tok = n0 - HALO + row;
        const void* src = (row < ROWS && tok >= 0 && tok < L) ? (const void*)(A + (size_t)tok * p.lda + lc * 8) : (const void*)g_conv_zero_page;
        dma16(src, smem + pc * 1024);
    }
#pragma unroll
    for (int t = 0; t < WAHEAD; ++t) dma_weights(t, t);

    // ---- fragment addressing.  Token fragment (MFMA B operand): halo row = 64 * wave + 16 * j + fr + tap, channels 32 * ks + 8 * fq ..;
    //      weight fragment (A operand): row = 16 * i + fr (output channel), same channel chunk.
    const int fr = lane & 15, fq = lane >> 4;
    const int wrow_sw = (fr >> 1) & 7;  // weight rows 16 * i + fr: (row >> 1) & 7 does not depend on i

    f32x4 acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tap = 0; tap < TAPS; ++tap) {
        // my pieces of the halo tile (tap 0) and of this tap's weights have landed: only the slices fetched behind them (2 pieces
        // per wave per tap, vmcnt retires in order) may still be in flight
        const int behind = min(WAHEAD - 1, TAPS - 1 - tap);
        if (behind >= 2)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (behind == 1)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... everybody's; and everybody is done with tap - 1, whose weight slot is refilled now
        if (tap + WAHEAD < TAPS) dma_weights(tap + WAHEAD, (tap + WAHEAD) % WSLOTS);
        const char* wb = smem + A_BYTES + (tap % WSLOTS) * W_BYTES + fr * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], af[TJ];
            const int wc = ((ks * 4 + fq) ^ wrow_sw) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + i * 2048 + wc);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int row = wave * TW + j * 16 + fr + tap;
                af[j] = *reinterpret_cast<const bf16x8*>((const char*)smem + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: out = mish(acc + bias) as bf16, pairs of feature tiles exchanged so that a lane stores 16 bytes
    f32x4 bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = *reinterpret_cast<const f32x4*>(p.bias + grp * 64 + i * 16 + 4 * fq);
    const int ncol = grp * 64 + 16 * (fq & 1) + 8 * (fq >> 1);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out_t) + (size_t)b * L * p.ldo + ncol;
    const int act = p.act;
    static_for<TJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int tok = n0 + wave * TW + j * 16 + fr;
        static_for<2>([&](auto hc) {
            constexpr int i0 = decltype(hc)::value * 2;
            f32x4 v0 = acc[i0][j] + bias4[i0], v1 = acc[i0 + 1][j] + bias4[i0 + 1];
            if (act == ACT_MISH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = fast_mish(v0[e]);
                    v1[e] = fast_mish(v1[e]);
                }
            }
            const u32x4 q = pair_swap(to_bf16x4(v0), to_bf16x4(v1));
            if (tok < L) *reinterpret_cast<u32x4*>(out + (size_t)tok * p.ldo + 32 * (i0 / 2)) = q;
        });
    });
}

}  // namespace

int g_conv31 = 1;  // tuning knob ("conv31"): 1 = dedicated halo-tile kernel for the dim-1024 grouped conv, 0 = implicit GEMM (gemm_fast.hip)

bool conv31_supported(const GemmParams& p, int precision, int epi) {
    if (!g_conv31 || precision != F5_PREC_BF16) return false;
    if (p.conv_cg != 64 || p.conv_win != 64 || p.N % 64 != 0 || p.N / 64 > 65535) return false;
    if (p.rows_per_batch <= 0 || p.M % p.rows_per_batch != 0 || (p.lda & 7) || (p.ldo & 7) || !p.bias || !p.out_t) return false;
    if (!(p.act == ACT_MISH || p.act == ACT_NONE)) return false;
    if (epi == EPI_GATE_T) return !p.gate && !p.rowmask;  // the store-only second conv of the input embedding
    return epi == EPI_STORE_T;
}

int g_conv31_tok = 0;  // tuning knob ("conv31_tok"): tokens per workgroup, 0 = by grid size, 128 or 256 forced
int launch_conv31(const GemmParams& p, hipStream_t stream) {
    const int L = p.rows_per_batch, nb = p.M / L;
    if (nb > 65535) return f5_fail(F5_EINVAL, "conv31: batch %d too large for one launch", nb);
    const bool small = g_conv31_tok == 128 || (g_conv31_tok == 0 && (long)cdiv(L, 256) * (p.N / 64) * nb < 2L * f5_cu_count());
    if (small) {
        const int tiles = cdiv(L, 128);
        hipLaunchKernelGGL(conv31_kernel<128>, dim3(tiles, p.N / 64, nb), dim3(256), 0, stream, p, tiles);
        F5_LAUNCH_CHECK();
        return 0;
    }
    const int tiles = cdiv(L, 256);
    hipLaunchKernelGGL(conv31_kernel<256>, dim3(tiles, p.N / 64, nb), dim3(256), 0, stream, p, tiles);
    F5_LAUNCH_CHECK();
    return 0;
}
