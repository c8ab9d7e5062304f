// gemm_w4.hip -- the block linears of a large batch on ONE wave per SIMD (round 4, late):  out[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate.
//
// Why a second tuned kernel.  Round 4 timed the vendor library at the four block call sites of C2 for the first time (tools/hipblaslt_ref.py): its
// hand-written 256 x 256 x 64 kernel with FOUR waves per workgroup was 17 - 22 % faster than gemm_fast.hip's 8-wave staggered kernel in isolation.
// The difference is the per-wave tile: 128 x 128 outputs per wave need 16 ds_read_b128 per 64 MFMAs, the 128 x 64 tile of the 8-wave kernel 12 per
// 32 -- a third less LDS traffic per FLOP -- and whole 128-byte lines per LDS-DMA row instead of half lines.  Round 1 had tried this tile shape with
// compiler-scheduled code and lost (every stall of a lone wave is exposed); here every non-MFMA instruction of the main loop is PLACED BY HAND
// between the MFMAs (inline asm, hand-counted s_waitcnt), which is what makes the shape pay: tools/gemm_w4_probe.hip developed the schedule
// (8192^3: 1 490 - 1 550 TFLOP/s against 1 320 for the vendor kernel on the same box; LDS bank conflicts 0).
//
// Structure:
//   * persistent grid, one 256-thread workgroup per CU, 256 x 256 output tile, wave w owns tokens (w >> 1) * 128.., features (w & 1) * 128..;
//     256 accumulator registers per lane in the AccVGPR file, two fragment sets (2 x 64 VGPRs) so that the reads of a 32-deep sub-step fly
//     under the MFMAs of the one before;
//   * operands by LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 bytes per instruction) into TWO 64-KiB stage buffers (64 k each) that are
//     refilled IN PLACE two iterations ahead: barrier B1 (every wave has read the buffer) -> 16 requests per wave spread over the MFMAs;
//     barrier B2 behind a counted vmcnt (the other buffer has landed) -> the next iteration's first fragments.  The DMA front crosses tile
//     boundaries, so a tile's first two stages land before / under the previous tile's epilogue;
//   * LDS image: row r (128 bytes), logical 16-byte chunk c at physical chunk c ^ ((r >> 1) & 7) -- applied on the DMA source address and on the
//     read address -- every 16-lane group of a ds_read_b128 covers the 16 slots of the 256-byte bank row once (SQ_LDS_BANK_CONFLICT = 0);
//   * same MFMA (weights on the row index), same K order, same start value (bias, or 0 under the LayerNorm fold) and the same epilogue arithmetic
//     as gemm_fast.hip's persistent build: a wave's 128 features are treated as two 64-feature halves = two "waves" of that kernel, so the output
//     bits do not depend on which of the two kernels a launch takes (tests/test_gpu_ops.py::test_w4_kernel_equals_the_8_wave_kernel).
//   * a second tile height (template parameter MI = 4: 128 x 256, 64 x 128 per wave) for small batches; there the folded projections also finish the
//     LayerNorm fold's row statistics themselves (finish_stats) and the kernel touches the weights of the launches behind it.
// Launch conditions (gemm_w4_ok): bf16 / fp16-fold operands, M % 128 == 0, N % 256 == 0, K % 128 == 0, K >= 256, the lean operand forms of the
// persistent 8-wave build, and enough tiles for the CUs (w4_tile_rows: 256-row tiles from three quarters of the CUs on, else 128-row tiles from half).
#include "gemm_tile.h"
#include "lnf_stats_math.h"
#include <string.h>

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

namespace {
constexpr int OPB = 32768;          // the weight operand of one stage: 256 rows x 128 bytes
constexpr int EPI_LDS_BYTES = 16384;  // behind the stage buffers: the LayerNorm fold's epilogue operands (4 waves x 2 KiB; 4 x 11 KiB on the 128-row tile, which can
                                      // finish the row statistics itself), or bias[N] | gate[N] of an in-place residual launch

// MI: token tiles (of 16 rows) per wave.  8: the 256 x 256 tile (128 x 128 per wave, 128 MFMAs per iteration).  4: a 128 x 256 tile (64 x 128 per wave, 128
// accumulator registers, 64 MFMAs per iteration) for launches that have no 256-row tile for every CU but a 128-row one (batches of 2 - 4 utterances).
template <int EPI, bool LNF, int MI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(GemmParams p, int tiles_n, int nblocks) {
    static_assert(EPI == EPI_STORE_T || EPI == EPI_ROPE_T || EPI == EPI_RESID, "block linears only");
    static_assert(!LNF || EPI != EPI_RESID, "LayerNorm fold: QKV (+ RoPE) and FF1 (+ GELU)");
    static_assert(MI == 8 || MI == 4, "token tiles per wave");
    constexpr int WROWS = MI * 16, BM = 2 * WROWS;  // token rows per wave / per tile
    constexpr int OPA = BM * 128;                   // the activation operand of one stage
    constexpr int WBASE = 2 * OPA, EPI_LDS = 2 * OPA + 2 * OPB;  // LDS: [A buffer 0][A buffer 1][W buffer 0][W buffer 1][epilogue operands]
    constexpr int NP = MI + 8, NR = MI + 8, NMF = 16 * MI;       // requests per wave, fragment reads per sub-step, MFMAs per iteration
    constexpr int LNF_AREA = MI == 4 ? 11264 : 2048;  // per wave: c1[128] | c2[128] | (mean, rstd)[128 rows] (| 16 partial planes x 64 rows | pivots of 64 rows)
    constexpr int EPI_BYTES = 4 * LNF_AREA > EPI_LDS_BYTES ? 4 * LNF_AREA : EPI_LDS_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[EPI_LDS + EPI_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    auto tile_mn = [&](int bid, int& tm0, int& tn0) {  // XCD band + L2 patch order, as gemm_fast.hip
        const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
        int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (p.tile_reverse) swz = nblocks - 1 - swz;
        const int gm = p.tile_group > 0 ? p.tile_group : 1;
        const int tiles_m_all = nblocks / tiles_n;
        const int grp = swz / (gm * tiles_n);
        const int gsz = min(gm, tiles_m_all - grp * gm);
        const int rin = swz - grp * gm * tiles_n;
        const int tile_n = rin / gsz, tile_m = grp * gm + (rin - tile_n * gsz);
        tm0 = tile_m * BM;
        tn0 = tile_n * 256;
    };
    // weight prefetch for the launches BEHIND this one (GemmParams::pf_p; small batches: every block's weights come from HBM again, and with the
    // statistics finished inside the consumer there is no statistics launch left to do it): thread t of workgroup b touches lines (2 b + r) 256 + t,
    // one dword each.  The loads are the oldest entries of the vector-memory queue and their values are looked at behind the last epilogue only.
    [[maybe_unused]] unsigned pfv[4] = {0u, 0u, 0u, 0u};
    if constexpr (MI == 4) {
        if (p.pf_n[0] | p.pf_n[1]) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const size_t line = ((size_t)blockIdx.x * 2 + r) * 256u + tid;
                if (line * 128u < p.pf_n[0]) pfv[r] = *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.pf_p[0]) + line * 128u);
                if (line * 128u < p.pf_n[1]) pfv[2 + r] = *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.pf_p[1]) + line * 128u);
            }
        }
    }
    const int G = gridDim.x;
    const int my_tiles = (nblocks - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / 64;

    // ---- DMA: per iteration this wave moves pieces wave * MI + jj of the activation tile and wave * 8 + jj of the weight tile; a piece is 8 rows x 128
    // bytes (lane l -> row l >> 3, physical chunk l & 7).  Per-lane byte offsets are constant, the tile and the K position sit in two scalar bases.
    unsigned voffA[MI], voffW[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int row = (wave * 8 + jj) * 8 + (lane >> 3);
        const int logical = (lane & 7) ^ ((row >> 1) & 7);
        voffW[jj] = (unsigned)(row * p.ldw * 2 + logical * 16);
    }
#pragma unroll
    for (int jj = 0; jj < MI; ++jj) {
        const int row = (wave * MI + jj) * 8 + (lane >> 3);
        const int logical = (lane & 7) ^ ((row >> 1) & 7);
        voffA[jj] = (unsigned)(row * p.lda * 2 + logical * 16);
    }
    const char* baseA;
    const char* baseW;
    int f_tile = blockIdx.x, f_k = 0;
    auto front_tile = [&]() {
        int sm, sn;
        tile_mn(f_tile, sm, sn);
        baseA = static_cast<const char*>(p.A) + (size_t)sm * p.lda * 2;
        baseW = static_cast<const char*>(p.W) + (size_t)sn * p.ldw * 2;
    };
    front_tile();
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int wdstA = (int)lds0 + wave * MI * 1024, wdstW = (int)lds0 + WBASE + wave * 8192;  // this wave's pieces inside an operand buffer
    // (s_nop: one wait state between the scalar write of M0 and the LDS-DMA that reads it; the hazard recogniser does not look inside an asm)
    auto dma16w = [&](int ldsdst, unsigned voff, const char* base) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsdst), "v"(voff), "s"(base) : "memory");
    };
    auto dma4w = [&](int ldsdst, unsigned voff, const void* base) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(ldsdst), "v"(voff), "s"(base) : "memory");
    };
    auto front_advance = [&]() {
        baseA += 128;
        baseW += 128;
        if (++f_k == nk) {
            f_k = 0;
            f_tile += G;
            if (f_tile < nblocks) {
                front_tile();
            } else {  // past the last tile: walk its stages again (valid addresses; the requests are never read)
                f_tile -= G;
                baseA -= (size_t)nk * 128;
                baseW -= (size_t)nk * 128;
            }
        }
    };
    // ---- fragment read addresses (LDS byte addresses; the 16-row tile index and the buffer are immediate offsets).  Layout: [A buffer 0][A buffer 1]
    // [W buffer 0][W buffer 1]
    const unsigned ra0 = lds0 + (unsigned)((wm * WROWS + fr) * 128 + ((fq ^ (fr >> 1)) * 16));  // sub-step 0: chunks 0..3
    const unsigned ra1 = ra0 ^ 64u;                                                               // sub-step 1: chunks 4..7
    const unsigned rw0 = lds0 + WBASE + (unsigned)((wn * 128 + fr) * 128 + ((fq ^ (fr >> 1)) * 16));
    const unsigned rw1 = rw0 ^ 64u;

    f32x4 acc[8][MI];  // [feature tile][token tile], AccVGPRs
    f32x4 fw[2][8], fa[2][MI];
#define W4_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto rd = [&](auto sc, auto xc, auto ec) {  // fragment e (0..7 weight tiles, 8..8 + MI - 1 token tiles) of sub-step S from buffer X
        constexpr int S = decltype(sc)::value, X = decltype(xc)::value, e = decltype(ec)::value;
        (void)fw; (void)fa; (void)ra0; (void)ra1; (void)rw0; (void)rw1;  // (asm operands alone do not capture inside a generic lambda)
        if constexpr (e < 8) {
            if constexpr (S == 0) W4_DSR(fw[0][e], rw0, X * OPB + e * 2048); else W4_DSR(fw[1][e], rw1, X * OPB + e * 2048);
        } else {
            if constexpr (S == 0) W4_DSR(fa[0][e - 8], ra0, X * OPA + (e - 8) * 2048); else W4_DSR(fa[1][e - 8], ra1, X * OPA + (e - 8) * 2048);
        }
    };
    // One 64-deep iteration on buffer X: 128 MFMAs, and between them -- RD1: the 16 fragment reads of sub-step 1 (slots 0 .. 15); B1 behind slot
    // 20: lgkmcnt(0) + barrier = every wave is done with buffer X; the 16 DMA requests of the stage two iterations ahead into buffer X, spread wide
    // (slots 22, 28, .. 112: a request every 6 MFMAs; every 3 from slot 42 measured 2 - 4 % slower at the four call sites, every 2 slower still:
    // tools/gemm_w4_probe.hip schedules 1 / 41 / 42 - 47, gpurun_out/r4aa_w4.log, r4ab_w4.log); B2 behind slot 86: counted vmcnt (this iteration's 11
    // requests so far may fly, everything older has landed) + barrier = the other buffer is complete; RD0: the next iteration's sub-step-0 fragments
    // from it (slots 88, 90, .. 118); lgkmcnt(0) behind the last MFMA.
    // The main loop exists in few copies with straight-line control flow between them (a conditional INSIDE the iteration loop made the register
    // allocator shuffle and spill the 256 tied AccVGPR operands).  So nothing about an iteration is conditional: the last two iterations of the last tile
    // request the last tile's stages again (never read; the final vmcnt(0) in front of the epilogue retires them before the wave ends), and a
    // tile's last iteration reads the next tile's first fragments although the epilogue reads them AGAIN behind itself -- which is what frees
    // their 64 registers inside the epilogue: the values read here are dead there.  WAITV false: a tile's first iteration after an epilogue (everything
    // requested before the epilogue was waited for in front of it; a wait here would also wait for the epilogue's stores: they share vmcnt).
    auto body = [&](auto xc, auto waitc) {
        constexpr int X = decltype(xc)::value;
        constexpr bool WAITV = decltype(waitc)::value;
        // (MI = 4, 64 MFMAs: 12 reads in slots 0..11, B1 behind 13, 12 requests at 14, 18, .. 58, B2 behind 42, the next reads at 44..55; requests every 3
        //  MFMAs, or everything two slots later: no difference at 1 / 2 / 4 utterances, profiles/r4_w4_128_row_schedule_variants.txt)
        constexpr int R1S = 1, B1P = MI == 8 ? 20 : 13, D0 = MI == 8 ? 22 : 14, DS = MI == 8 ? 6 : 4, B2P = MI == 8 ? 86 : 42, R0 = MI == 8 ? 88 : 44,
                      R0S = MI == 8 ? 2 : 1;
        static_for<NMF>([&](auto nc) {
            (void)acc; (void)fw; (void)fa; (void)voffA; (void)voffW; (void)baseA; (void)baseW;
            constexpr int n = decltype(nc)::value;
            constexpr int s = n / (8 * MI), i = (n % (8 * MI)) / MI, j = n % MI;
            if constexpr (LNF)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[s][i]), "v"(fa[s][j]));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[s][i]), "v"(fa[s][j]));
            if constexpr (n < NR * R1S && n % R1S == 0) rd(std::integral_constant<int, 1>{}, xc, std::integral_constant<int, n / R1S>{});
            if constexpr (n == B1P) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (n >= D0 && n < D0 + NP * DS && (n - D0) % DS == 0) {
                constexpr int pc = (n - D0) / DS;
                if constexpr (pc < MI)
                    dma16w(X * OPA + wdstA + pc * 1024, voffA[pc], baseA);
                else
                    dma16w(X * OPB + wdstW + (pc - MI) * 1024, voffW[pc - MI], baseW);
            }
            if constexpr (n == B2P) {
                constexpr int issued = (B2P - D0) / DS + 1 > NP ? NP : (B2P - D0) / DS + 1;  // this iteration's requests so far
                if constexpr (WAITV)
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(issued) : "memory");
                else
                    asm volatile("s_barrier" ::: "memory");
            }
            if constexpr (n >= R0 && n < R0 + NR * R0S && (n - R0) % R0S == 0)
                rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 1 - X>{}, std::integral_constant<int, (n - R0) / R0S>{});
            if constexpr (n == NMF - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        front_advance();
    };

    using T = std::true_type;
    using F = std::false_type;
    using X0 = std::integral_constant<int, 0>;
    using X1 = std::integral_constant<int, 1>;
    // ---- per-tile epilogue operands
    int m0, n0;
    tile_mn(blockIdx.x, m0, n0);
    char* const lnf_lds = smem + EPI_LDS + wave * LNF_AREA;  // LNF: c1[128] | c2[128] | (mean, rstd)[128 rows] of this wave, by LDS-DMA at the start of a tile
    [[maybe_unused]] auto stage_lnf = [&]() {
        const int ldst = (int)lds0 + EPI_LDS + wave * LNF_AREA;
        const unsigned nb = (unsigned)(n0 + wn * 128 + lane) * 4u;
        dma4w(ldst, nb, p.lnf_c1);
        dma4w(ldst + 256, nb + 256u, p.lnf_c1);
        dma4w(ldst + 512, nb, p.lnf_c2);
        dma4w(ldst + 768, nb + 256u, p.lnf_c2);
        if constexpr (MI == 4) {
            if (p.lnf_partial) {
                // statistics finished by THIS kernel (gemm.h: lnf_partial; single-utterance launches, where the two statistics launches of a block
                // were 13 of its 102 us): the producer's 16 partial planes of the wave's 64 rows and the rows' pivots, 8 bytes a row -- lanes
                // 0..31 fetch rows 2l, 2l + 1 (the planes are not padded: the upper half-wave stays out)
                if (lane < 32) {
                    const unsigned ro = (unsigned)((m0 + wm * WROWS + 2 * lane) * 8);
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        dma16w(ldst + 2048 + c * 512, ro, reinterpret_cast<const char*>(p.lnf_partial + (size_t)c * p.lnf_partial_ld * 2));
                    if (p.lnf_pivot) dma16w(ldst + 2048 + 8192, ro, reinterpret_cast<const char*>(p.lnf_pivot));
                }
                return;
            }
        }
        // lane l: rows 2l, 2l + 1 of the wave's (MI = 4: 64 rows past the wave's own ride along; the statistics buffer is padded by 256 rows)
        dma16w(ldst + 1024, (unsigned)((m0 + wm * WROWS + 2 * lane) * 8), reinterpret_cast<const char*>(p.lnf_stats));
    };
    // ... and turned into (mean, rstd) in front of the epilogue: lane l owns row l of the wave's 64, adds the planes in stats_finalize_kernel's order
    // (lnf_stats_math.h: same bits), feeds the staging area the epilogue reads; feature tile 0 also stores them for the next producer (its pivots)
    // and carries the fp16 range guard
    // (The 256-row tile has no LDS for the planes.  Tried: lane l loads rows 2l, 2l + 1 of all 16 planes into registers behind the main loop, for
    //  launches with at most two tiles per CU -- 4 x 1024: 27 980 -> 27 880 mel-frames/s, 3 x 1024: 23 080 -> 22 450: the exposed memory latency per
    //  tile costs what the two statistics launches cost.  profiles/r4_w4_inkernel_statistics_256_row_tiles_tried.txt.  Removed.)
    [[maybe_unused]] auto finish_stats = [&]() {
        if constexpr (MI == 4 && LNF) {
            if (p.lnf_partial) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(lnf_lds + 2048 + c * 512 + lane * 8);
                    s1 += v[0];
                    s2 += v[1];
                }
                const float pv = p.lnf_pivot ? *reinterpret_cast<const float*>(lnf_lds + 2048 + 8192 + lane * 8) : 0.0f;
                float mean, rstd, sumsq;
                lnf_row_stats(s1, s2, pv, p.K, mean, rstd, sumsq);
                *reinterpret_cast<f32x2*>(lnf_lds + 1024 + lane * 8) = f32x2{mean, rstd};
                if (p.lnf_stats_out && n0 == 0 && wn == 0) {
                    const int m = m0 + wm * WROWS + lane;
                    *reinterpret_cast<f32x2*>(p.lnf_stats_out + (size_t)m * 2) = f32x2{mean, rstd};
                    lnf_raise_guard(p.lnf_sat, !(sumsq < 65504.0f * 65504.0f), sumsq, p.lnf_sat_tag, m + p.row0);
                }
            }
        }
    };
    auto load_bias = [&](int tn0, f32x4 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.bias + tn0 + wn * 128 + i * 16 + 4 * fq);
    };
    auto acc_from = [&](const f32x4 (&b4)[8]) {  // every accumulator starts from its feature's bias (gemm_fast.hip: init_acc)
        static_for<8>([&](auto ic) {
            static_for<MI>([&](auto jc) { acc[decltype(ic)::value][decltype(jc)::value] = b4[decltype(ic)::value]; });
        });
    };
    // An accumulator reaches the vector ALU through four volatile v_accvgpr_read_b32 statements AT ITS USE.  Left to the compiler, the AGPR -> VGPR
    // copies of all 256 sit directly behind the last MFMA (where the asm operand is defined) and spill; pinned by an empty asm("" : "+a"(acc)) in
    // front of each use they stay put, but every tuple is first moved to one scratch AGPR tuple (4 v_accvgpr_mov_b32 per accumulator, 256 extra
    // vector instructions per tile and wave: the first builds of this file).
#define W4_ACC(i, j)                                                                         \
    ([&]() -> f32x4 {                                                                        \
        float c0_, c1_, c2_, c3_;                                                            \
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(c0_) : "a"(acc[i][j][0]));           \
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(c1_) : "a"(acc[i][j][1]));           \
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(c2_) : "a"(acc[i][j][2]));           \
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(c3_) : "a"(acc[i][j][3]));           \
        return f32x4{c0_, c1_, c2_, c3_};                                                    \
    }())

    // ---- store-only epilogue of one 64-feature half (= lean_epilogue of gemm_fast.hip for one of its waves)
    // (Non-temporal stores -- what the vendor kernel uses -- measured: QKV - 6 % and FF2 - 6 % in isolation, FF1 + 3 %; in situ at C2 every
    //  combination LOST 0.5 - 1 %: the kernels behind read the tile back.  gpurun_out/r4u_w4.log, r4z_ab.log.  Not kept.)
    // ROPE: this half is a q / k head that receives RoPE -- a wave-uniform property, taken as a BRANCH between two builds of the loop: in
    // F5TTS_Base (pe_attn_head = 1) 2 of the 48 heads of q | k | v rotate, and with a per-element select (what the first builds of this file
    // did, as the 8-wave kernel does) the other 46 paid for the rotation's 32 packed multiply-adds and 16 selects per step all the same.
    [[maybe_unused]] auto store_half_impl = [&](auto hc, auto actc, auto ropec) {
        constexpr int h = decltype(hc)::value, ACT = decltype(actc)::value;
        constexpr bool rope_wave = decltype(ropec)::value;
        const int nb = n0 + wn * 128 + h * 64;
        bf16_t* orow = reinterpret_cast<bf16_t*>(p.out_t) + (size_t)(m0 + wm * WROWS + fr) * p.ldo + nb + 16 * (fq & 1) + 8 * (fq >> 1);
        const size_t jstride = (size_t)16 * p.ldo;
        [[maybe_unused]] f32x4 rp[2][4];
        [[maybe_unused]] int pos0 = 0;
        auto load_rope = [&](auto jc, f32x4 (&dst)[4]) {
            constexpr int j = decltype(jc)::value;
            int pos = pos0 + 16 * j;
            pos -= pos >= p.rows_per_batch ? p.rows_per_batch : 0;  // rows_per_batch >= 128 (launcher)
            const float* t = p.rope + (size_t)pos * 64 + 4 * fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const f32x4*>(t + 16 * i);
        };
        if constexpr (rope_wave) {
            pos0 = (p.row0 + m0 + wm * WROWS + fr) % p.rows_per_batch;
            load_rope(std::integral_constant<int, 0>{}, rp[0]);
        }
        // LayerNorm fold: the half's column constants are read from the staging area ONCE (32 registers), a token tile's (mean, rstd) one tile
        // ahead.  (Read inside lnf_apply -- per feature tile and token tile, as the 8-wave kernel does -- a lone wave waited for the LDS in front of
        // every use: 12 reads and 8 waits per step.)
        [[maybe_unused]] f32x4 c1h[4], c2h[4];
        [[maybe_unused]] f32x2 stj[2];
        if constexpr (LNF) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                c1h[ii] = *reinterpret_cast<const f32x4*>(lnf_lds + ((h * 4 + ii) * 16 + 4 * fq) * 4);
                c2h[ii] = *reinterpret_cast<const f32x4*>(lnf_lds + 512 + ((h * 4 + ii) * 16 + 4 * fq) * 4);
            }
            stj[0] = *reinterpret_cast<const f32x2*>(lnf_lds + 1024 + fr * 8);
        }
        static_for<MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LNF && j + 1 < MI) stj[(j + 1) & 1] = *reinterpret_cast<const f32x2*>(lnf_lds + 1024 + ((j + 1) * 16 + fr) * 8);
            if constexpr (rope_wave && j + 1 < MI) load_rope(std::integral_constant<int, j + 1>{}, rp[(j + 1) & 1]);
            f32x4 vals[4];
            static_for<4>([&](auto ic) {
                constexpr int ii = decltype(ic)::value, i = h * 4 + ii;
                f32x4 v = W4_ACC(i, j);
                if constexpr (LNF) v = epi_lnf4(v, stj[j & 1][0], stj[j & 1][1], c1h[ii], c2h[ii]);
                if constexpr (ACT == ACT_GELU_TANH) v = epi_gelu_tanh4(v);
                if constexpr (rope_wave) v = epi_rope4(v, rp[j & 1][ii]);  // x_transformers apply_rotary_pos_emb: adjacent pairs, fp32 math
                vals[ii] = v;
            });
            const u32x4 q0 = pair_swap(to_bf16x4(vals[0]), to_bf16x4(vals[1]));
            const u32x4 q1 = pair_swap(to_bf16x4(vals[2]), to_bf16x4(vals[3]));
            bf16_t* o = orow + j * jstride;
            *reinterpret_cast<u32x4*>(o) = q0;
            *reinterpret_cast<u32x4*>(o + 32) = q1;
        });
    };
    [[maybe_unused]] auto store_half = [&](auto hc, auto actc) {
        if constexpr (EPI == EPI_ROPE_T) {
            const int nb = n0 + wn * 128 + decltype(hc)::value * 64;
            const int part = nb / p.rope_inner;
            if (part < 2 && ((nb - part * p.rope_inner) >> 6) < p.rope_heads) {
                store_half_impl(hc, actc, std::true_type{});
                return;
            }
        }
        store_half_impl(hc, actc, std::false_type{});
    };
    // ---- in-place update of the fp16 residual stream (= lean_resid_f16 of gemm_fast.hip for two of its waves), with the LayerNorm fold's partial
    // row statistics of the values just stored.  bias[N] and gate[N] of the launch live in LDS (staged once at kernel start), not in registers, and
    // the next tile's first fragments are read behind the epilogue: the registers go to the stream tile's loads in flight.
    [[maybe_unused]] const float* const lds_bias = reinterpret_cast<const float*>(smem + EPI_LDS);
    [[maybe_unused]] const float* const lds_gate = lds_bias + p.N;
    [[maybe_unused]] auto emit_stats = [&](int nb, int m /* row inside the wave's 128 */, float pivot, const f32x4& h0, const f32x4& h1, const f32x4& h2, const f32x4& h3) {
        const f32x4 npv{-pivot, -pivot, -pivot, -pivot};  // (h + (-pivot) == h - pivot bit for bit; the add form packs into v_pk_add_f32)
        const f32x4 d0 = h0 + npv, d1 = h1 + npv, d2 = h2 + npv, d3 = h3 + npv;
        const f32x4 a = (d0 + d1) + (d2 + d3);
        const f32x4 q = __builtin_elementwise_fma(d3, d3, __builtin_elementwise_fma(d2, d2, __builtin_elementwise_fma(d1, d1, d0 * d0)));
        const float s1 = (a[0] + a[1]) + (a[2] + a[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
        float a0 = s1, b0 = s2;  // (inline asm: gemm_fast.hip, emit_stats)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
        float t = a0 + b0, z = 0.0f;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(z));
        const float tot = t + z;
        if (fq < 2) {
            char* sb = reinterpret_cast<char*>(p.stats_out + ((size_t)(nb >> 6) * p.stats_ld + (m0 + wm * WROWS)) * 2);  // (scalar base + 32-bit lane offset)
            *reinterpret_cast<float*>(sb + (unsigned)((m * 2 + fq) * 4)) = tot;
        }
    };
    [[maybe_unused]] auto resid_tile = [&]() {
        // (the lane's row index through an opaque asm: otherwise every per-lane address of the epilogue is hoisted out of the tile loop as a
        //  64-bit loop invariant and spilled across the main loop, where the registers are the fragments')
        int fr = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(fr), "+v"(fq));
        // (one scalar base + eight 32-bit lane offsets, one per token tile; halves and feature-tile pairs are immediates: 64-bit addresses per
        //  access would cost 64 registers here)
        char* const sbase = reinterpret_cast<char*>(p.out_f) + ((size_t)(m0 + wm * WROWS) * p.ldof + n0 + wn * 128) * 2;
        unsigned offj[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) offj[j] = (unsigned)(((fr + 16 * j) * p.ldof + 16 * (fq & 1) + 8 * (fq >> 1)) * 2);
        unsigned keepbits = 0xffu;
        if (p.rowmask && p.rowbits) keepbits = ((unsigned)p.rowbits[((m0 + wm * WROWS) >> 7) * 16 + fr] >> (((m0 + wm * WROWS) & 127) >> 4)) & ((1u << MI) - 1u);
        float piv[MI];
        if (p.stats_out) {
            const float* pbase = p.stats_pivot ? p.stats_pivot + (size_t)(m0 + wm * WROWS) * 2 : nullptr;  // (scalar base + 32-bit lane offset)
#pragma unroll
            for (int j = 0; j < MI; ++j) piv[j] = pbase ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pbase) + (unsigned)((fr + 16 * j) * 8)) : 0.0f;
        }
        // A lone wave cannot hide a load behind another wave's work: the stream tile is requested AHEAD (half, token tile) steps = 2 AHEAD loads
        // ahead of its use -- the first AHEAD before the first store, the others one by one behind a step's stores (a load issued behind a store also
        // waits for that store, vmcnt retires in order; AHEAD steps later the store is long acknowledged).  More than 6 spills accumulators.
        u32x4 xs[2][MI][2];  // [half][token tile][feature-tile pair]: the stream tile as stored (8 fp16 per lane and entry)
        auto load_step = [&](auto sc) __attribute__((always_inline)) {
            constexpr int h = decltype(sc)::value / MI, j = decltype(sc)::value % MI;
            xs[h][j][0] = *reinterpret_cast<const u32x4*>(sbase + offj[j] + h * 128);
            xs[h][j][1] = *reinterpret_cast<const u32x4*>(sbase + offj[j] + h * 128 + 64);
        };
        constexpr int AHEAD = 4;
        static_for<AHEAD>([&](auto sc) { load_step(sc); });
        __builtin_amdgcn_sched_barrier(0);
        const bool all_kept = __builtin_amdgcn_ballot_w64(keepbits != ((1u << MI) - 1u)) == 0ull;  // wave-uniform: no masked row in this wave's 128 token rows
        auto widen_h = [](unsigned lo, unsigned hi) __attribute__((always_inline)) {
            const f16x4_t hv = __builtin_bit_cast(f16x4_t, u32x2{lo, hi});
            return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        };
        auto to_h16 = [](const f32x4& v) __attribute__((always_inline)) {  // saturating fp16 store form
            f16x4_t hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
            return hv;
        };
        auto h_f32 = [](const f16x4_t& hv) __attribute__((always_inline)) { return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}; };
        // partial row statistics: one scalar base per half, the row's 32-bit offset per token tile
        auto steps = [&](auto keptc) __attribute__((always_inline)) {
            constexpr bool ALL = decltype(keptc)::value;
            f32x4 g4[4];  // the half's gate values, read from LDS once per half (inside the step a lone wave waits for them in front of their use)
            static_for<2 * MI>([&](auto sc) __attribute__((always_inline)) {
                constexpr int h = decltype(sc)::value / MI, j = decltype(sc)::value % MI;
                const int nb = n0 + wn * 128 + h * 64;
                if constexpr (j == 0) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) g4[ii] = *reinterpret_cast<const f32x4*>(lds_gate + nb + ii * 16 + 4 * fq);
                }
                f32x4 x0, x1, x2, x3;
                {
                    const u32x4 q = xs[h][j][0];  // inverse of pair_swap: this lane's 8 stored features -> the accumulator layout
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                    x0 = widen_h(s0[0], s1[0]);
                    x1 = widen_h(s0[1], s1[1]);
                }
                {
                    const u32x4 q = xs[h][j][1];
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                    x2 = widen_h(s0[0], s1[0]);
                    x3 = widen_h(s0[1], s1[1]);
                }
                const f32x4 a0 = W4_ACC(h * 4 + 0, j), a1 = W4_ACC(h * 4 + 1, j), a2 = W4_ACC(h * 4 + 2, j), a3 = W4_ACC(h * 4 + 3, j);
                f32x4 v0 = epi_axpy4(a0, g4[0], x0), v1 = epi_axpy4(a1, g4[1], x1), v2 = epi_axpy4(a2, g4[2], x2), v3 = epi_axpy4(a3, g4[3], x3);
                if constexpr (!ALL) {  // masked query rows keep their value
                    const bool keep = (keepbits >> j) & 1u;
                    v0 = keep ? v0 : x0;
                    v1 = keep ? v1 : x1;
                    v2 = keep ? v2 : x2;
                    v3 = keep ? v3 : x3;
                }
                const f16x4_t h0 = to_h16(v0), h1 = to_h16(v1), h2 = to_h16(v2), h3 = to_h16(v3);
                *reinterpret_cast<u32x4*>(sbase + offj[j] + h * 128) = pair_swap(__builtin_bit_cast(bf16x4, h0), __builtin_bit_cast(bf16x4, h1));
                *reinterpret_cast<u32x4*>(sbase + offj[j] + h * 128 + 64) = pair_swap(__builtin_bit_cast(bf16x4, h2), __builtin_bit_cast(bf16x4, h3));
                if (p.stats_out) emit_stats(nb, fr + 16 * j, piv[j], h_f32(h0), h_f32(h1), h_f32(h2), h_f32(h3));
                if constexpr (decltype(sc)::value + AHEAD < 2 * MI) load_step(std::integral_constant<int, decltype(sc)::value + AHEAD>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (all_kept)
            steps(std::true_type{});
        else
            steps(std::false_type{});
    };

    if constexpr (EPI == EPI_RESID) {  // bias[N] | gate[N] of the launch into LDS (visible to every wave behind the prologue's barrier)
        float* lb = reinterpret_cast<float*>(smem + EPI_LDS);
        for (int i = tid; i < p.N; i += 256) {
            lb[i] = p.bias[i];
            lb[p.N + i] = p.gate ? p.gate[i] : 1.0f;
        }
    }
    // ---- prologue: stages 0 and 1 requested, stage 0 landed, first fragments read
    auto issue_all = [&](int X) {
#pragma unroll
        for (int pc = 0; pc < MI; ++pc) dma16w(X * OPA + wdstA + pc * 1024, voffA[pc], baseA);
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dma16w(X * OPB + wdstW + pc * 1024, voffW[pc], baseW);
    };
    if constexpr (LNF) stage_lnf();  // (in front of the operand requests: the counted wait below then does not depend on how many it issues)
    issue_all(0);
    front_advance();
    issue_all(1);
    front_advance();
    f32x4 bstart[8];
    if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bstart[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        load_bias(n0, bstart);
    }
    acc_from(bstart);
    if constexpr (EPI == EPI_RESID) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this thread's part of the bias / gate image is written)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NP) : "memory");  // stage 0 has landed for everyone
    static_for<NR>([&](auto ec) { rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ec); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // a tile: nk iterations (nk even, >= 4), buffer = iteration & 1
    for (int t = 0; t < my_tiles; ++t) {
        const bool last = t + 1 == my_tiles;
        if (t == 0)
            body(X0{}, T{});  // (the very first iteration waits: stage 1 was requested just now)
        else
            body(X0{}, F{});  // a tile's first iteration behind an epilogue
        body(X1{}, T{});
        for (int kt = 2; kt < nk; kt += 2) {
            body(X0{}, T{});
            body(X1{}, T{});
        }
        int nm0 = m0, nn0 = n0;
        if (!last) tile_mn(blockIdx.x + (t + 1) * G, nm0, nn0);
        if constexpr (EPI == EPI_RESID) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stages requested ahead have landed: from here on only the epilogue's own traffic
            __builtin_amdgcn_sched_barrier(0);
            resid_tile();
            __builtin_amdgcn_sched_barrier(0);
            f32x4 bn[8];  // the next tile's accumulator start: its features' bias, from LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) bn[i] = *reinterpret_cast<const f32x4*>(lds_bias + nn0 + wn * 128 + i * 16 + 4 * fq);
            acc_from(bn);
        } else {
            // the NEXT tile's accumulator start is requested before the stores and waited for while only loads are in flight (loads and stores
            // share vmcnt: a wait behind the stores would hold the wave until they are acknowledged)
            f32x4 bn[8];
            if constexpr (LNF) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                load_bias(nn0, bn);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(bn[i]));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and the stages requested ahead (and this tile's fold operands) have landed
            __builtin_amdgcn_sched_barrier(0);
            finish_stats();
            if (p.act == ACT_GELU_TANH) {
                store_half(X0{}, std::integral_constant<int, ACT_GELU_TANH>{});
                store_half(X1{}, std::integral_constant<int, ACT_GELU_TANH>{});
            } else {
                store_half(X0{}, std::integral_constant<int, ACT_NONE>{});
                store_half(X1{}, std::integral_constant<int, ACT_NONE>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            acc_from(bn);
        }
        m0 = nm0;
        n0 = nn0;
        if (!last) {
            // the next tile's first fragments (its stage 0 sits in buffer 0: nk is even), and its fold operands
            static_for<NR>([&](auto ec) { rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ec); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (also: the epilogue's reads of the staging area are done before it is rewritten)
            if constexpr (LNF) stage_lnf();
        }
    }
    if constexpr (MI == 4) {
        if (((pfv[0] ^ pfv[1]) ^ (pfv[2] ^ pfv[3])) == 0x7fc0dead && (p.pf_n[0] | p.pf_n[1]) == 0xffffffffu) p.stats_out[0] = 0.f;  // (never true: keeps the prefetch loads alive)
    }
#undef W4_ACC
#undef W4_DSR
}
}  // namespace

int g_gemm_w4 = 1;  // tuning knob ("gemm_w4"): 1 = whole-tile block linears with enough tiles for the CUs (w4_tile_rows) run on the one-wave-per-SIMD kernel
int g_gemm_w4_ink = 1;  // tuning knob ("gemm_w4_ink"): folded projections on the kernel's 128-row tiles finish the row statistics themselves (no statistics launch)
int g_gemm_w4_bm = 0;  // diagnostic knob ("gemm_w4_bm"): token rows per tile, 0 = by tile count, 128 / 256 forced where the shape allows

int gemm_persist_grid();  // gemm_fast.hip

// token rows per tile of this launch on the one-wave-per-SIMD kernel (0: the launch does not take it).  Measured at 1 - 16 utterances x 1024 frames
// (gpurun_out/r4ak_ab.log, r4am_ab.log; t256 / t128 = tiles of the two heights, as a share of the CUs):
//   * 256 rows wherever they reach three quarters of the CUs -- the tall tile wins even with a quarter of the CUs idle against the short one on all of
//     them (3 x 1024 FF1: 192 tall tiles 32.7 us, 384 short ones 34.3; 8 x 1024 out-projection: 256 tall 42.7, 512 short 48.5: the short tile moves
//     1.5 x the operand bytes per FLOP through LDS) -- but not at half (4 x 1024 FF2: 128 tall tiles 50.3 us, 256 short ones 37.5);
//   * otherwise 128 rows from half the CUs on (1 x 1024: the fused projection on 192 short tiles 20.2 against 25.8 us on the 8-wave kernel's narrow
//     tiles, FF1 on 128 short tiles 18.4 against 21.0) -- not at a quarter (1 x 1024 FF2: 30.7 against 24.9).
static int w4_tile_rows(const GemmParams& p) {
    const int pg = gemm_persist_grid();
    const int t256 = p.M % 256 == 0 ? (p.M / 256) * (p.N / 256) : 0, t128 = p.M % 128 == 0 ? (p.M / 128) * (p.N / 256) : 0;
    if (g_gemm_w4_bm == 256) return t256 >= pg ? 256 : 0;
    if (g_gemm_w4_bm == 128) return t128 >= pg ? 128 : 0;
    if (4 * t256 >= 3 * pg) return 256;
    return 2 * t128 >= pg ? 128 : 0;
}

// LayerNorm fold: a consumer launch of this shape can finish the row statistics itself on this kernel -- returns the tile height that does (128: the
// partial planes travel through the LDS the short tile's stage buffers leave), 0 otherwise; dit_eval then drops the statistics launch in front of it
int gemm_w4_lnf_inkernel(int M, int N, int K) {
    GemmParams t;
    memset(&t, 0, sizeof(t));
    t.M = M; t.N = N; t.K = K;
    if (!g_gemm_w4 || !g_gemm_w4_ink || K != 1024 || N % 256 != 0) return 0;
    const int rows = w4_tile_rows(t);
    if (rows == 128) return 128;
    return 0;
}

bool gemm_w4_ok(const GemmParams& p, int mode, int epi) {
    if (!g_gemm_w4 || mode != GEMM_DENSE) return false;
    if (p.N % 256 != 0 || p.K % 128 != 0 || p.K < 256 || (p.lda & 7) || (p.ldw & 7) || p.a_row_mod != 0 || p.row0 != 0) return false;
    if (w4_tile_rows(p) == 0) return false;  // small launches: the 8-wave kernel's narrower tiles
    if ((size_t)255 * (size_t)(p.lda > p.ldw ? p.lda : p.ldw) * 2 + 128 > 0x7fffffffull) return false;
    if (p.fin_counter) return false;  // (weight-prefetch ranges, pf_p: served by the 128-row tile, which is the one small launches take)
    if (p.lnf_partial && !(gemm_w4_lnf_inkernel(p.M, p.N, p.K) != 0 && p.lnf_stats && p.lnf_ncols == 16 && p.lnf_partial_ld >= p.M)) return false;
    const bool lnf = p.lnf_stats != nullptr;
    if (epi == EPI_STORE_T || epi == EPI_ROPE_T) {
        if (!p.out_t || (p.ldo & 7) || !(p.act == ACT_NONE || p.act == ACT_GELU_TANH)) return false;
        if (lnf ? (!p.lnf_c1 || !p.lnf_c2) : !p.bias) return false;
        if (epi == EPI_ROPE_T && (p.rows_per_batch < 128 || !p.rope || p.rope_inner % 64 != 0)) return false;
        return true;
    }
    if (epi == EPI_RESID) {
        if (lnf || !p.add2_f16 || !p.out_f || (p.ldof & 7) || p.act != ACT_NONE || !p.bias) return false;
        if ((p.gate && p.gate_bstride != 0) || (p.rowmask && !p.rowbits)) return false;
        if (p.stats_out && (p.N % 64 != 0 || p.stats_ld < p.M)) return false;
        if ((size_t)p.N * 8 > (size_t)EPI_LDS_BYTES) return false;  // bias[N] | gate[N] live in LDS
        return true;
    }
    return false;
}

template <int MI> static int launch_w4(const GemmParams& p, int epi, hipStream_t stream) {
    const int tiles_n = p.N / 256, nblocks = (p.M / (MI * 32)) * tiles_n;
    const int pg = gemm_persist_grid();
    const dim3 grid(nblocks < pg ? nblocks : pg), block(256);
    const bool lnf = p.lnf_stats != nullptr;
    if (epi == EPI_STORE_T && lnf)
        hipLaunchKernelGGL((gemm_w4_kernel<EPI_STORE_T, true, MI>), grid, block, 0, stream, p, tiles_n, nblocks);
    else if (epi == EPI_STORE_T)
        hipLaunchKernelGGL((gemm_w4_kernel<EPI_STORE_T, false, MI>), grid, block, 0, stream, p, tiles_n, nblocks);
    else if (epi == EPI_ROPE_T && lnf)
        hipLaunchKernelGGL((gemm_w4_kernel<EPI_ROPE_T, true, MI>), grid, block, 0, stream, p, tiles_n, nblocks);
    else if (epi == EPI_ROPE_T)
        hipLaunchKernelGGL((gemm_w4_kernel<EPI_ROPE_T, false, MI>), grid, block, 0, stream, p, tiles_n, nblocks);
    else if (epi == EPI_RESID)
        hipLaunchKernelGGL((gemm_w4_kernel<EPI_RESID, false, MI>), grid, block, 0, stream, p, tiles_n, nblocks);
    else
        return f5_fail(F5_EINVAL, "gemm_w4: unsupported epilogue %d", epi);
    F5_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_w4(const GemmParams& p, int epi, hipStream_t stream) {
    return w4_tile_rows(p) == 128 ? launch_w4<4>(p, epi, stream) : launch_w4<8>(p, epi, stream);
}
