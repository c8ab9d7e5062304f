// gemm_fast.hip -- tuned bf16 GEMM for gfx950:  out[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate.
//
// Structure (MI355X-first, see /opt/skills/guides/cdna_hip_programming.md section 5):
//   * 256-token x BN-feature output tile per 512-thread workgroup (8 wavefronts, one workgroup per CU, 2 waves per SIMD),
//     v_mfma_f32_16x16x32_bf16 with the weight rows on the MFMA row index, so each lane finishes with 4
//     consecutive output features of one token (vector stores, lane-local RoPE pairs);
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR staging) through a ring of 32-deep
//     K-steps (5 slots = the whole 160 KiB LDS for the 256-wide tile): four K-steps are in flight while one is consumed
//     (counted s_waitcnt vmcnt, never 0 in the loop), raw s_barriers;
//   * schedules (template VAR): 1 = the two waves of a SIMD staggered by one barrier interval (one feeds the matrix pipe while
//     the other issues DMA and reads fragments); 30 = the same on a PERSISTENT grid (one workgroup per CU walks tiles, the DMA
//     front crosses into the next tile under the epilogue) -- the production schedule for whole-tile block linears;
//     0 = plain ring (64-wide tiles; the 128-wide tile runs schedule 1: FF2 at M = 8192 44.6 -> 41.5 us, the 64-wide one gains nothing).  Measured alternatives (paired / split DMA issue, half-slab ring, store flavours, a 4-wave
//     128x128-per-wave kernel) and the timing-only ablation builds are recorded in DESIGN.md and profiles/r1_06..r1_10; their code
//     left the library in round 2 (git history: 6761115);
//   * every accumulator starts from its feature's bias (identical fp32 sums in every variant and tile width); whole tiles take
//     the lean store-only epilogue, ragged tiles the generic one;
//   * LDS image is lane-linear (a DMA instruction writes 16 rows x 64 B); bank conflicts of the ds_read_b128 fragment
//     reads are removed by an XOR swizzle applied on the *source* address and on the read (16-byte chunk ^= (-(row>>2))&3);
//   * blockIdx -> tile mapping is XCD-aware (each XCD's L2 sees a contiguous band of token tiles x all feature tiles);
//   * GEMM_CONV31: the same engine as an implicit GEMM for the grouped Conv1d(k=31): K-steps walk (tap, 64 channels),
//     the DMA source row is shifted by tap-15 and rows outside the utterance read a zero page.
//   * rows >= M / features >= N are clamped on load and dropped in the epilogue, so no padding contract on the caller.
#include "gemm_tile.h"
#include "lnf_stats_math.h"

// This file is compiled as TWO translation units (build time): gemm_fast.hip itself holds every instantiation without the LayerNorm fold,
// gemm_fast_lnf.hip (#define F5_LNF_TU + #include of this file) the LNF ones behind launch_gemm_fast_lnf().
#ifdef F5_LNF_TU
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];  // (GEMM_CONV31 only; never read in the LNF instantiations)
#else
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];  // zero-initialised
#endif
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LNF (LayerNorm fold, gemm.h): fp16 operands on v_mfma_f32_16x16x32_f16 (same rate and fragment layout as the bf16 shape), accumulator
// start 0, epilogue value rstd (acc - mean c1) + c2 in front of the activation / RoPE.  EPI_STORE_T and EPI_ROPE_T only.
template <int BN, int WM, int MODE, int EPI, int VAR, int NS, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm_fast_kernel(GemmParams p, int tiles_n, int nblocks) {
    static_assert(!LNF || (MODE == GEMM_DENSE && (EPI == EPI_STORE_T || EPI == EPI_ROPE_T)), "LayerNorm fold: QKV (+ RoPE) and FF1 (+ GELU) only");
    constexpr int BK = 32, WN = 64, NSTAGE = NS;
    constexpr int WAVES_N = BN / WN;
    constexpr int BM = (8 / WAVES_N) * WM;  // 256 token rows (WM = 128 / 64 / 32 for BN = 256 / 128 / 64); 128 for the small-launch tiles (BN = 128, WM = 32; BN = 64, WM = 16)
    constexpr int APW = BM / 128;           // activation DMA pieces (16 rows x 64 B) per wave per K-step
    static_assert(BM == 256 || BM == 128, "token-tile heights");
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    constexpr int WJ = BN >= 256 ? 2 : 1;  // weight DMA pieces (16 rows x 64 B) per wave per K-step (BN = 64: waves 4-7 duplicate 0-3)
    constexpr int PPW = APW + WJ;          // DMA pieces per wave per K-step (vmcnt bookkeeping)
    static_assert((BM / WM) * WAVES_N == 8, "8 waves");
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    if (p.clk) {  // diagnostic: the clock the chip holds under this kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            p.clk[(size_t)blockIdx.x * 4 + 0] = t0;
            p.clk[(size_t)blockIdx.x * 4 + 1] = r0;
        }
    }
    // weight prefetch for the launches behind this one (GemmParams::pf_p): thread t of workgroup b touches line 512 b + t of each range; the
    // loads are the oldest entries of the vector-memory queue (they retire before any operand piece) and their value is only looked at after
    // the epilogue
    [[maybe_unused]] unsigned pf0 = 0u, pf1 = 0u;  // (two registers, no arithmetic before the end: an xor here would wait for the loads)
    if constexpr (VAR != 30) {                     // (small launches only; the persistent builds have no register to spare)
        if (p.pf_n[0] | p.pf_n[1]) {
            const size_t line = (size_t)blockIdx.x * 512u + tid;
            if (line * 128u < p.pf_n[0]) pf0 = *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.pf_p[0]) + line * 128u);
            if (line * 128u < p.pf_n[1]) pf1 = *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.pf_p[1]) + line * 128u);
        }
    }
    auto clk_end = [&]() {
        if constexpr (VAR != 30) {
            if ((pf0 ^ pf1) == 0x7fc0dead && (p.pf_n[0] | p.pf_n[1]) == 0xffffffffu) p.clk[0] = pf0;  // (never true: keeps the prefetch loads alive)
        }
        if (p.clk) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) {
                p.clk[(size_t)blockIdx.x * 4 + 2] = t1;
                p.clk[(size_t)blockIdx.x * 4 + 3] = r1;
            }
        }
    };

    // XCD-aware bijective remap (blocks b and b+8 share an XCD): give each XCD a contiguous band of tiles.
    // Inside an XCD band, walk groups of `gm` token tiles x all feature tiles with the token tile fastest: the ~32 tiles an
    // XCD runs together then form a squarer (gm x 32/gm) patch of the output, which minimises the distinct operand slices
    // its L2 has to hold per K-step
    auto tile_mn = [&](int bid, int& tm0, int& tn0) {
        const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
        int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (p.tile_reverse) swz = nblocks - 1 - swz;
        const int gm = p.tile_group > 0 ? p.tile_group : 1;
        const int tiles_m_all = nblocks / tiles_n;
        const int grp = swz / (gm * tiles_n);
        const int gsz = min(gm, tiles_m_all - grp * gm);  // last group may be short
        const int rin = swz - grp * gm * tiles_n;
        const int tile_n = rin / gsz, tile_m = grp * gm + (rin - tile_n * gsz);
        tm0 = tile_m * BM;
        tn0 = tile_n * BN;
    };
    static_assert(VAR == 0 || VAR == 1 || VAR == 30, "schedules: 0 plain ring, 1 staggered wave groups, 30 staggered + persistent grid");
    constexpr bool P30 = VAR == 30;      // production persistent schedule
    constexpr bool PERSIST = VAR == 30;  // persistent grid: a block walks tiles bid, bid + gridDim.x, ... and prefetches across tile boundaries
    int m0, n0;                          // tile being computed (epilogue side)
    tile_mn(blockIdx.x, m0, n0);

    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W);

    // ---- DMA source bookkeeping: 2 A pieces + WJ W pieces per wave per K-step, each piece = 16 rows x 64 B
    const int prow = lane >> 2, pchunk = lane & 3;
    const int plc = pchunk ^ ((0 - (prow >> 2)) & 3);  // logical 16-byte chunk stored at this physical slot (piece rows are 16-aligned)
    const bf16_t* a_src[APW];
    int a_pos[APW];
    bool a_ok[APW];
    const bf16_t* w_src[WJ];
    int w_piece[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) w_piece[j] = (wave * WJ + j) % (BN / 16);
    auto setup_src = [&](int sm0, int sn0) {  // DMA sources of tile (sm0, sn0)
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            int gm = sm0 + (wave * APW + j) * 16 + prow;
            a_ok[j] = gm < p.M;
            if (gm >= p.M) gm = p.M - 1;
            a_pos[j] = 0;
            if constexpr (MODE == GEMM_DENSE) {
                if (p.a_row_mod > 0) gm %= p.a_row_mod;
            } else {
                a_pos[j] = gm % p.rows_per_batch;
            }
            a_src[j] = A + (size_t)gm * p.lda + plc * 8;
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            int gn = sn0 + w_piece[j] * 16 + prow;
            if (gn >= p.N) gn = p.N - 1;
            if constexpr (MODE == GEMM_DENSE)
                w_src[j] = W + (size_t)gn * p.ldw + plc * 8;
            else
                w_src[j] = W + (size_t)gn * p.conv_win + plc * 8;
        }
    };
    setup_src(m0, n0);
    const int cslices = MODE == GEMM_CONV31 ? p.conv_win / BK : 1;
    const int nk = MODE == GEMM_CONV31 ? 31 * cslices : p.K / BK;
    const int win0 = MODE == GEMM_CONV31 ? (n0 / p.conv_cg) * p.conv_cg : 0;
    const int L = p.rows_per_batch;

    // piece ids of one K-step for this wave: 0,1 = activation pieces, 2.. = weight pieces
    auto issue_piece = [&](int slot, int kt, int piece) {
        char* sbase = smem + slot * STAGE;
        if constexpr (MODE == GEMM_DENSE) {
            if (piece < APW)
                dma16(a_src[piece < APW ? piece : 0] + (size_t)kt * BK, sbase + (wave * APW + piece) * 1024);
            else
                dma16(w_src[piece >= APW ? piece - APW : 0] + (size_t)kt * BK, sbase + A_BYTES + w_piece[piece >= APW ? piece - APW : 0] * 1024);
        } else {
            const int tap = kt / cslices, sl = kt - tap * cslices;
            const int ch0 = win0 + sl * BK;
            if (piece < APW) {
                const int ap = piece < APW ? piece : 0;
                const int sp = a_pos[ap] + tap - 15;  // Conv1d(padding=15): zero outside [0, L) of this utterance
                const bool ok = a_ok[ap] && sp >= 0 && sp < L && (ch0 + plc * 8) < p.N;
                const void* src = ok ? (const void*)(a_src[ap] + (ptrdiff_t)(tap - 15) * p.lda + ch0) : (const void*)g_zero_page;
                dma16(src, sbase + (wave * APW + piece) * 1024);
            } else {
                const int wp = piece >= APW ? piece - APW : 0;
                dma16(w_src[wp] + (size_t)tap * p.N * p.conv_win + sl * BK, sbase + A_BYTES + w_piece[wp] * 1024);
            }
        }
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int pc = 0; pc < PPW; ++pc) issue_piece(kt % NSTAGE, kt, pc);
    };
    // ---- fragment read offsets (bytes inside a stage)
    const int fr = lane & 15, fq = lane >> 4;
    const int c0 = (fq ^ ((0 - (fr >> 2)) & 3)) * 16;
    const int a_off = (wm * WM + fr) * 64 + c0;
    const int w_off = A_BYTES + (wn * WN + fr) * 64 + c0;

    // ---- epilogue operands that do not depend on the token are fetched at the start of a tile, so their latency hides under the main loop
    int ncol[NI];
    bool okn[NI];
    f32x4 bias4[NI];
    [[maybe_unused]] f32x4 gate4[NI];
    [[maybe_unused]] bool rope_wave = false;
    bool wide_ok;
    int nwide;
    [[maybe_unused]] unsigned keepbits = 0xffu;  // lean epilogue: bit j = row (j * 16 + fr) of this wave's token rows is kept
    // LNF: the epilogue operands of this wave -- the column constants of its 64 features (c1 = row sums of W', c2 = bias + W . shift: 2 x 256
    // bytes, two dword LDS-DMA instructions) and (mean, rstd) of its WM token rows (8 bytes a row, one dwordx4 LDS-DMA instruction) -- are
    // staged in a 2-KiB LDS area of the wave's own and read back inside the epilogue: 48 registers fewer than holding them (the QKV + RoPE
    // build spilled with them in registers).  Persistent schedule: the area is the start of the wave's OWN activation pieces in the ring slot
    // the DMA front writes next (it holds no live stage; only this wave writes there again, behind its own reads in program order, so no
    // barrier is involved).  Other schedules: the ring is dead after the main loop (one barrier), the area is smem + wave * 2048.
    // Tiles with at most 4 token tiles per wave (every tile but the 256-wide one) have the registers to HOLD the operands instead (40 at most):
    // they are loaded with the other per-tile epilogue operands before the main loop, nothing is exposed and no barrier is needed -- the small
    // launches that take these tiles are latency-bound.
    constexpr bool LNF_LDS = LNF && (WM / 16) >= 8;
    [[maybe_unused]] f32x4 c1r[LNF && !LNF_LDS ? NI : 1], c2r[LNF && !LNF_LDS ? NI : 1];
    [[maybe_unused]] f32x2 lstr[LNF && !LNF_LDS ? WM / 16 : 1];
    [[maybe_unused]] char* lnf_lds = nullptr;
    [[maybe_unused]] auto stage_lnf = [&](int tm0, int tn0, char* area) {
        lnf_lds = area;
        int n = tn0 + wn * WN + lane;
        n = n < p.N ? n : 0;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.lnf_c1 + n), (lptr_t)lnf_lds, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.lnf_c2 + n), (lptr_t)(lnf_lds + 256), 4, 0, 0);
        // rows tm0 + wm * WM .. + WM - 1: lane l fetches rows 2l, 2l + 1 (the statistics buffer is padded, rows >= M are never used)
        if (WM >= 128 || lane < WM / 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.lnf_stats + ((size_t)(tm0 + wm * WM) + 2 * lane) * 2), (lptr_t)(lnf_lds + 512), 16, 0, 0);
    };
    [[maybe_unused]] auto lnf_apply = [&](const f32x4& a, int i, int j) {
        if constexpr (LNF && !LNF_LDS) return epi_lnf4(a, lstr[j][0], lstr[j][1], c1r[i], c2r[i]);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(lnf_lds + (i * 16 + 4 * fq) * 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(lnf_lds + 256 + (i * 16 + 4 * fq) * 4);
        const f32x2 st = *reinterpret_cast<const f32x2*>(lnf_lds + 512 + (j * 16 + fr) * 8);
        return epi_lnf4(a, st[0], st[1], c1, c2);
    };
    // LayerNorm fold, statistics inside the kernel (gemm.h: lnf_partial): called behind the first operand requests, so that the partial sums'
    // loads fly with them (in front of them they put one whole memory latency on the critical path of a latency-bound launch)
    [[maybe_unused]] auto lnf_inkernel_stats = [&]() {
        if constexpr (LNF && !LNF_LDS) {
            if (p.lnf_partial) {
                // statistics inside the kernel: the lnf_ncols partial sums of each of this lane's rows, added in stats_finalize_kernel's order
                    // (the four lanes that share a row load the same addresses); issued before the main loop, behind the first operand fetch
                    bool bad = false;
                    float worst = 0.f;
                    int worst_row = 0;
#pragma unroll
                    for (int j = 0; j < WM / 16; ++j) {
                        int m = m0 + wm * WM + j * 16 + fr;
                        const bool okrow = m < p.M;
                        m = okrow ? m : p.M - 1;
                        float s1 = 0.f, s2 = 0.f;
                        if (p.lnf_ncols == 16) {  // dim 1024: all sixteen loads of a row in flight together (a runtime loop issues them one behind the other)
                            f32x2 v[16];
#pragma unroll
                            for (int c = 0; c < 16; ++c) v[c] = *reinterpret_cast<const f32x2*>(p.lnf_partial + ((size_t)c * p.lnf_partial_ld + m) * 2);
#pragma unroll
                            for (int c = 0; c < 16; ++c) {
                                s1 += v[c][0];
                                s2 += v[c][1];
                            }
                        } else {
                            for (int c = 0; c < p.lnf_ncols; ++c) {
                                const f32x2 v = *reinterpret_cast<const f32x2*>(p.lnf_partial + ((size_t)c * p.lnf_partial_ld + m) * 2);
                                s1 += v[0];
                                s2 += v[1];
                            }
                        }
                        const float pv = p.lnf_pivot ? p.lnf_pivot[(size_t)m * 2] : 0.0f;
                        float mean, rstd, sumsq;
                        lnf_row_stats(s1, s2, pv, p.K, mean, rstd, sumsq);
                        lstr[j] = f32x2{mean, rstd};
                        if (p.lnf_stats_out && n0 == 0 && wn == 0 && fq == 0 && okrow) {  // feature tile 0: the next producer's pivots, the range guard
                            *reinterpret_cast<f32x2*>(p.lnf_stats_out + (size_t)m * 2) = lstr[j];
                            if (!(sumsq < 65504.0f * 65504.0f)) {
                                bad = true;
                                worst = sumsq;
                                worst_row = m;
                            }
                        }
                    }
                    if (p.lnf_stats_out && n0 == 0 && wn == 0) lnf_raise_guard(p.lnf_sat, bad, worst, p.lnf_sat_tag, worst_row + p.row0);
            }
        }
    };
    auto prep_epilogue = [&]() {
        if constexpr (EPI == EPI_GATE_T || EPI == EPI_RESID) {
            const int mw = m0 + wm * WM;
            if (p.rowmask && p.rowbits && mw < p.M) keepbits = (unsigned)p.rowbits[(mw >> 7) * 16 + fr] >> ((mw & 127) >> 4);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * WN + i * 16 + 4 * fq;
            okn[i] = n + 3 < p.N;  // N % 4 == 0: a lane's 4 features are valid together
            ncol[i] = okn[i] ? n : 0;
            if constexpr (LNF)
                bias4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            else if constexpr (!P30)
                bias4[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ncol[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_GATE_T || EPI == EPI_RESID)
                gate4[i] = (p.gate && p.gate_bstride == 0) ? *reinterpret_cast<const f32x4*>(p.gate + ncol[i]) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
        // fused QKV projection: a wave's 64 features are exactly one head of q, k or v -> RoPE applies to the whole wave tile or not at all
        if constexpr (EPI == EPI_ROPE_T) {
            const int nw = n0 + wn * WN;
            const int part = nw / p.rope_inner;
            rope_wave = part < 2 && ((nw - part * p.rope_inner) >> 6) < p.rope_heads;
        }
        // 16-byte paired stores: after pair_swap a lane in 16-lane row r owns features (r&1 ? tile i+1 : tile i) * 16 + 8*(r>>1) .. +7
        wide_ok = (n0 + wn * WN + WN <= p.N) && (p.ldo & 7) == 0;
        nwide = n0 + wn * WN + 16 * (fq & 1) + 8 * (fq >> 1);
        if constexpr (LNF && !LNF_LDS) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                c1r[i] = *reinterpret_cast<const f32x4*>(p.lnf_c1 + ncol[i]);
                c2r[i] = *reinterpret_cast<const f32x4*>(p.lnf_c2 + ncol[i]);
            }
            if (!p.lnf_partial) {
#pragma unroll
                for (int j = 0; j < WM / 16; ++j) {
                    int m = m0 + wm * WM + j * 16 + fr;
                    m = m < p.M ? m : p.M - 1;
                    lstr[j] = *reinterpret_cast<const f32x2*>(p.lnf_stats + (size_t)m * 2);
                }
            }
        }
    };
    prep_epilogue();

    f32x4 acc[NI][MI];
    // Every accumulator starts from the bias of its feature (all kernel variants and tile widths alike, so that the fp32 sums --
    // and with them the bf16 outputs -- do not depend on which variant a batch size selects).  Called AFTER the first DMA issue:
    // waiting for the bias loads must not delay the operand fetch.
    auto init_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = bias4[i];
    };

    auto read_frags = [&](int kt, bf16x8 (&wf)[NI], bf16x8 (&af)[MI]) {
        const char* sb = smem + (PERSIST ? kt : kt % NSTAGE) * STAGE;  // PERSIST: the caller passes the ring slot
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + w_off + i * 1024);
#pragma unroll
        for (int j = 0; j < MI; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sb + a_off + j * 1024);
    };
    auto mma = [&](const bf16x8 (&wf)[NI], const bf16x8 (&af)[MI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                if constexpr (LNF)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[i]), __builtin_bit_cast(f16x8_t, af[j]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
    };

    // wait until at most `tiles` K-steps' worth of this wave's DMA pieces are still outstanding (vmcnt retires in order)
    auto wait_pieces = [&](int tiles) {
        if (tiles >= 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (tiles == 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (tiles == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // LayerNorm fold, producer side: partial row statistics of the 64 features this wave just stored for token row m (values h = the
    // fp16-ROUNDED stream elements, pivot = the row's previous mean): (sum (h - pivot), sum (h - pivot)^2) over the lane's 16 values, then over
    // the four lanes that share the row (v_permlane16_swap, v_permlane32_swap); lane groups 0 / 1 store the two sums.  The order of the additions is fixed and does not depend on the tile shape (a wave always owns 64 features).
    [[maybe_unused]] auto emit_stats = [&](int m, bool row_ok, float pivot, const f32x4& h0, const f32x4& h1, const f32x4& h2, const f32x4& h3) {
        const f32x4 pv{pivot, pivot, pivot, pivot};
        const f32x4 d0 = h0 - pv, d1 = h1 - pv, d2 = h2 - pv, d3 = h3 - pv;
        const f32x4 a = (d0 + d1) + (d2 + d3);
        const f32x4 q = __builtin_elementwise_fma(d3, d3, __builtin_elementwise_fma(d2, d2, __builtin_elementwise_fma(d1, d1, d0 * d0)));
        const float s1 = (a[0] + a[1]) + (a[2] + a[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
        // The four lanes of a row sit in the four 16-lane rows of the wave.  v_permlane16_swap(s1, s2) leaves [s1 r0, s2 r0, s1 r2, s2 r2] and
        // [s1 r1, s2 r1, s1 r3, s2 r3]: their sum holds s1 of rows 0+1 in row 0, s2 of rows 0+1 in row 1, the same for rows 2+3 in rows 2 / 3.
        // v_permlane32_swap against a zero register then brings the upper half under the lower one: row 0 ends with the total of s1, row 1 with
        // the total of s2 = the two floats of the partial.  INLINE ASM on purpose: with the builtin, hipcc (ROCm 7.2) adds result 0 to ITSELF
        // when the two results of a swap feed one add ("v_permlane16_swap v70, v66; v_add_f32 v66, v70, v70" in the .s of round 4's first
        // build: the sums then covered a quarter of every row, which the op-level fp64 test caught).  s_nop 1: the wait states the compiler's
        // hazard recogniser puts between a vector write of the operands and the swap; it does not look inside an asm statement.
        float a0 = s1, b0 = s2;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
        float t = a0 + b0, z = 0.0f;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(z));
        const float tot = t + z;
        if (fq < 2 && row_ok) {
            float* dst = p.stats_out + ((size_t)((n0 + wn * WN) >> 6) * p.stats_ld + m) * 2 + fq;
            if (p.fin_counter)  // read by another workgroup of THIS launch: agent-scope (write-through) store
                __hip_atomic_store(dst, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *dst = tot;
        }
    };
    [[maybe_unused]] auto h_round4 = [](const f32x4& v) {  // what the fp16 stream holds after a saturating store of v
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)(_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        return r;
    };
    // ---------------------------------------------------------------- epilogue (store-only wherever the call site allows)
    // Loads (row mask, RoPE table, addend / residual) are issued in groups BEFORE any store of the group: vmcnt retires
    // in order, so a load issued behind a store would also wait for that store's round trip.
    auto generic_epilogue = [&]() {
    // token tiles per load group (register budget).  RoPE on the 256-wide tile (MI = 8, 128 accumulators): groups of 4 held 64 table registers
    // beside the lean form's double buffer and spilled (256 VGPRs + 52 B/lane of scratch in the non-persistent builds): 2.
    constexpr int JG0 = (EPI == EPI_RESID || EPI == EPI_GATE_T || (EPI == EPI_ROPE_T && MI >= 8)) ? 2 : (MI >= 4 ? 4 : MI);
    constexpr int JG = JG0 < MI ? JG0 : MI;
    static_for<MI / JG>([&](auto gc) {
        constexpr int j0 = decltype(gc)::value * JG;
        int mrow[JG];
        bool okm[JG];
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            const int m = m0 + wm * WM + (j0 + jj) * 16 + fr;
            okm[jj] = m < p.M;
            mrow[jj] = okm[jj] ? m : p.M - 1;
        }
        [[maybe_unused]] f32x4 aux[JG][NI];
        [[maybe_unused]] bool keep[JG];
        [[maybe_unused]] float pivg[JG];
        if constexpr (EPI == EPI_GATE_T) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) keep[jj] = p.rowmask ? p.rowmask[mrow[jj]] != 0 : true;
            if (p.gate && p.gate_bstride != 0) {  // per-sample time (DiT.forward with a time vector): gate row depends on the token's batch
#pragma unroll
                for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        aux[jj][i] = *reinterpret_cast<const f32x4*>(p.gate + (size_t)((mrow[jj] + p.row0) / p.rows_per_batch) * p.gate_bstride + ncol[i]);
            }
        } else if constexpr (EPI == EPI_ROPE_T) {
            if (rope_wave) {
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
                    const int pos = (mrow[jj] + p.row0) % p.rows_per_batch;
#pragma unroll
                    for (int i = 0; i < NI; ++i) aux[jj][i] = *reinterpret_cast<const f32x4*>(p.rope + (size_t)pos * 64 + 16 * i + 4 * fq);
                }
            }
        } else if constexpr (EPI == EPI_ADD2) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (p.add2_f16) {
                        const f16x4_t hv = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const _Float16*>(p.addend) + (size_t)mrow[jj] * p.ldadd + ncol[i]);
                        aux[jj][i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                    } else {
                        aux[jj][i] = *reinterpret_cast<const f32x4*>(p.addend + (size_t)mrow[jj] * p.ldadd + ncol[i]);
                    }
                }
        } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                pivg[jj] = (p.stats_out && p.stats_pivot) ? p.stats_pivot[(size_t)mrow[jj] * 2] : 0.0f;
                keep[jj] = p.rowmask ? p.rowmask[mrow[jj]] != 0 : true;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (p.add2_f16) {  // fp16 residual stream (bf16 production mode)
                        const f16x4_t hv = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const _Float16*>(p.out_f) + (size_t)mrow[jj] * p.ldof + ncol[i]);
                        aux[jj][i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                    } else {
                        aux[jj][i] = *reinterpret_cast<const f32x4*>(p.out_f + (size_t)mrow[jj] * p.ldof + ncol[i]);
                    }
                }
            }
        }
        static_for<JG>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            const size_t mr = (size_t)mrow[jj];
            f32x4 vals[NI];
            static_for<NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                f32x4 v = acc[i][j0 + jj];  // bias: see init_acc
                if constexpr (LNF) v = lnf_apply(v, i, j0 + jj);
                if constexpr (EPI == EPI_STORE_T || EPI == EPI_STORE_F32 || EPI == EPI_GATE_T || EPI == EPI_RESID) {
                    if (p.act == ACT_GELU_TANH) {
                        v = epi_gelu_tanh4(v);
                    } else if (p.act == ACT_MISH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_mish(v[e]);
                    } else if (p.act == ACT_GELU_ERF) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_gelu_erf(v[e]);
                    }
                }
                if constexpr (EPI == EPI_GATE_T) {
                    if (p.gate) v *= (p.gate_bstride != 0 ? aux[jj][i] : gate4[i]);
                    if (!keep[jj]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if constexpr (EPI == EPI_ROPE_T) {
                    if (rope_wave) v = epi_rope4(v, aux[jj][i]);  // x_transformers apply_rotary_pos_emb: adjacent pairs, fp32 math
                } else if constexpr (EPI == EPI_ADD2) {
                    v += aux[jj][i];
                } else if constexpr (EPI == EPI_RESID) {
                    v = p.gate ? epi_axpy4(v, gate4[i], aux[jj][i]) : v + aux[jj][i];
                }
                vals[i] = v;
            });
            if constexpr (EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T) {
                bf16_t* orow = reinterpret_cast<bf16_t*>(p.out_t) + mr * p.ldo;
                if (wide_ok) {  // all four feature tiles of this wave in range and 16-byte alignable: pairwise 16-byte stores
                    static_for<NI / 2>([&](auto hc) {
                        constexpr int i = decltype(hc)::value * 2;
                        const u32x4 q = pair_swap(to_bf16x4(vals[i]), to_bf16x4(vals[i + 1]));
                        if (okm[jj]) *reinterpret_cast<u32x4*>(orow + nwide + 32 * (i / 2)) = q;
                    });
                } else {
                    static_for<NI>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if (okm[jj] && okn[i]) *reinterpret_cast<bf16x4*>(orow + ncol[i]) = to_bf16x4(vals[i]);
                    });
                }
            } else {
                static_for<NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const bool ok = okm[jj] && okn[i];
                    if constexpr (EPI == EPI_STORE_F32) {
                        if (ok) *reinterpret_cast<f32x4*>(p.out_f + mr * p.ldof + ncol[i]) = vals[i];
                    } else if constexpr (EPI == EPI_ADD2) {
                        if (ok) {
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out_t) + mr * p.ldo + ncol[i]) = to_bf16x4(vals[i]);
                            if (p.add2_f16) {
                                f16x4_t hv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) hv[e] = (_Float16)__builtin_amdgcn_fmed3f(vals[i][e], -65504.0f, 65504.0f);
                                *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(p.out_f) + mr * p.ldof + ncol[i]) = hv;
                            } else {
                                *reinterpret_cast<f32x4*>(p.out_f + mr * p.ldof + ncol[i]) = vals[i];
                            }
                        }
                    } else if constexpr (EPI == EPI_RESID) {
                        if (ok && keep[jj]) {
                            if (p.add2_f16) {
                                f16x4_t hv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) hv[e] = (_Float16)__builtin_amdgcn_fmed3f(vals[i][e], -65504.0f, 65504.0f);
                                *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(p.out_f) + mr * p.ldof + ncol[i]) = hv;
                            } else {
                                *reinterpret_cast<f32x4*>(p.out_f + mr * p.ldof + ncol[i]) = vals[i];
                            }
                        }
                    }
                });
                if constexpr (EPI == EPI_RESID && NI == 4) {
                    if (p.stats_out) {  // (launcher: fp16 stream, N % 64 == 0) a masked row keeps its stream values: they are what is summed
                        emit_stats((int)mr, okm[jj], pivg[jj], keep[jj] ? h_round4(vals[0]) : aux[jj][0], keep[jj] ? h_round4(vals[1]) : aux[jj][1],
                                   keep[jj] ? h_round4(vals[2]) : aux[jj][2], keep[jj] ? h_round4(vals[3]) : aux[jj][3]);
                    }
                }
            }
        });
    });
    };

    // ---- lean epilogue for whole tiles (the hot case): no bounds logic, one base pointer, packed fp32 math, software-pipelined
    //      RoPE table loads, row mask from one preloaded byte.  The generic epilogue below costs ~5x the instructions.
    [[maybe_unused]] auto lean_epilogue = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        bf16_t* orow = reinterpret_cast<bf16_t*>(p.out_t) + (size_t)(m0 + wm * WM + fr) * p.ldo + nwide;
        const size_t jstride = (size_t)16 * p.ldo;
        [[maybe_unused]] f32x4 rp[2][NI];
        [[maybe_unused]] int pos0 = 0;
        auto load_rope = [&](auto jc, f32x4 (&dst)[NI]) {
            constexpr int j = decltype(jc)::value;
            int pos = pos0 + 16 * j;
            pos -= pos >= p.rows_per_batch ? p.rows_per_batch : 0;  // rows_per_batch >= WM (checked by the caller of the lean path)
            const float* t = p.rope + (size_t)pos * 64 + 4 * fq;
#pragma unroll
            for (int i = 0; i < NI; ++i) dst[i] = *reinterpret_cast<const f32x4*>(t + 16 * i);
        };
        if constexpr (EPI == EPI_ROPE_T) {
            pos0 = (p.row0 + m0 + wm * WM + fr) % p.rows_per_batch;
            if (rope_wave) load_rope(std::integral_constant<int, 0>{}, rp[0]);
        }
        static_for<MI>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (EPI == EPI_ROPE_T && j + 1 < MI) {
                if (rope_wave) load_rope(std::integral_constant<int, j + 1>{}, rp[(j + 1) & 1]);
            }
            f32x4 vals[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x4 v = acc[i][j];
                if constexpr (LNF) v = lnf_apply(v, i, j);
                if constexpr (ACT == ACT_GELU_TANH) {
                    v = epi_gelu_tanh4(v);
                } else if constexpr (ACT == ACT_MISH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_mish(v[e]);
                }
                if constexpr (EPI == EPI_GATE_T) v = v * gate4[i];
                if constexpr (EPI == EPI_ROPE_T) {
                    if (rope_wave) v = epi_rope4(v, rp[j & 1][i]);
                }
                vals[i] = v;
            }
            u32x4 q0 = pair_swap(to_bf16x4(vals[0]), to_bf16x4(vals[1]));
            u32x4 q1 = pair_swap(to_bf16x4(vals[2]), to_bf16x4(vals[3]));
            if constexpr (EPI == EPI_GATE_T) {
                const bool keep = (keepbits >> j) & 1u;
                q0 = keep ? q0 : u32x4{0u, 0u, 0u, 0u};
                q1 = keep ? q1 : u32x4{0u, 0u, 0u, 0u};
            }
            bf16_t* o = orow + j * jstride;
            *reinterpret_cast<u32x4*>(o) = q0;
            *reinterpret_cast<u32x4*>(o + 32) = q1;
        });
    };
    // ---- lean form of EPI_ADD2 with the fp16 addend / fp16 stream (the input embedding of the bf16 mode, a purely HBM-bound launch):
    //      16-byte accesses throughout.  The addend is loaded at the position a lane STORES (8 consecutive features of one of two adjacent
    //      16-feature tiles) and brought to the accumulator layout by the inverse of pair_swap; both outputs leave through pair_swap.
    [[maybe_unused]] auto lean_add2_f16 = [&]() __attribute__((always_inline)) {
        const size_t row0 = (size_t)(m0 + wm * WM + fr);
        const _Float16* arow = reinterpret_cast<const _Float16*>(p.addend) + row0 * p.ldadd + nwide;
        _Float16* hrow = reinterpret_cast<_Float16*>(p.out_f) + row0 * p.ldof + nwide;
        bf16_t* orow = reinterpret_cast<bf16_t*>(p.out_t) + row0 * p.ldo + nwide;
        u32x4 adA[2], adB[2];  // addends of the even / odd token tile in flight, per feature tile pair (named apart: no runtime-indexed arrays)
        auto load_add = [&](auto jc, u32x4 (&dst)[2]) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            dst[0] = *reinterpret_cast<const u32x4*>(arow + (size_t)16 * j * p.ldadd);
            dst[1] = *reinterpret_cast<const u32x4*>(arow + (size_t)16 * j * p.ldadd + 32);
        };
        auto widen_h = [](unsigned lo, unsigned hi) __attribute__((always_inline)) {
            const f16x4_t h = __builtin_bit_cast(f16x4_t, u32x2{lo, hi});
            return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        };
        auto to_h = [](const f32x4& v) __attribute__((always_inline)) {
            f16x4_t h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
            return __builtin_bit_cast(bf16x4, h);  // 8 bytes through pair_swap's bit shuffle
        };
        load_add(std::integral_constant<int, 0>{}, adA);
        static_for<MI>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < MI) {
                if constexpr ((j + 1) & 1)
                    load_add(std::integral_constant<int, j + 1>{}, adB);
                else
                    load_add(std::integral_constant<int, j + 1>{}, adA);
            }
            f32x4 v0, v1, v2, v3;
            {
                const u32x4 q = (j & 1) ? adB[0] : adA[0];
                // inverse of pair_swap: {q0, q1} / {q2, q3} are the two halves of this lane's 8 stored features
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                v0 = acc[0][j] + widen_h(s0[0], s1[0]);
                v1 = acc[1][j] + widen_h(s0[1], s1[1]);
            }
            {
                const u32x4 q = (j & 1) ? adB[1] : adA[1];
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                v2 = acc[2][j] + widen_h(s0[0], s1[0]);
                v3 = acc[3][j] + widen_h(s0[1], s1[1]);
            }
            const size_t jo = (size_t)16 * j;
            *reinterpret_cast<u32x4*>(orow + jo * p.ldo) = pair_swap(to_bf16x4(v0), to_bf16x4(v1));
            *reinterpret_cast<u32x4*>(orow + jo * p.ldo + 32) = pair_swap(to_bf16x4(v2), to_bf16x4(v3));
            *reinterpret_cast<u32x4*>(hrow + jo * p.ldof) = pair_swap(to_h(v0), to_h(v1));
            *reinterpret_cast<u32x4*>(hrow + jo * p.ldof + 32) = pair_swap(to_h(v2), to_h(v3));
        });
    };
    // ---- lean form of EPI_RESID on the fp16 residual stream (attention out-projection and second FF linear of the bf16 mode): the stream tile
    //      is read at the position a lane stores, brought to the accumulator layout by the inverse of pair_swap, x += gate * acc (the bias is in
    //      the accumulator's start value), written back through pair_swap: 16-byte accesses throughout.  Masked query rows keep their value.
    [[maybe_unused]] auto lean_resid_f16 = [&]() __attribute__((always_inline)) {
        const size_t row0 = (size_t)(m0 + wm * WM + fr);
        _Float16* hrow = reinterpret_cast<_Float16*>(p.out_f) + row0 * p.ldof + nwide;
        u32x4 adA[2], adB[2];  // stream values of the even / odd token tile in flight (named apart: no runtime-indexed arrays)
        auto load_x = [&](auto jc, u32x4 (&dst)[2]) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            dst[0] = *reinterpret_cast<const u32x4*>(hrow + (size_t)16 * j * p.ldof);
            dst[1] = *reinterpret_cast<const u32x4*>(hrow + (size_t)16 * j * p.ldof + 32);
        };
        auto widen_h = [](unsigned lo, unsigned hi) __attribute__((always_inline)) {
            const f16x4_t h = __builtin_bit_cast(f16x4_t, u32x2{lo, hi});
            return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        };
        auto to_h = [](const f32x4& v) __attribute__((always_inline)) {
            f16x4_t h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
            return __builtin_bit_cast(bf16x4, h);  // 8 bytes through pair_swap's bit shuffle
        };
        load_x(std::integral_constant<int, 0>{}, adA);
        [[maybe_unused]] float piv[MI];  // LayerNorm fold: the rows' pivots, requested before the first stream tile is waited for
        if (p.stats_out) {
#pragma unroll
            for (int j = 0; j < MI; ++j) piv[j] = p.stats_pivot ? p.stats_pivot[(row0 + (size_t)16 * j) * 2] : 0.0f;
        }
        static_for<MI>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < MI) {
                if constexpr ((j + 1) & 1)
                    load_x(std::integral_constant<int, j + 1>{}, adB);
                else
                    load_x(std::integral_constant<int, j + 1>{}, adA);
            }
            const bool keep = (keepbits >> j) & 1u;
            f32x4 x0, x1, x2, x3;
            {
                const u32x4 q = (j & 1) ? adB[0] : adA[0];
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                x0 = widen_h(s0[0], s1[0]);
                x1 = widen_h(s0[1], s1[1]);
            }
            {
                const u32x4 q = (j & 1) ? adB[1] : adA[1];
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                x2 = widen_h(s0[0], s1[0]);
                x3 = widen_h(s0[1], s1[1]);
            }
            const f32x4 v0 = keep ? epi_axpy4(acc[0][j], gate4[0], x0) : x0;
            const f32x4 v1 = keep ? epi_axpy4(acc[1][j], gate4[1], x1) : x1;
            const f32x4 v2 = keep ? epi_axpy4(acc[2][j], gate4[2], x2) : x2;
            const f32x4 v3 = keep ? epi_axpy4(acc[3][j], gate4[3], x3) : x3;
            const size_t jo = (size_t)16 * j;
            *reinterpret_cast<u32x4*>(hrow + jo * p.ldof) = pair_swap(to_h(v0), to_h(v1));
            *reinterpret_cast<u32x4*>(hrow + jo * p.ldof + 32) = pair_swap(to_h(v2), to_h(v3));
            if (p.stats_out) {  // LayerNorm fold: partial statistics of the rounded values (x0..x3 of a masked row are already fp16 values)
                emit_stats((int)row0 + 16 * j, true, piv[j], h_round4(v0), h_round4(v1), h_round4(v2), h_round4(v3));
            }
        });
    };
    // (Tried and dropped, round 3, in-situ A/B on one box, gpurun_out/r3o_*: requesting the tile's first stream values three K-steps before
    //  the tile ends -- no effect, 252 registers; fetching the next tile's gate vectors / row-mask byte ahead of this tile's stores, so that
    //  the compiler's vmcnt(0) in front of their first use does not sit behind the stores -- out-projection 171 -> 187 us, FF2 293 -> 308 us.)
    // whole tile + the operand forms the lean epilogue assumes; anything else takes the generic path
    [[maybe_unused]] auto lean_ok = [&]() {
        if constexpr (!(EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T)) return false;
        bool ok = m0 + BM <= p.M && n0 + BN <= p.N && (p.ldo & 7) == 0 && (LNF || p.bias != nullptr) && p.act != ACT_GELU_ERF;
        if constexpr (EPI == EPI_GATE_T) ok = ok && p.gate_bstride == 0 && (!p.rowmask || p.rowbits);
        if constexpr (EPI == EPI_ROPE_T) ok = ok && p.rows_per_batch >= WM;
        return ok;
    };
    auto epilogue = [&]() {
        if constexpr (LNF_LDS && !P30) {  // the ring is dead once every wave has retired its last fragment reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage_lnf(m0, n0, smem + wave * 2048);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (P30 && EPI == EPI_RESID) {  // (the launcher guarantees whole tiles, the fp16 stream and lean operand forms)
            lean_resid_f16();
            return;
        } else if constexpr (P30) {  // the launcher guarantees whole tiles and lean operand forms: no generic code in this build
            if (p.act == ACT_GELU_TANH)
                lean_epilogue(std::integral_constant<int, ACT_GELU_TANH>{});
            else
                lean_epilogue(std::integral_constant<int, ACT_NONE>{});
            return;
        } else if constexpr (EPI == EPI_ADD2 && NI == 4) {
            if (p.lean_epi && p.add2_f16 && p.act == ACT_NONE && m0 + BM <= p.M && n0 + BN <= p.N && ((p.ldo | p.ldof | p.ldadd) & 7) == 0) {
                lean_add2_f16();
                return;
            }
        } else if constexpr (EPI == EPI_RESID && NI == 4) {
            if (p.lean_epi && p.add2_f16 && p.act == ACT_NONE && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldof & 7) == 0 && p.bias != nullptr &&
                (!p.gate || p.gate_bstride == 0) && (!p.rowmask || p.rowbits)) {
                lean_resid_f16();
                return;
            }
        } else if constexpr (EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T) {
            if (p.lean_epi && lean_ok()) {
                if (p.act == ACT_GELU_TANH)
                    lean_epilogue(std::integral_constant<int, ACT_GELU_TANH>{});
                else if (p.act == ACT_MISH)
                    lean_epilogue(std::integral_constant<int, ACT_MISH>{});
                else
                    lean_epilogue(std::integral_constant<int, ACT_NONE>{});
                return;
            }
        }
        generic_epilogue();
    };

    if constexpr (PERSIST) {
        // ---- persistent staggered ring: the ring never drains between tiles.  Global stage g = (tile ordinal) * nk + kt lives in
        //      slot g % NSTAGE; the DMA front runs D stages ahead and crosses into the next tile's operands, so a tile's first
        //      K-steps are already in LDS when the previous tile's epilogue ends.  The two wave groups re-synchronise around
        //      every epilogue (both store at the same time) and re-stagger afterwards.
        constexpr int D = NSTAGE - 1;
        const int G = gridDim.x;
        const int my_tiles = (nblocks - (int)blockIdx.x + G - 1) / G;
        const int total = my_tiles * nk;
        int it_tile = blockIdx.x, it_k = 0, gi = 0, slot_i = 0;  // DMA front: tile, K-step inside it, global stage, ring slot
        auto issue_next = [&]() {
#pragma unroll
            for (int pc = 0; pc < PPW; ++pc) issue_piece(slot_i, it_k, pc);
            ++gi;
            slot_i = slot_i + 1 == NSTAGE ? 0 : slot_i + 1;
            if (++it_k == nk) {
                it_k = 0;
                it_tile += G;
                if (it_tile < nblocks) {
                    int sm, sn;
                    tile_mn(it_tile, sm, sn);
                    setup_src(sm, sn);
                }
            }
        };
        // VAR 30: the bias is folded into the accumulator start value and is not kept during the main loop (register budget: 128
        // accumulators + 48 fragments + 16 gate + addresses); the next tile's bias is fetched under the epilogue.
        auto load_bias = [&](int tn0, f32x4 (&dst)[NI]) {
#pragma unroll
            for (int i = 0; i < NI; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.bias + tn0 + wn * WN + i * 16 + 4 * fq);
        };
        auto acc_from = [&](const f32x4 (&b4)[NI]) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = b4[i];
        };
        [[maybe_unused]] f32x4 b0[NI];
        if constexpr (P30 && LNF) {
#pragma unroll
            for (int i = 0; i < NI; ++i) b0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if constexpr (P30) {
            load_bias(n0, b0);
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (gi < total) issue_next();
        if constexpr (P30)
            acc_from(b0);
        else
            init_acc();
        wait_pieces(min(D - 1, total - 1));
        __builtin_amdgcn_s_barrier();
        const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
        int g = 0, slot_c = 0;
        bf16x8 wf[NI], af[MI];
        int nm0 = m0, nn0 = n0;
        for (int t = 0; t < my_tiles; ++t) {
            if (t > 0) {
                if constexpr (P30) {
                    m0 = nm0;
                    n0 = nn0;
                } else {
                    tile_mn(blockIdx.x + t * G, m0, n0);
                }
                prep_epilogue();
                if constexpr (!P30) init_acc();
            }
            if (late) __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                if (gi < total) issue_next();
                read_frags(slot_c, wf, af);
                // my pieces of global stage g+1 must have landed before the barrier that precedes anybody's next P phase.
                // P30, first three K-steps of a tile after the first: the four stages fetched ahead landed BEFORE the previous epilogue
                // (explicit vmcnt(0) there), so no wait is needed -- and none may be issued: vmcnt retires in order and the epilogue's
                // stores sit in front of the DMA issued since, a wait here would hold the wave until all its stores are acknowledged.
                if (!(P30 && t > 0 && kt < 3)) wait_pieces(g + 1 < total ? (gi - 1) - (g + 1) : 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                mma(wf, af);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                ++g;
                slot_c = slot_c + 1 == NSTAGE ? 0 : slot_c + 1;
            }
            if constexpr (P30) {
                // Next tile's bias (after the last tile: the same tile's again, unused) -- unconditional code on purpose: with the
                // loads, their use and the accumulator restart under `if (more tiles)`, the compiler's vmcnt bookkeeping merged the two
                // paths into a vmcnt(0) BEHIND the epilogue's stores (loads and stores share the counter on this ISA), i.e. every tile
                // waited for its own 128 KiB of stores to be acknowledged before the next main loop could start.
                f32x4 bn[NI];
                if (t + 1 < my_tiles) tile_mn(blockIdx.x + (t + 1) * G, nm0, nn0);
                if constexpr (LNF) {  // no bias: THIS tile's column constants and row statistics travel across the barrier instead
                    stage_lnf(m0, n0, smem + slot_i * STAGE + wave * APW * 1024);  // the DMA front's next slot holds no live stage
#pragma unroll
                    for (int i = 0; i < NI; ++i) bn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    load_bias(nn0, bn);  // in flight across the re-synchronising barrier
                }
                if (!late) __builtin_amdgcn_s_barrier();  // same barrier count for both groups; both now store together
#pragma unroll
                for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(bn[i]));  // the wait for the bias sits here: only loads are in flight
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and every K-step fetched ahead has landed (see the main loop)
                __builtin_amdgcn_sched_barrier(0);
                epilogue();
                acc_from(bn);
            } else {
                if (!late) __builtin_amdgcn_s_barrier();
                epilogue();
            }
        }
        clk_end();
        return;
    }
    constexpr int D = NSTAGE - 1;  // K-steps issued ahead of the one being consumed
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nk) issue(d);
    lnf_inkernel_stats();
    init_acc();
    wait_pieces(min(D - 1, nk - 1));  // K-step 0 has landed for this wave
    __builtin_amdgcn_s_barrier();     // ... and for every wave

    if constexpr (VAR == 0) {
        // ---- plain ring: every wave does {refill, fragment reads, MFMAs} per K-step, one barrier per K-step
        bf16x8 wf[NI], af[MI];
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + D < nk) issue(kt + D);  // refills slot (kt-1) % NSTAGE: its readers passed the previous barrier
            read_frags(kt, wf, af);
            mma(wf, af);
            if (kt + 1 < nk) {
                wait_pieces(min(kt + D, nk - 1) - (kt + 1));
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        // ---- ring + staggered wave groups (the two waves that share a SIMD never run the same phase together):
        //   every wave alternates  P_k: {refill slot (k-1) % NSTAGE by DMA, ds_read the fragments of K-step k, counted vmcnt
        //   for K-step k+1}  |barrier|  C_k: {32 MFMAs}  |barrier| ...; waves 4-7 run one barrier interval behind waves 0-3,
        //   so in every interval one group feeds the matrix pipe while its SIMD partners do their LDS/DMA work.
        const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
        if (late) __builtin_amdgcn_s_barrier();
        bf16x8 wf[NI], af[MI];
        for (int kt = 0; kt < nk; ++kt) {
            // -- P_kt.  Slot (kt-1) % NSTAGE is free: both groups retired their reads of it (lgkmcnt(0) below) at least one barrier ago.
            if (kt + D < nk) issue(kt + D);
            read_frags(kt, wf, af);
            // my pieces of K-step kt+1 must have landed before the barrier that precedes anybody's P_{kt+1}
            wait_pieces(kt + 1 < nk ? min(kt + D, nk - 1) - (kt + 1) : 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // -- C_kt
            __builtin_amdgcn_s_setprio(1);
            mma(wf, af);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (!late) __builtin_amdgcn_s_barrier();  // same barrier count for both groups
    }

    epilogue();
    if constexpr (EPI == EPI_RESID && !PERSIST && !LNF) {
        if (p.fin_counter) {
            // LayerNorm fold: finish the row statistics of this block of token rows in the launch (gemm.h: fin_counter).  Every wave has stored
            // its partial sums write-through; drain them, meet, draw ONE ticket per workgroup; the workgroup that draws the last one owns the block.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(smem);  // (the operand ring is dead: every wave is past its last fragment read)
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.fin_counter + m0 / BM, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag == (unsigned)(tiles_n - 1)) {
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(p.fin_counter + m0 / BM, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                }
                __syncthreads();
                bool bad = false;
                float sumsq = 0.f;
                const int m = m0 + tid;
                if (tid < BM && m < p.M) {
                    const int ncols = p.N >> 6;
                    float s1 = 0.f, s2 = 0.f;
                    if (ncols == 16) {
                        float v1[16], v2[16];
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            const float* src = p.stats_out + ((size_t)c * p.stats_ld + m) * 2;
                            v1[c] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v2[c] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            s1 += v1[c];
                            s2 += v2[c];
                        }
                    } else {
                        for (int c = 0; c < ncols; ++c) {
                            const float* src = p.stats_out + ((size_t)c * p.stats_ld + m) * 2;
                            s1 += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s2 += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    const float pv = p.stats_pivot ? p.stats_pivot[(size_t)m * 2] : 0.0f;
                    float mean, rstd;
                    lnf_row_stats(s1, s2, pv, p.N, mean, rstd, sumsq);
                    *reinterpret_cast<f32x2*>(p.fin_stats + (size_t)m * 2) = f32x2{mean, rstd};
                    bad = !(sumsq < 65504.0f * 65504.0f);
                }
                lnf_raise_guard(p.lnf_sat, bad, sumsq, p.lnf_sat_tag, m + p.row0);
            }
        }
    }
    clk_end();
}

#ifdef F5_LNF_TU
extern unsigned long long* g_gemm_clk_buf;
extern int g_gemm_variant, g_gemm_group, g_gemm_persist_grid, g_gemm_persist, g_gemm_reverse_sites, g_gemm_group_sites, g_gemm_tile, g_gemm_bm128, g_gemm_lean,
    g_gemm_split_tail;
int gemm_persist_grid();
static int persist_grid() { return gemm_persist_grid(); }
#else
unsigned long long* g_gemm_clk_buf = nullptr;  // diagnostic (f5_debug_gemm_clock)
int g_gemm_variant = 1;  // tuning knob (f5_tuning_set("gemm_variant", v)): 0 = plain ring everywhere, 1 = staggered wave groups (+ persistent grid)
int g_gemm_group = 0;    // tuning knob ("gemm_group"): token tiles per L2 patch (0 = by shape, 1 = feature-tile-fastest order)
int g_gemm_persist_grid = 0;  // tuning knob ("gemm_persist_grid"): workgroups of the persistent kernel (0 = one per CU of the device)
int g_gemm_persist = 1;       // tuning knob ("gemm_persist"): 1 = whole-tile block linears run on the persistent grid
int gemm_persist_grid() {
    if (g_gemm_persist_grid > 0) return g_gemm_persist_grid;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n & ~7;  // a multiple of the 8 XCDs keeps a workgroup's tiles on one XCD
        if (cus < 8) cus = 8;
    }
    return cus;
}
// tuning knob ("gemm_reverse_sites"): bit mask of block call sites (bit 0 qkv, 1 out, 2 ff1, 3 ff2) that walk their tiles backwards.  Default 8:
// FF2 reads FF1's 256 MiB output -- more than the 256 MB Infinity Cache holds -- so walking it in FF1's order finds the oldest lines evicted
// all the way, the opposite order meets the newest half while it is still cached: FF2 272.5 -> 267.3 us in situ, C2 30 185 -> 30 328
// mel-frames/s (same-box A/B; reversing FF1 instead does the same, both together nothing, the other sites nothing).  Same values either way.
int g_gemm_reverse_sites = 8;
int g_gemm_group_sites = 0;  // diagnostic knob ("gemm_group_sites"): patch height per block call site, two decimal digits each: qkv|out|ff1|ff2
int g_gemm_tile = 0;   // diagnostic knob ("gemm_tile"): bm * 1000 + bn forces the tile of every tuned-GEMM launch that supports it (0 = by shape)
int g_gemm_bm128 = 1;  // tuning knob ("gemm_bm128"): 128-row token tiles when the 256-row ones leave CUs without a workgroup (single-utterance launches)
int g_gemm_lean = 1;   // tuning knob ("gemm_lean"): 1 = lean epilogue on whole tiles, 0 = generic epilogue everywhere
int g_gemm_split_tail = 0;  // tuning knob ("gemm_split_tail"): 1 = M % 256 != 0 launches run as persistent whole tiles + a tail launch.  Measured a LOSS (the tail launch is pure latency: 8 x 1001 346 vs 288 ms); off.  dit_eval rounds the rows up instead ("gemm_pad_rows")

static int persist_grid() { return gemm_persist_grid(); }
#endif

// tile walk of a launch: L2 patch height and direction per block call site (also used by the one-wave-per-SIMD kernel, gemm_w4.hip)
static void apply_site_knobs(GemmParams& p) {
    // L2 patch height: 8 token tiles per XCD patch.  (Until late in round 3 the 8-feature-tile projection -- FF1, N = 2048 -- took 16: +3 % on
    // that launch in isolation, but in situ the FF2 launch behind it reads FF1's output and runs 277 -> 267 us when FF1 wrote it in patches of
    // 8 like its own: same-box A/B x 3 at C2, 30 408 -> 30 575 mel-frames/s.)
    p.tile_group = g_gemm_group > 0 ? g_gemm_group : 8;
    if (g_gemm_reverse_sites > 0) {  // bit 0 qkv, 1 out-projection, 2 FF1, 3 FF2 walk their tiles backwards
        const int site = p.site - 1;
        if (site >= 0 && ((g_gemm_reverse_sites >> site) & 1)) p.tile_reverse = 1;
    }
    if (g_gemm_group_sites > 0) {  // diagnostic: per call site, decimal digits pairs qkv|out|ff1|ff2 (e.g. 8160804)
        const int site = p.site - 1;
        static const int div[4] = {1000000, 10000, 100, 1};
        if (site >= 0) {
            const int v = (g_gemm_group_sites / div[site]) % 100;
            if (v > 0) p.tile_group = v;
        }
    }
}

template <int BN, int WM, int MODE, int EPI, bool LNF = false> static int launch_fast(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    apply_site_knobs(p);
    p.lean_epi = g_gemm_lean;
    p.clk = g_gemm_clk_buf;
    constexpr int BMv = (8 / (BN / 64)) * WM;  // token rows of a tile (see the kernel)
    const int tiles_m = cdiv(p.M, BMv), tiles_n = cdiv(p.N, BN);
    const int nblocks = tiles_m * tiles_n;
    dim3 grid(nblocks), block(512);
    // persistent grid (one workgroup per CU walking tiles, next tile's first K-steps prefetched under the epilogue, bias folded into the
    // accumulator start): whole tiles with the lean epilogue's operand forms only
    // (K >= 128: the last K-step of a tile must be one that still waits for the next tile's first stage, see the main loop)
    const bool store_ok = (EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T) && (p.ldo & 7) == 0 && p.out_t && (p.act == ACT_NONE || p.act == ACT_GELU_TANH) &&
                          (EPI != EPI_GATE_T || (p.gate_bstride == 0 && (!p.rowmask || p.rowbits))) && (EPI != EPI_ROPE_T || p.rows_per_batch >= 128);
    const bool resid_ok = EPI == EPI_RESID && p.add2_f16 && p.out_f && (p.ldof & 7) == 0 && p.act == ACT_NONE && (!p.gate || p.gate_bstride == 0) &&
                          (!p.rowmask || p.rowbits);  // in-place update of the fp16 residual stream
    const bool persist_ok = MODE == GEMM_DENSE && (store_ok || resid_ok) && p.M % 256 == 0 && p.K >= 128 && p.N % 256 == 0 && (p.bias || LNF);
    if constexpr (BN == 256 && MODE == GEMM_DENSE && (EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T || EPI == EPI_RESID)) {
        // A token count that is not a multiple of the tile height (8 x 1001 frames): the whole tiles go to the persistent schedule like any
        // other launch, the < 256 tail rows to a launch of their own on 128-wide tiles (generic epilogue).  Every accumulator sums K in the
        // same order whatever the tile and the lean and generic epilogues round alike, so the values do not depend on the split.  (Round 3
        // sent the WHOLE launch to the non-persistent schedule, whose QKV + RoPE build spilled: VERDICT round 3, item 3.)
        if (g_gemm_variant != 0 && g_gemm_persist && g_gemm_split_tail && !persist_ok && p.M > 256 && p.M % 256 != 0 && p.a_row_mod == 0 && p0.row0 == 0 &&
            MODE == GEMM_DENSE && (store_ok || resid_ok) && p.K >= 128 && p.N % 256 == 0 && (p.bias || LNF)) {
            const int mw = p.M / 256 * 256;
            GemmParams pw = p0, pt = p0;
            pw.M = mw;
            pt.M = p.M - mw;
            pt.row0 = mw;
            pt.A = static_cast<const char*>(p.A) + (size_t)mw * p.lda * 2;
            if (p.out_t) pt.out_t = static_cast<char*>(p.out_t) + (size_t)mw * p.ldo * 2;
            if (p.out_f) pt.out_f = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out_f) + (size_t)mw * p.ldof * (p.add2_f16 ? 2 : 4));
            if (p.rowmask) pt.rowmask = p.rowmask + mw;
            if (p.rowbits) pt.rowbits = p.rowbits + (size_t)(mw >> 7) * 16;
            if (p.lnf_stats) pt.lnf_stats = p.lnf_stats + (size_t)mw * 2;
            if (p.stats_out) pt.stats_out = p.stats_out + (size_t)mw * 2;
            if (p.stats_pivot) pt.stats_pivot = p.stats_pivot + (size_t)mw * 2;
            if (p.lnf_partial) pt.lnf_partial = p.lnf_partial + (size_t)mw * 2;
            if (p.lnf_pivot) pt.lnf_pivot = p.lnf_pivot + (size_t)mw * 2;
            if (p.lnf_stats_out) pt.lnf_stats_out = p.lnf_stats_out + (size_t)mw * 2;
            pw.pf_n[0] = pw.pf_n[1] = 0u;
            F5_TRY((launch_fast<BN, WM, MODE, EPI, LNF>(pw, stream)));
            return launch_fast<128, 64, MODE, EPI, LNF>(pt, stream);
        }
    }
    if constexpr (BN == 256) {
        if (g_gemm_variant == 0)
            hipLaunchKernelGGL((gemm_fast_kernel<BN, WM, MODE, EPI, 0, 4, LNF>), grid, block, 0, stream, p, tiles_n, nblocks);
        else if (g_gemm_persist && persist_ok) {
            if constexpr (EPI == EPI_STORE_T || EPI == EPI_GATE_T || EPI == EPI_ROPE_T || EPI == EPI_RESID)
                hipLaunchKernelGGL((gemm_fast_kernel<BN, WM, GEMM_DENSE, EPI, 30, 5, LNF>), dim3(nblocks < persist_grid() ? nblocks : persist_grid()), block,
                                   0, stream, p, tiles_n, nblocks);
        } else
            hipLaunchKernelGGL((gemm_fast_kernel<BN, WM, MODE, EPI, 1, 5, LNF>), grid, block, 0, stream, p, tiles_n, nblocks);
    } else if (BN == 128 && g_gemm_variant != 0) {  // staggered wave groups pay on the 128-wide tile too (M = 8192: FF2 44.6 -> 41.5 us, out-projection 26.3 -> 24.4 us)
        hipLaunchKernelGGL((gemm_fast_kernel<BN, WM, MODE, EPI, 1, 5, LNF>), grid, block, 0, stream, p, tiles_n, nblocks);
    } else {
        hipLaunchKernelGGL((gemm_fast_kernel<BN, WM, MODE, EPI, 0, 4, LNF>), grid, block, 0, stream, p, tiles_n, nblocks);
    }
    F5_LAUNCH_CHECK();
    return 0;
}

#ifndef F5_LNF_TU
int launch_gemm_fast_lnf(const GemmParams& p, int epi, int bm, int bn, hipStream_t stream);  // gemm_fast_lnf.hip
bool gemm_w4_ok(const GemmParams& p, int mode, int epi);                                     // gemm_w4.hip
int launch_gemm_w4(const GemmParams& p, int epi, hipStream_t stream);
bool gemm_fast_supported(const GemmParams& p, int precision, int mode, int epi) {
    if (precision != F5_PREC_BF16 || p.M <= 0 || p.N <= 0) return false;
    if (p.lda & 7) return false;
    if (mode == GEMM_DENSE) {
        if (p.K <= 0 || p.K % 32 != 0 || (p.ldw & 7)) return false;
        if (p.N % 4 != 0) return false;
        if (epi == EPI_ROPE_T && (p.rope_inner % 64 != 0 || p.N % 64 != 0)) return false;
        if (epi == EPI_RESID && p.gate && p.gate_bstride != 0) return false;
        return epi >= EPI_STORE_T && epi <= EPI_GATE_T;
    }
    if (mode == GEMM_CONV31) {
        if (p.conv_win <= 0 || p.conv_win % 32 != 0 || p.N % 64 != 0 || p.conv_cg <= 0 || (p.conv_cg & 7)) return false;
        if (p.rows_per_batch <= 0 || p.M % p.rows_per_batch != 0) return false;
        return epi == EPI_STORE_T || epi == EPI_GATE_T;
    }
    return false;
}

void gemm_fast_tile(const GemmParams& p, int* pbm, int* pbn) {
    // tile width by occupancy: the 256-wide tile is the efficient one, narrower tiles keep the 256 CUs busy when the token count
    // is small (single-utterance serving: M = 2 x frames)
    const int tiles_m = cdiv(p.M, 256);
    int bn = 256, bm = 256;
    // (a 128-wide tiling of more workgroups than CUs would run in two rounds: the 256-wide one, with at least 129 tiles then, wins --
    //  M = 3072 QKV: 144 tiles of 256 x 256 33.3 us against 288 of 256 x 128 42.0 us)
    if (p.N % 256 != 0 || (tiles_m * (p.N / 256) < 160 && tiles_m * cdiv(p.N, 128) <= persist_grid())) bn = 128;
    if (bn == 128 && p.N % 64 == 0 && tiles_m * cdiv(p.N, 128) < 160) bn = 64;
    // Small launches (single-utterance serving: M = 2 x frames): 128-row token tiles.  (a) the 256 x 64 tiling still leaves CUs without a
    // workgroup: 128 x 64 doubles the count (M = 2048: out-projection 20.8 -> 15.8 us, FF2 33.6 -> 24.6 us); (b) 256 x 64 fills the CUs but
    // 128 x 128 does too with a fifth fewer operand rows per K-step (M = 4096: out-projection 22.2 -> 20.7 us, FF2 35.4 -> 32.5 us; M = 2048
    // FF1 20.4 -> 19.4 us).  Every accumulator sums K in the same order whatever the tile, so the result bits do not change.
    // Knob "gemm_bm128": 0 never, 1 these two rules, 2 128-row tiles everywhere (tests).
    if (g_gemm_bm128 == 2) {
        bm = 128;
        bn = (p.N % 128 == 0 && cdiv(p.M, 128) * (p.N / 128) >= 320) ? 128 : 64;
        if (p.N % 64 != 0) bn = 128;
    } else if (g_gemm_bm128 == 1 && bn == 64 && tiles_m * cdiv(p.N, 64) < 160) {
        bm = 128;
    } else if (g_gemm_bm128 == 1 && bn == 64 && p.N % 128 == 0 && cdiv(p.M, 128) * (p.N / 128) >= 160) {
        bm = 128;
        bn = 128;
    }
    if (g_gemm_tile > 0) {
        const int fbm = g_gemm_tile / 1000, fbn = g_gemm_tile % 1000;
        if ((fbm == 256 || fbm == 128) && (fbn == 256 || fbn == 128 || fbn == 64) && !(fbm == 128 && fbn == 256) && p.N % fbn == 0) {
            bm = fbm;
            bn = fbn;
        }
    }
    *pbm = bm;
    *pbn = bn;
}

bool gemm_fast_resid_finishes(const GemmParams& p) {
    int bm, bn;
    gemm_fast_tile(p, &bm, &bn);
    const bool resid_ok = p.add2_f16 && p.out_f && (p.ldof & 7) == 0 && p.act == ACT_NONE && (!p.gate || p.gate_bstride == 0) && (!p.rowmask || p.rowbits);
    const bool persistent = bn == 256 && g_gemm_variant != 0 && g_gemm_persist && resid_ok && p.M % 256 == 0 && p.K >= 128 && p.N % 256 == 0 && p.bias;
    const bool split = bn == 256 && g_gemm_split_tail && p.M % 256 != 0;  // (whole tiles on the persistent schedule + a tail launch)
    return !persistent && !split && p.N % 64 == 0 && p.N / 64 <= 64;
}

bool gemm_fast_lnf_inkernel(const GemmParams& p) {
    int bm, bn;
    gemm_fast_tile(p, &bm, &bn);
    return bn != 256;  // (the 256-wide tile stages finalized statistics through LDS; every other tile holds them in registers)
}

int launch_gemm_fast(const GemmParams& p, int mode, int epi, hipStream_t stream) {
    if (mode == GEMM_CONV31) {
        if (conv31_supported(p, F5_PREC_BF16, epi)) return launch_conv31(p, stream);
        if (epi == EPI_STORE_T) return launch_fast<64, 32, GEMM_CONV31, EPI_STORE_T>(p, stream);
        if (epi == EPI_GATE_T) return launch_fast<64, 32, GEMM_CONV31, EPI_GATE_T>(p, stream);
        return f5_fail(F5_EINVAL, "gemm_fast(conv31): unsupported epilogue %d", epi);
    }
    if (g_gemm_variant != 0 && g_gemm_persist && gemm_w4_ok(p, mode, epi)) {  // large whole-tile block linears: one wave per SIMD (gemm_w4.hip)
        GemmParams q = p;
        apply_site_knobs(q);
        return launch_gemm_w4(q, epi, stream);
    }
    int bn, bm;
    gemm_fast_tile(p, &bm, &bn);
    if (p.lnf_stats) {  // LayerNorm fold: fp16 operands, statistics + column constants in the epilogue (QKV + RoPE, FF1 + GELU)
        if (!p.lnf_c1 || !p.lnf_c2 || (epi != EPI_STORE_T && epi != EPI_ROPE_T)) return f5_fail(F5_EINVAL, "gemm_fast: LayerNorm fold needs c1, c2 and a store / RoPE epilogue");
        return launch_gemm_fast_lnf(p, epi, bm, bn, stream);
    }
    if (p.fin_counter && (!p.stats_out || !p.fin_stats || epi != EPI_RESID || !gemm_fast_resid_finishes(p)))
        return f5_fail(F5_ESTATE, "gemm_fast: statistics can be finished inside the launch by the non-persistent in-place residual schedules only");
    if (p.stats_out && (epi != EPI_RESID || !p.add2_f16 || p.N % 64 != 0 || p.stats_ld < p.M))
        return f5_fail(F5_EINVAL, "gemm_fast: row statistics need the in-place fp16 residual epilogue and N % 64 == 0");
#define F5_FAST_CASE(E)                                                                   \
    case E:                                                                               \
        if (bm == 128 && bn == 128) return launch_fast<128, 32, GEMM_DENSE, E>(p, stream); \
        if (bm == 128) return launch_fast<64, 16, GEMM_DENSE, E>(p, stream);              \
        if (bn == 256) return launch_fast<256, 128, GEMM_DENSE, E>(p, stream);            \
        if (bn == 128) return launch_fast<128, 64, GEMM_DENSE, E>(p, stream);             \
        return launch_fast<64, 32, GEMM_DENSE, E>(p, stream);
    switch (epi) {
        F5_FAST_CASE(EPI_STORE_T)
        F5_FAST_CASE(EPI_STORE_F32)
        F5_FAST_CASE(EPI_RESID)
        F5_FAST_CASE(EPI_ADD2)
        F5_FAST_CASE(EPI_ROPE_T)
        F5_FAST_CASE(EPI_GATE_T)
    }
#undef F5_FAST_CASE
    return f5_fail(F5_EINVAL, "gemm_fast: unsupported epilogue %d", epi);
}
#else  // F5_LNF_TU
int launch_gemm_fast_lnf(const GemmParams& p, int epi, int bm, int bn, hipStream_t stream) {
    if (bn == 256 && p.lnf_partial) return f5_fail(F5_ESTATE, "gemm_fast: the 256-wide LayerNorm-fold tile takes finalized statistics (gemm_fast_lnf_inkernel)");
#define F5_FAST_LNF(E)                                                                           \
    if (epi == E) {                                                                              \
        if (bm == 128 && bn == 128) return launch_fast<128, 32, GEMM_DENSE, E, true>(p, stream); \
        if (bm == 128) return launch_fast<64, 16, GEMM_DENSE, E, true>(p, stream);               \
        if (bn == 256) return launch_fast<256, 128, GEMM_DENSE, E, true>(p, stream);             \
        if (bn == 128) return launch_fast<128, 64, GEMM_DENSE, E, true>(p, stream);              \
        return launch_fast<64, 32, GEMM_DENSE, E, true>(p, stream);                              \
    }
    F5_FAST_LNF(EPI_STORE_T)
    F5_FAST_LNF(EPI_ROPE_T)
#undef F5_FAST_LNF
    return f5_fail(F5_EINVAL, "gemm_fast: LayerNorm fold supports the store and RoPE epilogues");
}
#endif
