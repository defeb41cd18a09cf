// conv_split16.h -- the split-scheme Conv1d implicit GEMM of conv_split.h on the 16x16x32 MFMA shape
// (v_mfma_f32_16x16x32_{f16,bf16}) for the wide MRF / FFN layers (C_in a multiple of 32, plain convs).
//
// Why another shape.  The split kernels are POWER limited, not issue limited (with random operands a
// pure MFMA stream sustains ~1.7 PF at whatever duty cycle; the clock follows the energy per instruction).  What is left
// to win is energy per useful FLOP, and the 16x16x32 shape spends less of it than 32x32x16:
//   * one operand fragment (64 lanes x 8 halves) now spans K = 32 instead of 16: a 64x64 wave tile is 4x4 tiles fed by
//     4 + 4 fragments per 32-deep k-step -- HALF the LDS / L1 operand bytes per MAC of the 2x2 tiling of 32x32x16;
//   * a quarter of the accumulator registers are read and written per instruction (tools/probes/mfma_power.hip: +7 %
//     sustained rate for the bare instruction stream).
// Data path: CI = 32 input channels per chunk = one k-step per tap.
//   * B (activations): global -> registers -> (leaky-ReLU, scale, split) -> LDS as [piece][octet 0..3][column][8 channels]
//     (16 bytes per (column, octet), columns contiguous): a lane (column l & 15, octet l >> 4) reads its fragment at any
//     tap shift with one ds_read_b128, 16 lanes on consecutive addresses -- conflict free without swizzling; stores too.
//     Double buffered, one barrier per chunk.  One fragment set, refilled in place during the last MFMA group of a step.
//   * A (weights): host-packed [m16 tile][chunk * k + tap][piece][lane][8 halves] (lane = row l & 15, channels 8 (l >> 4)..),
//     one buffer_load_dwordx4 per (m16 tile, piece) per step, re-fetched in place right after the row tile's MFMA group.
#pragma once
#include "conv_split.h"

// plane input (XPL instantiations): 1 = global -> LDS directly (buffer_load_dwordx4 ... lds), 0 = through registers
// (16-byte loads, 16-byte ds_write): an A/B switch for experiment builds (tools/build_exp.sh)
#ifndef PARROT_XPL_DMA
#define PARROT_XPL_DMA 1
#endif

namespace parrot {

// XPL: the input comes as an operand plane (ConvParams::xplane: [piece][C / 8][T][8 x 16 bit] per batch row, written by the
// producing layer's epilogue below): the slab staging is one 16-byte load and one 16-byte LDS store per (octet, column, piece) --
// no leaky ReLU / scale / split in this kernel, and none repeated by the M-blocks and halo columns that share an input.
template <class SCH, int WAVES_M, int WAVES_N, int TM, int TN, int K, int MINW = 2, bool XPL = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_split16_kernel(const ConvParams p) {
    static_assert(K > 0, "tap count is a template parameter");
    constexpr int NPC = SCH::NP, NTERM = SCH::NT;
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    constexpr int COLS = (BN + CONV_HALO + 63) / 64 * 64;  // staged columns per chunk
    constexpr int OCT_BYTES = COLS * 16, PIECE_BYTES = 4 * OCT_BYTES, BUF_BYTES = NPC * PIECE_BYTES;
    constexpr int LIVE = COLS * 4;                   // (octet, column) items per chunk
    constexpr int ITEMS = (LIVE + NT - 1) / NT;
    constexpr int STEP_BYTES = NPC * 1024;
    static_assert(LIVE % NT == 0, "every thread stages the same number of items");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [2][NP][4 octets][COLS][8 halves]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g4 = lane >> 4, l15 = lane & 15;

    // XCD-aware block order (see conv_split.h): a contiguous run of column tiles per XCD, M-blocks back to back
    const int n_mb = (p.M + BM - 1) / BM;
    const int n_tiles = p.tiles_n * p.B, tpx = (n_tiles + 7) >> 3;  // column tiles per XCD
    const int vb = blockIdx.x;
    const int xcd = vb & 7, seq = vb >> 3;  // (workgroup id i runs on XCD i % 8, in id order)
    const int mblock = seq % n_mb;
    const int tile_id = xcd * tpx + seq / n_mb;
    if (tile_id >= n_tiles || seq / n_mb >= tpx) return;
    const int b = tile_id / p.tiles_n;
    const int tn0 = tile_id - b * p.tiles_n;
    const int t0 = tn0 * BN;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const int W = BN + (K - 1) * p.dil;
    const char* __restrict__ xb = XPL ? reinterpret_cast<const char*>(p.xplane) + (size_t)b * p.xplane_bstride
                                      : reinterpret_cast<const char*>(p.x + (size_t)b * p.x_bstride);

    // ---- slab fetch: buffer-addressed, out-of-row / out-of-reach columns read 0 (= the zero padding) ----------------
    const size_t xaddr = reinterpret_cast<size_t>(xb);
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((size_t)x_hi << 32) | x_lo), 0, 0x7fffffff, 0x00020000);
    // (plain input: a row of one channel; plane input: a row of one channel OCTET, 16 bytes per column)
    const int row_bytes = XPL ? p.Tin * 16 : p.Tin * 4;
    const int xpiece_bytes = (p.Cin >> 3) * row_bytes;  // (plane input: bytes of one piece of a batch row)
    const float slope = (p.pre == PRE_LRELU) ? p.pre_slope : 1.f;
    int voff[ITEMS], soct[ITEMS];
    static_for<ITEMS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int item = tid + NT * i;
        const int so = item / COLS;  // octet (wave-uniform: COLS is a multiple of 64)
        const int col = item - so * COLS;
        const int tin = t0 - p.pad_left + col;
        voff[i] = (col < W && tin >= 0 && tin < Tlim) ? tin * (XPL ? 16 : 4) : (int)0x80000000;
        soct[i] = __builtin_amdgcn_readfirstlane(so * (XPL ? 1 : 8)) * row_bytes;
    });
    float stage[XPL ? 1 : ITEMS][8];
    const u32x4 xdesc = {x_lo, x_hi, 0x7fffffffu, 0x00020000u};  // (the same descriptor as xrsrc, as four scalars for the asm below)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw + wave * 1024);
    // (plane input: global -> LDS without passing through registers -- buffer_load_dwordx4 ... lds writes lane l's 16 bytes to
    //  M0 base + 16 l: a wave's 64 consecutive items are 64 consecutive columns of one octet = 1 KiB contiguous in the slab;
    //  out-of-range lanes (padding, row ends) write zeros)
    u32x4 pstage[(XPL && !PARROT_XPL_DMA) ? ITEMS : 1][NPC];
    auto load_slab = [&](int c) __attribute__((always_inline)) {
        if constexpr (XPL && !PARROT_XPL_DMA) {
            const int cbase = c * 4 * row_bytes;
            static_for<ITEMS>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
                    pstage[i][pc] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff[i], cbase + soct[i] + pc * xpiece_bytes, 0);
            });
        } else if constexpr (XPL) {
            // (inline asm, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler orders every later ds_read behind an LDS-writing
            //  load it knows about -- a vmcnt wait in front of the CURRENT chunk's fragment reads -- and the scheduling constraints
            //  spill registers inside the K loop (round 3 measured that consumer 27 % slower than the converting kernel); hidden
            //  in asm the loads are just eight more VMEM instructions, published by the vmcnt(0) + barrier that ends the chunk)
            const int cbase = c * 4 * row_bytes;
            const unsigned dst = lds_base + (c & 1) * BUF_BYTES;
            static_for<ITEMS>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const unsigned d = dst + pc * PIECE_BYTES + NT * i * 16;
                    const int vo = voff[i], so = cbase + soct[i] + pc * xpiece_bytes;
                    const u32x4 xd = xdesc;
                    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(d), "v"(vo), "s"(xd), "s"(so) : "memory");
                }
            });
        } else {
            const int cbase = c * 32 * row_bytes;
            static_for<ITEMS>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    stage[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff[i], cbase + soct[i] + e * row_bytes, 0));
            });
        }
    };
    auto store_slab = [&](int buf) __attribute__((always_inline)) {
        char* dst = smem_raw + buf * BUF_BYTES;
        static_for<ITEMS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int item = tid + NT * i;
            if constexpr (XPL && !PARROT_XPL_DMA) {
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<u32x4*>(dst + pc * PIECE_BYTES + item * 16) = pstage[i][pc];
            } else if constexpr (XPL) {
                (void)dst; (void)item;  // (the loads of load_slab landed in LDS themselves; the barrier below publishes them)
            } else {
                unsigned q[4][NPC];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    SCH::split(pre_scale<SCH>(stage[i][2 * e], slope), pre_scale<SCH>(stage[i][2 * e + 1], slope), q[e]);
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)  // item = octet * COLS + column: consecutive lanes, consecutive 16-byte slots
                    *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES + item * 16) = uint4{q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
            }
        });
    };

    // ---- accumulators: C/D layout of 16x16 MFMA: column l & 15, rows 4 (l >> 4) + r --------------------------------
    // Prologue and epilogue address the (M, Tout) output / residual tile through buffer instructions (conv_mfma.h, RowTile): one
    // VGPR of lane-dependent offset per column tile, scalar row offsets, out-of-range columns / rows dropped by the hardware --
    // a few VALU instructions per accumulator register instead of ~25 (pointer arithmetic + bounds selects per element).
    f32x4 acc[TM][TN];
    const int m_wave = mblock * BM + wm * TM * 16;
    const int n_wave = t0 + wn * TN * 16;
    // accumulators start at bias * scale; the residual is added in the epilogue, after the sum, as the reference does
    // (xt = c2(xt); x = xt + x, models.py:37-38) -- for every tile shape and alignment alike
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        float bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = p.bias ? p.bias[min(m_wave + tm * 16 + 4 * g4 + r, p.M - 1)] * p.acc_scale : 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tm][tn][r] = bs[r];
    }

    // ---- weight stream ---------------------------------------------------------------------------------------------
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wfrag));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wfrag) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    int abase[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
        abase[tm] = __builtin_amdgcn_readfirstlane((int)((mblock * (BM / 16) + wm * TM + tm) * p.n_it * STEP_BYTES));
    const unsigned lane16 = lane * 16;
    // ONE register set per operand, refilled in place: a step's MFMAs run row tile by row tile, so A[tm] is dead after its
    // group (NTERM * TN MFMAs) and is re-fetched for the next step right there (3 groups = 36+ MFMAs ahead of its next use);
    // the column fragments are re-read from LDS during the step's last group, each right after its last MFMA.
    s16x8 A[TM][NPC], Bv[TN][NPC];
    auto load_a_tile = [&](s16x8 (&a)[NPC], int tm, int step) __attribute__((always_inline)) {
        const int soff = step * STEP_BYTES;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc)
            a[pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, abase[tm] + soff, 0));
    };
    const int colbase = wn * TN * 16 + l15;
    auto load_b_tile = [&](s16x8 (&bb)[NPC], const char* __restrict__ xs, int tn, int tap) __attribute__((always_inline)) {
        const char* src = xs + g4 * OCT_BYTES + (colbase + tn * 16 + tap * p.dil) * 16;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) bb[pc] = *reinterpret_cast<const s16x8*>(src + pc * PIECE_BYTES);
    };

    const int nchunks = p.nchunks;  // chunks of 32 channels
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) load_a_tile(A[tm], tm, 0);
    load_slab(0);
    store_slab(0);
    if constexpr (XPL && PARROT_XPL_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {  // one chunk = K straight-line steps
        const char* __restrict__ xs = smem_raw + (c & 1) * BUF_BYTES;
        const bool more = c + 1 < nchunks;
        if (more) load_slab(c + 1);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) load_b_tile(Bv[tn], xs, tn, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int t = 0; t < NTERM; ++t)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[tm][tn] = mfma16<SCH>(A[tm][SCH::pa(t)], Bv[tn][SCH::pb(t)], acc[tm][tn]);
                        if (tm == TM - 1 && t == NTERM - 1 && j + 1 < K) load_b_tile(Bv[tn], xs, tn, j + 1);
                    }
                load_a_tile(A[tm], tm, c * K + j + 1);  // (past the last step: the stream is padded by one step)
            }
        }
        // issue order: one memory instruction in the shadow of each MFMA -- in the chunk's first step the slab loads, a row
        // tile's weight refetch right after its group, a column tile's fragment refill right after its last MFMA
        constexpr int PER_TM = NTERM * TN;
#pragma unroll
        for (int i = 0; i < NPC * TN; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            int slab_left = (j == 0) ? ((XPL && PARROT_XPL_DMA) ? 0 : XPL ? ITEMS * NPC : ITEMS * 8) : 0;  // (plane input by DMA: the slab loads are asm, issued ahead of the step)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int m = 0; m < PER_TM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (tm == TM - 1 && m >= PER_TM - TN && j + 1 < K) {
#pragma unroll
                        for (int i = 0; i < NPC; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    } else if (slab_left > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        --slab_left;
                    }
                }
#pragma unroll
                for (int i = 0; i < NPC; ++i) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_slab((c + 1) & 1);
        // this wave's slab loads have landed in LDS: everything but the step's own weight refills (the TM * NPC youngest loads, for the
        // next chunk's first step -- issued long after the slab loads, returns are in order) has come back
        if constexpr (XPL && PARROT_XPL_DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * NPC) : "memory");
        __syncthreads();
    }

    // ---- epilogue (plain convs only): scale back, activation, residual, MRF accumulation modes ----------------------------
    // v = y_old + (max(acc * out_scale, relu ? 0 : -inf) + res)
    float* __restrict__ yb = p.y + (size_t)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const int out_row_bytes = p.Tout * 4;
    const bool late_res = rb != nullptr;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    const bool relu = p.act == ACT_RELU;
    const RowTile yt = row_tile(yb, p.M, p.Tout), rt = row_tile(late_res ? rb : yb, p.M, p.Tout);
    if (p.yplane) {
        // Producer of an operand plane (EPI_STORE only): y = max(acc * out_scale, relu ? 0 : -inf) + res as below, then ALSO (or
        // only: plane_only) split(pre(y)) with the NEXT layer's leaky ReLU -- the very instructions that layer's slab staging
        // would run on the stored fp32 value, so its MFMA operands are bit for bit the same -- as [piece][M / 8][Tout][8 x 16 bit].
        // A lane holds 4 of an octet's 8 channels of ONE column (C/D layout); the other 4 sit in lane ^ 16: v_permlane16_swap
        // between two tiles gives the even 16-lane rows the whole octet of the first tile and the odd rows that of the second
        // (the fused kernels' write_p): one 16-byte store per lane and piece, 16 lanes = 256 contiguous bytes.
        int vo[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n_wave + tn * 16 + l15;
            vo[tn] = (n < p.Ncols) ? (4 * g4 * p.Tout + n) * 4 : (int)0x80000000;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m0 = m_wave + tm * 16;
            float rv[TN][4];
            if (late_res) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rv[tn][r] = row_tile_load(rt, vo[tn], (m0 + r) * out_row_bytes);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[tm][tn][r] * p.out_scale;
                    if (relu) v = fmaxf(v, 0.f);
                    if (late_res) v = v + rv[tn][r];
                    acc[tm][tn][r] = v;
                }
            if (!p.plane_only) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) row_tile_store(yt, acc[tm][tn][r], vo[tn], (m0 + r) * out_row_bytes);
            }
        }
        char* pbase = reinterpret_cast<char*>(p.yplane) + (size_t)b * p.yplane_bstride;
        const size_t pa = reinterpret_cast<size_t>(pbase);
        const unsigned p_lo = __builtin_amdgcn_readfirstlane((unsigned)pa), p_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const int ypiece_bytes = (p.M >> 3) * p.Tout * 16;
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)p_hi << 32) | p_lo), 0,
                                                                                __builtin_amdgcn_readfirstlane(NPC * ypiece_bytes), 0x00020000);
        const float nslope = p.yplane_slope;
        constexpr bool PAIR_M = (TM % 2 == 0);  // pair row tiles (any TN), or -- one row tile per wave -- column tiles
        static_assert(PAIR_M || TN % 2 == 0, "operand-plane epilogue: an even number of row or column tiles per wave");
        constexpr int NA = PAIR_M ? TM / 2 : TM, NB = PAIR_M ? TN : TN / 2;
#pragma unroll
        for (int ia = 0; ia < NA; ++ia)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int tm0 = PAIR_M ? 2 * ia : ia, tm1 = PAIR_M ? 2 * ia + 1 : ia;
                const int tn0 = PAIR_M ? ib : 2 * ib, tn1 = PAIR_M ? ib : 2 * ib + 1;
                unsigned qq[2][2][NPC];  // [tile of the pair][channel pair e][piece]
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    SCH::split(pre_scale<SCH>(acc[tm0][tn0][2 * e], nslope), pre_scale<SCH>(acc[tm0][tn0][2 * e + 1], nslope), qq[0][e]);
                    SCH::split(pre_scale<SCH>(acc[tm1][tn1][2 * e], nslope), pre_scale<SCH>(acc[tm1][tn1][2 * e + 1], nslope), qq[1][e]);
                }
                // this lane's octet after the swap: even 16-lane rows own the first tile's, odd rows the second's
                const int tmq = (g4 & 1) ? tm1 : tm0, tnq = (g4 & 1) ? tn1 : tn0;
                const int oct = ((m_wave + tmq * 16) >> 3) + (g4 >> 1);
                const int n = n_wave + tnq * 16 + l15;
                const int off = (n < p.Ncols && oct * 8 < p.M) ? (oct * p.Tout + n) * 16 : (int)0x80000000;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{r0[0], r1[0], r0[1], r1[1]}, prsrc, off, pc * ypiece_bytes, 0);
                }
            }
        return;
    }
    if (p.epi16) {
        // 16-byte epilogue: the C/D layout gives a lane ONE column of four rows, i.e. four 4-byte accesses to four different rows
        // per tile (64-80 stores, and as many loads per residual / accumulate operand, per lane: ~100 clocks of issue each beside
        // another workgroup's MFMAs).  Each 16-row tile of the wave goes through a wave-private LDS region instead
        // ([16 rows][TN * 16 + 4 floats]: the slab buffers are dead after the last chunk's barrier, no further barrier needed: one
        // wave's DS instructions execute in order) and leaves as row segments: lane -> 4 consecutive columns of one row, a wave
        // instruction -> 3-4 contiguous 256- / 320-byte runs.  Same values, same arithmetic per element.
        constexpr int RS = TN * 16 + 4, GPR = TN * 4;  // staged row stride (floats; 4 RS = 16 mod 32: 2-way write conflicts only), 16-byte groups per row
        float* stg = reinterpret_cast<float*>(smem_raw) + wave * (16 * RS);
        int vq[TN], lq[TN];
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int g = lane + 64 * q, row = g / GPR, cg = g - row * GPR;
            const int n = n_wave + cg * 4;
            vq[q] = (n < p.Ncols) ? (row * p.Tout + n) * 4 : (int)0x80000000;
            lq[q] = row * RS + cg * 4;
        }
        const int wbase = 4 * g4 * RS + l15;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m0 = m_wave + tm * 16;
            const int soff = m0 * out_row_bytes;
            f32x4 rv[TN], yv[TN];
            if (late_res) {
#pragma unroll
                for (int q = 0; q < TN; ++q) rv[q] = row_tile_load4(rt, vq[q], soff);
            }
            if (has_acc) {
#pragma unroll
                for (int q = 0; q < TN; ++q) yv[q] = row_tile_load4(yt, vq[q], soff);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[tm][tn][r] * p.out_scale;
                    if (relu) v = fmaxf(v, 0.f);
                    stg[wbase + r * RS + tn * 16] = v;
                }
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + lq[q]);
                if (late_res) v = v + rv[q];
                if (has_acc) {
                    v = yv[q] + v;
                    if (do_div) v = v / p.div;
                }
                row_tile_store4(yt, v, vq[q], soff);
            }
        }
        return;
    }
    // rows that do not start on 16-byte boundaries (odd lengths): 4-byte accesses in the C/D layout, the same arithmetic
    int vo[TN];  // byte offset of this lane's (row 4 g4, column) inside the tile rows; 0x80000000: column out of range
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n_wave + tn * 16 + l15;
        vo[tn] = (n < p.Ncols) ? (4 * g4 * p.Tout + n) * 4 : (int)0x80000000;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m0 = m_wave + tm * 16;
        float v[TN][4];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[tn][r] = acc[tm][tn][r] * p.out_scale;
        if (relu) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[tn][r] = fmaxf(v[tn][r], 0.f);
        }
        if (late_res) {
            float rv[TN][4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[tn][r] = row_tile_load(rt, vo[tn], (m0 + r) * out_row_bytes);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[tn][r] = v[tn][r] + rv[tn][r];
        }
        if (has_acc) {
            float yv[TN][4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[tn][r] = row_tile_load(yt, vo[tn], (m0 + r) * out_row_bytes);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[tn][r] = yv[tn][r] + v[tn][r];
            if (do_div) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[tn][r] = v[tn][r] / p.div;
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) row_tile_store(yt, v[tn][r], vo[tn], (m0 + r) * out_row_bytes);
    }
}

template <class SCH, int WAVES_M, int WAVES_N, int TM, int TN, int K, int MINW = 2, bool XPL = false>
inline hipError_t launch_conv_split16_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16, COLS = (BN + CONV_HALO + 63) / 64 * 64;
    // slab double buffer, or the epilogue's wave-private staging rows ([16][TN * 16 + 4] floats per wave) where that is larger
    const size_t lds = std::max((size_t)2 * SCH::NP * 4 * COLS * 16, (size_t)WAVES_M * WAVES_N * 16 * (TN * 16 + 4) * sizeof(float));
    auto kern = conv_split16_kernel<SCH, WAVES_M, WAVES_N, TM, TN, K, MINW, XPL>;
    static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned tiles = (unsigned)p.tiles_n * p.B, n_mb = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3((tiles + 7) / 8 * 8 * n_mb), dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    return hipGetLastError();
}

// tile variants: 0 = 128 x 128 (waves 2x2, wave tile 64x64 = 4x4 MFMA tiles), 1 = 64 x 192 (waves 1x4, wave tile 64x48);
//                2 = 128 x 64, 3 = 64 x 64: the same row tiling with a third / half of the columns, for launches that would not
//                    give every CU a workgroup (small batches): the latency of a workgroup is its K loop x MFMAs per step
//                    (8 x 1 / 4 x 1 waves, launch_conv_split16_small_s below);
//                4 = 128 x 160 (waves 2x2, wave tile 64x80): for launches whose 128-column tiling ends in a half-empty round
//                    (split16_wide_fits below)
// The 64-row layers with k = 11 (stage 2) run on a 64 x 128 tile of 2 x 2 waves (32 x 64 each) at three workgroups per CU instead of
// 64 x 192 (1 x 4 waves, 64 x 48 each) at two: 2.32 -> 2.21 ms per step (round 3 A/B; the switch is gone, k = 7 / 9 keep 64 x 192).
// (k: the layer's tap count -- the 64-row variant 1 is the 64 x 128 kernel for k = 11 and the 64 x 192 one for k = 7 / 9: the
//  host's tile width must be the kernel's BN, or the launch carries workgroups whose tiles are all out of range)
inline void split16_tile(int variant, int& bm, int& bn, int k = 11) {
    bm = (variant & 1) ? 64 : 128;
    bn = variant == 4 ? 160 : variant >= 2 ? 64 : (variant ? (k == 11 ? 128 : 192) : 128);
}
// Two 128-row workgroups fit a CU, so a launch runs in rounds of 2 x CUs workgroups, and a workgroup takes the same time whether
// its CU is shared or not (profiles/r03q_s16_launch_timeline.jsonl): stage 0 at B = 64 is 1280 tiles of 128 x 128 = 2.5 rounds, the last one
// with half of the slots empty.  The same layer on 128 x 160 tiles is 1024 workgroups = 2 full rounds of 1.25x the work each.
// Rule: rounds x tile width; ties go to the wider tile (a 64 x 80 wave tile fetches 20 % fewer weight fragments per MAC: stage 1 at
// B = 64, 2560 -> 2048 workgroups = 5 -> 4 rounds, the conv_split16 kernels together 6.80 -> 6.74 ms per step).  (Tile shapes never
// change the arithmetic of an output.)
inline bool split16_wide_fits(long ncols, long rows, int n_mb, int n_cus) {
    const long slots = 2L * n_cus;
    const long wg128 = (ncols + 127) / 128 * rows * n_mb, wg160 = (ncols + 159) / 160 * rows * n_mb;
    const long cost128 = (wg128 + slots - 1) / slots * 128, cost160 = (wg160 + slots - 1) / slots * 160;
    return cost160 <= cost128;
}
// (k = 3: three steps per 32-channel chunk do not amortise the larger slab -- measured 3-6 % slower than conv_split_kernel)
// (the single-MFMA modes take it too: with a third of the MFMAs per chunk the 16-channel chunks of conv_split_kernel are bound by
//  their per-chunk barrier / staging latency -- bf16 step 13.8 -> 11.9 ms, the wide layers 495 -> 746 TF)
inline bool split16_has(int scheme, int k) { return (scheme == SchF16x3::ID || scheme == SchBf16::ID || scheme == SchF16::ID) && (k == 7 || k == 9 || k == 11); }
// The small tiles (variants 2 / 3) run alone on their CU and stream their weights from L2 through the vector L1 at 64 B/clk: on a
// 2 x 2 wave grid both column waves fetch the same row fragments -- 360 KB per 32-channel chunk at k = 11 = 5.6 k clocks against
// 4.5 k of MFMA (round-3 chunk trace, profiles/r03q_s16_chunk_trace.log).  Their waves are stacked along M instead (each wave all 64 columns): every weight fragment
// is fetched once and more LDS fragments are read (128 B/clk).  One utterance, stage 0, k = 11: 32.8 -> 26.2 us per launch (8 x 1
// waves; 30.2 with 4 x 1), FFN conv 35.0 -> 24.8, stage 2 (64 x 64 tile, 4 x 1 waves) 19.1 -> 11.2; B = 1 3.11 -> 2.90 ms, B = 4
// 3.85 -> 3.62 ms (profiles/r03r_small_tile_wave_grid_ab.txt).
template <class SCH>
inline hipError_t launch_conv_split16_small_s(int variant, const ConvParams& p, hipStream_t s) {
    if (variant == 2) switch (p.k) {
            case 7: return launch_conv_split16_t<SCH, 8, 1, 1, 4, 7, 1>(p, s);
            case 9: return launch_conv_split16_t<SCH, 8, 1, 1, 4, 9, 1>(p, s);
            case 11: return launch_conv_split16_t<SCH, 8, 1, 1, 4, 11, 1>(p, s);
            default: return hipErrorInvalidValue;
        }
    switch (p.k) {
        case 7: return launch_conv_split16_t<SCH, 4, 1, 1, 4, 7>(p, s);
        case 9: return launch_conv_split16_t<SCH, 4, 1, 1, 4, 9>(p, s);
        case 11: return launch_conv_split16_t<SCH, 4, 1, 1, 4, 11>(p, s);
        default: return hipErrorInvalidValue;
    }
}
template <class SCH>
inline hipError_t launch_conv_split16_wide_s(int variant, const ConvParams& p, hipStream_t s) {  // variants 1 and 4
    if (variant == 4) switch (p.k) {
            case 7: return launch_conv_split16_t<SCH, 2, 2, 4, 5, 7>(p, s);
            case 9: return launch_conv_split16_t<SCH, 2, 2, 4, 5, 9>(p, s);
            case 11: return launch_conv_split16_t<SCH, 2, 2, 4, 5, 11>(p, s);
            default: return hipErrorInvalidValue;
        }
    // (round 6: the same 64 x 128 tile as 1 x 2 waves of 64 x 64 in 128-thread workgroups -- half the LDS fragment reads per MFMA, 6
    //  waves per CU instead of 12 -- is bit-identical and SLOWER: 421 / 468 vs 347 / 389 us per launch, profiles/EXPERIMENTS.md)
    if (p.k == 11) return launch_conv_split16_t<SCH, 2, 2, 2, 4, 11, 3>(p, s);
    switch (p.k) {
        case 7: return launch_conv_split16_t<SCH, 1, 4, 4, 3, 7>(p, s);
        case 9: return launch_conv_split16_t<SCH, 1, 4, 4, 3, 9>(p, s);
        default: return hipErrorInvalidValue;
    }
}
template <class SCH>
inline hipError_t launch_conv_split16_s(int variant, const ConvParams& p, hipStream_t s) {  // variant 0
    switch (p.k) {
        case 7: return launch_conv_split16_t<SCH, 2, 2, 4, 4, 7>(p, s);
        case 9: return launch_conv_split16_t<SCH, 2, 2, 4, 4, 9>(p, s);
        case 11: return launch_conv_split16_t<SCH, 2, 2, 4, 4, 11>(p, s);
        default: return hipErrorInvalidValue;
    }
}
// operand-plane input (ConvParams::xplane: the MRF's k = 7 / 11 layers only): the XPL instantiations of every tile variant
template <class SCH>
inline hipError_t launch_conv_split16_xpl_s(int variant, const ConvParams& p, hipStream_t s) {
    if (p.k != 7 && p.k != 11) return hipErrorInvalidValue;
    const bool k7 = p.k == 7;
    switch (variant) {
        case 0: return k7 ? launch_conv_split16_t<SCH, 2, 2, 4, 4, 7, 2, true>(p, s) : launch_conv_split16_t<SCH, 2, 2, 4, 4, 11, 2, true>(p, s);
        case 1: return k7 ? launch_conv_split16_t<SCH, 1, 4, 4, 3, 7, 2, true>(p, s) : launch_conv_split16_t<SCH, 2, 2, 2, 4, 11, 3, true>(p, s);
        case 2: return k7 ? launch_conv_split16_t<SCH, 8, 1, 1, 4, 7, 1, true>(p, s) : launch_conv_split16_t<SCH, 8, 1, 1, 4, 11, 1, true>(p, s);
        case 3: return k7 ? launch_conv_split16_t<SCH, 4, 1, 1, 4, 7, 2, true>(p, s) : launch_conv_split16_t<SCH, 4, 1, 1, 4, 11, 2, true>(p, s);
        case 4: return k7 ? launch_conv_split16_t<SCH, 2, 2, 4, 5, 7, 2, true>(p, s) : launch_conv_split16_t<SCH, 2, 2, 4, 5, 11, 2, true>(p, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_conv_split16_xpl_f16x3(int variant, const ConvParams& p, hipStream_t s);    // csrc/tu_split16_xpl.hip
hipError_t launch_conv_split16_xpl_bf16(int variant, const ConvParams& p, hipStream_t s);     // csrc/tu_split16_single.hip
hipError_t launch_conv_split16_xpl_f16(int variant, const ConvParams& p, hipStream_t s);      // csrc/tu_split16_single_f16.hip
hipError_t launch_conv_split16_wide_f16x3(int variant, const ConvParams& p, hipStream_t s);   // csrc/tu_split16_wide.hip
hipError_t launch_conv_split16_small_f16x3(int variant, const ConvParams& p, hipStream_t s);  // csrc/tu_split16.hip
hipError_t launch_conv_split16_f16x3(int variant, const ConvParams& p, hipStream_t s);  // csrc/tu_split16.hip
hipError_t launch_conv_split16_bf16(int variant, const ConvParams& p, hipStream_t s);   // csrc/tu_split16_single.hip
hipError_t launch_conv_split16_f16(int variant, const ConvParams& p, hipStream_t s);    // csrc/tu_split16_single.hip
inline hipError_t launch_conv_split16(int scheme, int variant, const ConvParams& p, hipStream_t s) {
    if (p.xplane)
        return scheme == SchF16x3::ID ? launch_conv_split16_xpl_f16x3(variant, p, s)
               : scheme == SchBf16::ID ? launch_conv_split16_xpl_bf16(variant, p, s)
               : scheme == SchF16::ID  ? launch_conv_split16_xpl_f16(variant, p, s) : hipErrorInvalidValue;
    if (scheme == SchF16x3::ID)
        return (variant == 2 || variant == 3)   ? launch_conv_split16_small_f16x3(variant, p, s)
               : (variant == 1 || variant == 4) ? launch_conv_split16_wide_f16x3(variant, p, s)
                                                : launch_conv_split16_f16x3(variant, p, s);
    if (scheme == SchBf16::ID) return launch_conv_split16_bf16(variant, p, s);
    if (scheme == SchF16::ID) return launch_conv_split16_f16(variant, p, s);
    return hipErrorInvalidValue;
}

}  // namespace parrot
