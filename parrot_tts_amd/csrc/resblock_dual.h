// resblock_dual.h -- the fused ResBlock1 pair kernels of resblock_split.h as DUAL-WINDOW ANTI-PHASE workgroups.
//
// Why.  A fused pair kernel alternates two phases per conv: M (the conv: MFMAs fed from LDS) and V (write_p: leaky ReLU,
// scale, 16-bit split, half-wave swap, LDS store -- ~300 VALU instructions per lane and conv, none of which can start
// before the conv's last MFMA).  With one window per workgroup the co-resident workgroups of a CU run in lockstep (they
// start together and take equally long), so the matrix pipe idles during every V phase and the VALU during every M phase:
// rocprofv3 (profiles/r03a_sqdeep_*.csv) shows 5.4-13.5 VALU instructions per MFMA and SQ_VALU_MFMA_BUSY 0.37-0.61 --
// M and V times ADD UP instead of overlapping.
//
// What.  One workgroup = 8 waves = TWO independent windows A (waves 0-3) and B (waves 4-7); a workgroup's waves go to
// the SIMDs round-robin, so every SIMD holds one A wave and one B wave.  B runs one barrier behind A: while A multiplies, B
// converts, and vice versa -- every s_barrier of the workgroup swaps the roles.  The MFMA stream of one window and the
// VALU stream of the other are independent instruction streams of two waves of the SAME SIMD, which the hardware issues
// side by side (an MFMA occupies the matrix pipe for 32 clocks and one issue slot in eight).  96 KiB of LDS and up to 256
// VGPRs per workgroup: one workgroup per CU.
//
// VALU diet on top (per lane and pair of convs, 32 channels: 973 -> ~560 instructions):
//   * the sequence-edge masking (v_cndmask per element and conv) only exists in the EDGE instantiation of the main loop,
//     taken by windows that touch a sequence end; interior windows run mask-free code;
//   * the running residual stays in the accumulator's scale between pairs (no multiply per element and pair), the
//     accumulator init of the second conv is one fma per element;
//   * the low fp16 piece is formed with v_fma_mix_f32 (x - float(hi) in one instruction, no separate f16 -> f32 convert);
//   * fragment reads are not clamped to the window: the operand buffer has guard bytes on both sides, reads that leave
//     a plane land in a neighbouring plane (finite garbage that only ever feeds garbage columns, as before).
// Data layout, weight stream, halo bookkeeping and epilogue modes are resblock_split.h's.
#pragma once
#include "resblock_split.h"

namespace parrot {

#ifdef RBD_TRACE
// experiment builds only (tools/build_exp.sh trace -DRBD_TRACE): shader-clock timestamps of one workgroup's phases
__device__ unsigned long long g_rbd_trace[8][64];
#define RBD_MARK()                                                                                  \
    do {                                                                                            \
        if (trace_on && lane == 0 && tr_i < 62) g_rbd_trace[wave][tr_i] = __builtin_amdgcn_s_memtime(); \
        ++tr_i;                                                                                     \
    } while (0)
#define RBD_TRACE_INIT()                                                                            \
    const bool trace_on = p.stagger != 0 && (int)blockIdx.x == (int)gridDim.x / 2;                  \
    int tr_i = 0;                                                                                   \
    if (trace_on && lane == 0) {                                                                    \
        unsigned hwid;                                                                              \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));                          \
        g_rbd_trace[wave][63] = hwid;                                                               \
    }
#else
#define RBD_MARK() do { } while (0)
#define RBD_TRACE_INIT() do { } while (0)
#endif

constexpr int RBD_GUARD = 1024;  // guard bytes either side of a window's operand buffer (max read-ahead 40 columns x 16 B)

// (x0, x1) -> NP packed 16-bit pairs; the fp16 two-piece split takes the fma-mix shortcut
template <class SCH>
__device__ __forceinline__ void split_fast(float x0, float x1, unsigned (&q)[SCH::NP]) {
    if constexpr (SCH::F16 && SCH::NP == 2) {
        q[0] = pk_f16(x0, x1);
        float r0, r1;  // x - float(hi): exact, one instruction each (f16 source selected by op_sel)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(q[0]), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(q[0]), "v"(x1));
        q[1] = pk_f16(r0, r1);
    } else {
        SCH::split(x0, x1, q);
    }
}

// NCH = channel chunks of 16: 2 -> 32 channels (window 384: four waves side by side), 4 -> 64 channels (window 192: 2 row
// tiles x 2 column halves) -- per window exactly the tiling of resblock_split_kernel<SCH, NCH>.
template <class SCH, int NCH, int NWIN>
__global__ __launch_bounds__(256 * NWIN, NWIN == 1 ? 3 : 1) void resblock_dual_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT, STEP_BYTES = SCH::NP * 1024;
    constexpr int C = 16 * NCH, WAVES_M = NCH / 2, WAVES_N = 4 / WAVES_M, NTW = 3, W = WAVES_N * NTW * 32;
    constexpr int OCT_BYTES = W * 16, CH_BYTES = 2 * OCT_BYTES, PIECE_BYTES = NCH * CH_BYTES;
    constexpr int WIN_BYTES = NPC * PIECE_BYTES + 2 * RBD_GUARD;
    static_assert(NCH == 2 || NCH == 4, "32 or 64 channels");
    static_assert((NPC - 1) * PIECE_BYTES + (NCH - 1) * CH_BYTES + OCT_BYTES < 65536, "fragment offsets must fit the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];  // NWIN x WIN_BYTES
    const int K = p.k;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int win = (NWIN == 2) ? wave >> 2 : 0, w4 = wave & 3;  // window of this wave, wave inside the window
    const int wm = w4 / WAVES_N;               // row tile (32 output channels)
    const int wn = w4 % WAVES_N;               // column group (96 columns)
    const int half = lane >> 5;
    const int l31 = lane & 31;
    char* const smem_raw = smem_all + win * WIN_BYTES + RBD_GUARD;
    // two consecutive windows of the launch per workgroup (neighbours in the same row share their halo columns in L2);
    // an odd window count leaves the last workgroup's B without work: it re-computes the last window and stores nothing
    const int total = p.tiles * p.B;
    const int wid_raw = NWIN * (int)blockIdx.x + win;
    const bool live = wid_raw < total;
    const int wid = live ? wid_raw : total - 1;
    const int b = wid / p.tiles;
    const int tile = wid - b * p.tiles;
    const int t_base = tile * p.TT - p.H;  // sequence position of window column 0
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    const bool edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);  // window not fully inside the row
    RBD_TRACE_INIT();
    RBD_MARK();  // 0: start

    {   // guard bytes: zero (never NaN patterns from an earlier kernel's LDS contents)
        const int t256 = tid & 255;
        char* g = smem_all + win * WIN_BYTES + (t256 < 128 ? 0 : NPC * PIECE_BYTES);  // (the second half lands on the tail guard)
        *reinterpret_cast<uint2*>(g + t256 * 8) = uint2{0u, 0u};
    }

    int col[NTW];   // this lane's window column per tile
    bool tok[NTW];  // ... inside the sequence
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        col[nt] = wn * (NTW * 32) + nt * 32 + l31;
        const int t = t_base + col[nt];
        tok[nt] = t >= 0 && t < Tlim;
    }

    // ---- R <- x window, in the C/D layout: row m = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, column = lane & 31 -----
    f32x16 R[NTW], acc[NTW];
    {
        const size_t xaddr = reinterpret_cast<size_t>(p.x + (size_t)b * C * p.T);
        const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)x_hi << 32) | x_lo), 0, 0x7fffffff, 0x00020000);
        const int row_bytes = p.T * 4;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int voff = tok[nt] ? (t_base + col[nt]) * 4 + 4 * half * row_bytes : (int)0x80000000;  // out of range -> 0
#pragma unroll
            for (int r = 0; r < 16; ++r)
                R[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes, 0));
        }
    }

    // ---- weights: one buffer descriptor, per-lane constant offset, scalar step offset (stream order [conv][row tile][chunk * K + tap]) ----
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wstream));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wstream) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    s16x8 A[2][NPC], Bv[2][NTW][NPC];
    auto load_a_piece = [&](s16x8 (&a)[NPC], int pc, int gstep) __attribute__((always_inline)) {
        a[pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, gstep * STEP_BYTES, 0));
    };
    const int conv_steps = WAVES_M * NCH * K;  // steps of one conv in the stream
    int gbase = wm * NCH * K;                  // this wave's first step of the current conv
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
        load_a_piece(A[0], pc, gbase);      // (tap 0, chunk 0)
        load_a_piece(A[1], pc, gbase + K);  // (tap 0, chunk 1)
    }

    // B runs one barrier behind A from here on (A converts its first operand while B is still loading)
    if (NWIN == 2 && win) __syncthreads();

    const int center = (K - 1) / 2;
    // fragment base address of this lane per tile: column col[nt], octet `half` of chunk 0, piece 0 (unclamped: guard bytes)
    const char* fbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) fbase[nt] = smem_raw + half * OCT_BYTES + col[nt] * 16;
    auto load_b = [&](s16x8 (&bb)[NTW][NPC], int ch, int shift16) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const char* src = fbase[nt] + shift16;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[nt][pc] = *reinterpret_cast<const s16x8*>(src + (pc * PIECE_BYTES + ch * CH_BYTES));
        }
    };
    // one conv out of P into acc (initialised by the caller): see resblock_split_kernel
    auto conv = [&](int dil) __attribute__((always_inline)) {
        load_b(Bv[0], 0, -center * dil * 16);
        for (int j = 0; j < K; ++j) {
            const int shift16 = (j - center) * dil * 16;
            const int n0 = (j + 1 < K) ? gbase + j + 1 : gbase + conv_steps;  // (next tap, chunk 0); chunk c is c * K steps further
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int set = ch & 1;
#ifndef EXP_RBD_NO_B
                if (ch + 1 < NCH) load_b(Bv[set ^ 1], ch + 1, shift16);
                else load_b(Bv[set ^ 1], 0, shift16 + dil * 16);
#endif
                const int nx = (ch + 2 < NCH) ? gbase + (ch + 2) * K + j : n0 + (ch + 2 - NCH) * K;  // two steps ahead
#pragma unroll
                for (int t = 0; t < NTERM; ++t) {
#ifndef EXP_RBD_NO_MFMA
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[nt] = mfma32<SCH>(A[set][SCH::pa(t)], Bv[set][nt][SCH::pb(t)], acc[nt]);
#endif
#ifndef EXP_RBD_NO_A
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        if (t == SCH::last_a(pc)) load_a_piece(A[set], pc, nx);
#else
                    (void)nx;
#endif
                }
            }
            // issue order: one memory instruction in the shadow of each MFMA (see conv_split.h)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                int ds_left = NPC * NTW;
#pragma unroll
                for (int m = 0; m < NTERM * NTW; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    bool refetch = false;
#pragma unroll
                    for (int pc = 1; pc < NPC; ++pc) refetch = refetch || (m == (SCH::last_a(pc) + 1) * NTW - 1);
                    if (refetch) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else if (ds_left > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        --ds_left;
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        gbase += conv_steps;
    };
    auto bias_rows = [&](const float* __restrict__ bias, float (&bv)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half];
    };

    // The whole pair loop, instantiated with and without the sequence-edge masking.  R is kept in the scale `rs` of the
    // accumulator it was last taken from (rs = 1 for the freshly loaded window).
    float rs = 1.f;
    auto pairs = [&](auto edge_c) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        // accumulator tile -> operand buffer: lrelu(c v) = max(c v, c slope v), (mask,) split, half-wave swap, 16-byte stores
        auto write_p = [&](const f32x16 (&v)[NTW], float mul) __attribute__((always_inline)) {
            const float m2 = mul * p.slope;
#ifdef EXP_RBD_NO_WP
            if (p.n_conv >= 0) return;
#endif
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned qq[2][2][NPC];  // [octet g][pair e][piece]
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x2 vv = {v[nt][8 * q + 4 * g + 2 * e], v[nt][8 * q + 4 * g + 2 * e + 1]};
                            const f32x2 a = vv * mul, bq = vv * m2;
                            float v0 = max_nc(a[0], bq[0]), v1 = max_nc(a[1], bq[1]);
                            if constexpr (EDGE) {
                                v0 = tok[nt] ? v0 : 0.f;
                                v1 = tok[nt] ? v1 : 0.f;
                            }
                            split_fast<SCH>(v0, v1, qq[g][e]);
                        }
                    char* dst = smem_raw + (2 * wm + q) * CH_BYTES + half * OCT_BYTES + col[nt] * 16;
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                        *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
                    }
                    // (keep the groups apart: interleaving all 48 elements' chains costs more registers than the 168 of three workgroups per CU)
                    if constexpr (NWIN == 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        RBD_MARK();  // 1: before the first conversion
        write_p(R, SCH::XS);  // (out-of-range columns were loaded as zeros)
        RBD_MARK();  // 2
        __syncthreads();
        RBD_MARK();  // 3
        for (int m = 0; m < p.n_conv; m += 2) {
            float bv[16];
            // h = conv_d(P) + b1   (accumulator scale s1 = XS * wsc[m])
            const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
            bias_rows(p.bias[m], bv);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = bv[r] * s1;
            conv(p.dil[m]);
            RBD_MARK();  // 4 + 8i: conv1 done
            __syncthreads();  // every wave of the window is done reading P (and the other window's role flips)
            RBD_MARK();  // 5
            write_p(acc, SCH::XS / s1);  // P = split(lrelu(h)), masked
            RBD_MARK();  // 6
            __syncthreads();
            RBD_MARK();  // 7
            // R = conv_1(P) + b2 + R: the residual rides in the accumulator (scale s2); R is in scale rs
            bias_rows(p.bias[m + 1], bv);
            const float rr = s2 / rs;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = fmaf(R[nt][r], rr, bv[r] * s2);  // = s2 * (R + b2), one rounding
            conv(p.dil[m + 1]);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (EDGE) R[nt][r] = tok[nt] ? acc[nt][r] : 0.f;
                    else R[nt][r] = acc[nt][r];
                }
            rs = s2;
            RBD_MARK();  // 8: conv2 done
            if (m + 2 < p.n_conv) {
                __syncthreads();
                RBD_MARK();  // 9
                write_p(R, SCH::XS / rs);
                RBD_MARK();  // 10
                __syncthreads();
                RBD_MARK();  // 11
            }
        }
    };
    if (edge) pairs(std::true_type{});
    else pairs(std::false_type{});
    if (NWIN == 2 && !win) __syncthreads();  // A's share of the barrier B spent at the start

    // ---- write the central TT columns (registers -> global, 128-byte runs per row) ---------------------------------
    float* __restrict__ yb = p.y + (size_t)b * C * p.T + (size_t)(32 * wm) * p.T;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    const float irs = 1.f / rs;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = col[nt] - p.H;
        const int t = tile * p.TT + c;
        const bool ok = live && c >= 0 && c < p.TT && t < p.T;
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = 0.f;
        if (has_acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = yb[ok ? (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.T + t : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = yv[r] + R[nt][r] * irs;
            if (do_div) v = v / p.div;
            if (ok) yb[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.T + t] = v;
        }
    }
}

// ---- 16-channel variant (resblock16_split_kernel's tiling: 16x16x32 MFMA, k-step = 16 channels x two taps, window 768) ----
template <class SCH, int NWIN>
__global__ __launch_bounds__(256 * NWIN, NWIN == 1 ? 3 : 1) void resblock16_dual_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT, STEP_BYTES = SCH::NP * 1024;
    constexpr int C = 16, NT = 12, W = RBS16_W, NP = NT / 2;
    constexpr int OCT_BYTES = W * 16, PIECE_BYTES = 2 * OCT_BYTES;  // [piece][octet][col][8 channels]
    constexpr int WIN_BYTES = NPC * PIECE_BYTES + 2 * RBD_GUARD;
    static_assert((NPC - 1) * PIECE_BYTES + NT * 256 < 65536, "fragment offsets must fit the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];  // 2 x WIN_BYTES
    const int K = p.k, S = (K + 1) / 2;  // tap pairs per conv (even: the host only takes k = 3, 7, 11)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int win = (NWIN == 2) ? wave >> 2 : 0, w4 = wave & 3;
    const int g4 = lane >> 4;  // C/D row group; as an operand lane: channel octet g4 & 1, tap parity g4 >> 1
    const int l15 = lane & 15;
    char* const smem_raw = smem_all + win * WIN_BYTES + RBD_GUARD;
    const int total = p.tiles * p.B;
    const int wid_raw = NWIN * (int)blockIdx.x + win;
    const bool live = wid_raw < total;
    const int wid = live ? wid_raw : total - 1;
    const int b = wid / p.tiles;
    const int tile = wid - b * p.tiles;
    const int t_base = tile * p.TT - p.H;
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    const bool edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);
    RBD_TRACE_INIT();
    RBD_MARK();  // 0: start
    const int col0 = w4 * (NT * 16) + l15;  // this lane's window column in tile 0 (tile nt adds 16 nt)

    {   // guard bytes: zero
        const int t256 = tid & 255;
        char* g = smem_all + win * WIN_BYTES + (t256 < 128 ? 0 : NPC * PIECE_BYTES);
        *reinterpret_cast<uint2*>(g + t256 * 8) = uint2{0u, 0u};
    }

    f32x4 R[NT], acc[NT];
    {
        const size_t xaddr = reinterpret_cast<size_t>(p.x + (size_t)b * C * p.T);
        const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)x_hi << 32) | x_lo), 0, 0x7fffffff, 0x00020000);
        const int row_bytes = p.T * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int t = t_base + col0 + nt * 16;
            const int voff = (t >= 0 && t < Tlim) ? t * 4 + 4 * g4 * row_bytes : (int)0x80000000;  // out of range -> 0
#pragma unroll
            for (int r = 0; r < 4; ++r) R[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, r * row_bytes, 0));
        }
    }

    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wstream));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wstream) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    s16x8 A[2][NPC], Bv[2][2][NPC];  // weights: two step sets; operands: two pair sets x two tiles x NPC pieces
    auto load_a_piece = [&](s16x8 (&a)[NPC], int pc, int gstep) __attribute__((always_inline)) {
        a[pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, gstep * STEP_BYTES, 0));
    };
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
        load_a_piece(A[0], pc, 0);
        load_a_piece(A[1], pc, 1);
    }

    if (NWIN == 2 && win) __syncthreads();  // B runs one barrier behind A

    const int center = (K - 1) / 2;
    int gstep = 0;
    const char* const fb = smem_raw + (g4 & 1) * OCT_BYTES + col0 * 16;  // this lane's fragment base (tile 0, piece 0), unclamped reads
    // fragments of tile pair pr at byte shift sh16 (= 16 x (2s + tap parity - center) * dil)
    auto load_b = [&](s16x8 (&bb)[2][NPC], int pr, int sh16) __attribute__((always_inline)) {
        const char* src = fb + sh16;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[q][pc] = *reinterpret_cast<const s16x8*>(src + ((2 * pr + q) * 256 + pc * PIECE_BYTES));
    };
    auto conv = [&](int dil) __attribute__((always_inline)) {
        const int tpd16 = ((g4 >> 1) * dil - center * dil) * 16;  // this lane's tap-parity shift, centred (bytes)
        load_b(Bv[0], 0, tpd16);
        for (int s2 = 0; s2 < S; s2 += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int s = s2 + sub;
                const int sh_cur = tpd16 + 32 * s * dil, sh_next = sh_cur + 32 * dil;
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    const int cur = pr & 1;  // NP is even: every step starts on operand set 0
                    if (pr + 1 < NP) load_b(Bv[cur ^ 1], pr + 1, sh_cur);
                    else load_b(Bv[cur ^ 1], 0, sh_next);  // (after the last step: discarded)
#pragma unroll
                    for (int t = 0; t < NTERM; ++t) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[2 * pr + q] = mfma16<SCH>(A[sub][SCH::pa(t)], Bv[cur][q][SCH::pb(t)], acc[2 * pr + q]);
                        if (pr == NP - 1) {  // last pair of the step: each weight piece is dead after its last term
#pragma unroll
                            for (int pc = 0; pc < NPC; ++pc)
                                if (t == SCH::last_a(pc)) load_a_piece(A[sub], pc, gstep + s + 2);
                        }
                    }
                }
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int pr = 0; pr < NP; ++pr)
#pragma unroll
                    for (int m = 0; m < 2 * NTERM; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (m < 2 * NPC) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        bool refetch = false;
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) refetch = refetch || (m == 2 * (SCH::last_a(pc) + 1) - 1);
                        if (pr == NP - 1 && refetch) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        gstep += S;
    };

    auto pairs = [&](auto edge_c) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        auto write_p = [&](const f32x4 (&v)[NT], float mul) __attribute__((always_inline)) {
            const float m2 = mul * p.slope;
#pragma unroll
            for (int np = 0; np < NT; np += 2) {  // tile pairs: lane groups with even g4 end up owning tile np's octet, odd ones tile np+1's
                unsigned qq[2][2][NPC];           // [tile of the pair][pair e][piece]
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bool tk = true;
                    if constexpr (EDGE) {
                        const int t = t_base + col0 + (np + u) * 16;
                        tk = t >= 0 && t < Tlim;
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 vv = {v[np + u][2 * e], v[np + u][2 * e + 1]};
                        const f32x2 a = vv * mul, bq = vv * m2;
                        float v0 = max_nc(a[0], bq[0]), v1 = max_nc(a[1], bq[1]);
                        if constexpr (EDGE) {
                            v0 = tk ? v0 : 0.f;
                            v1 = tk ? v1 : 0.f;
                        }
                        split_fast<SCH>(v0, v1, qq[u][e]);
                    }
                }
                const int c = col0 + (np + (g4 & 1)) * 16;
                char* dst = smem_raw + (g4 >> 1) * OCT_BYTES + c * 16;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                    *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
                }
                if constexpr (NWIN == 1) __builtin_amdgcn_sched_barrier(0);
            }
        };
        RBD_MARK();  // 1
        write_p(R, SCH::XS);
        RBD_MARK();  // 2
        __syncthreads();
        RBD_MARK();  // 3
        for (int m = 0; m < p.n_conv; m += 2) {
            float bv[4];
            const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[m][4 * g4 + r] * s1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt][r] = bv[r];
            conv(p.dil[m]);
            RBD_MARK();  // 4 + 8i
            __syncthreads();
            RBD_MARK();  // 5
            write_p(acc, SCH::XS / s1);
            RBD_MARK();  // 6
            __syncthreads();
            RBD_MARK();  // 7
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = p.bias[m + 1][4 * g4 + r] * s2;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt][r] = bv[r];
            conv(p.dil[m + 1]);
            const float i2 = 1.f / s2;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bool tk = true;
                if constexpr (EDGE) {
                    const int t = t_base + col0 + nt * 16;
                    tk = t >= 0 && t < Tlim;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaf(acc[nt][r], i2, R[nt][r]);  // the residual is added after the sum, as the reference does
                    R[nt][r] = (EDGE && !tk) ? 0.f : v;
                }
            }
            RBD_MARK();  // 8
            if (m + 2 < p.n_conv) {
                __syncthreads();
                RBD_MARK();  // 9
                write_p(R, SCH::XS);
                RBD_MARK();  // 10
                __syncthreads();
                RBD_MARK();  // 11
            }
        }
    };
    if (edge) pairs(std::true_type{});
    else pairs(std::false_type{});
    if (NWIN == 2 && !win) __syncthreads();  // A's share of the barrier B spent at the start

    float* __restrict__ yb = p.y + (size_t)b * C * p.T;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = col0 + nt * 16 - p.H;
        const int t = tile * p.TT + c;
        const bool ok = live && c >= 0 && c < p.TT && t < p.T;
        float yv[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = yb[ok ? (size_t)(4 * g4 + r) * p.T + t : 0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = yv[r] + R[nt][r];
            if (do_div) v = v / p.div;
            if (ok) yb[(size_t)(4 * g4 + r) * p.T + t] = v;
        }
    }
}

template <class SCH, int NWIN>
inline hipError_t launch_resblock_dual_t(int C, const ResblockSplitParams& p, hipStream_t s) {
    const size_t lds = NWIN * ((size_t)SCH::NP * 2 * RBS_W * 32 + 2 * RBD_GUARD);
    auto kern = (C == 16) ? resblock16_dual_kernel<SCH, NWIN> : (C == 64) ? resblock_dual_kernel<SCH, 4, NWIN> : resblock_dual_kernel<SCH, 2, NWIN>;
    static DynLdsOnce lds_once[3];
    const int slot = (C == 16) ? 2 : (C == 64) ? 1 : 0;
    {
        hipError_t e = ensure_dyn_lds(lds_once[slot], reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((p.tiles * p.B + NWIN - 1) / NWIN), dim3(256 * NWIN), lds, s, p);
    return hipGetLastError();
}
// nwin = 2: dual-window anti-phase workgroups; nwin = 1: one window per workgroup (three per CU) with the same lean VALU code
template <class SCH>
inline hipError_t launch_resblock_dual_s(int C, int nwin, const ResblockSplitParams& p, hipStream_t s) {
    return nwin == 2 ? launch_resblock_dual_t<SCH, 2>(C, p, s) : launch_resblock_dual_t<SCH, 1>(C, p, s);
}
inline bool resblock_dual_has(int scheme, int C) { return scheme == SchF16x3::ID && (C == 16 || C == 32 || C == 64); }
hipError_t launch_resblock_dual_f16x3(int C, int nwin, const ResblockSplitParams& p, hipStream_t s);  // csrc/tu_resblock_dual.hip

}  // namespace parrot
