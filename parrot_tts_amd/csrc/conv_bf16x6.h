// conv_bf16x6.h -- the same dilated Conv1d implicit GEMM as conv_mfma.h, with every fp32 operand split into
// three bf16 pieces and the product evaluated with six bf16 MFMAs into an fp32 accumulator:
//
//     x = x1 + x2 + x3,  w = w1 + w2 + w3   (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 mantissa bits)
//     x*w ~= x1w1 + x1w2 + x2w1 + x2w2 + x1w3 + x3w1       (dropped: x2w3, x3w2, x3w3 <= 2^-23 |xw|)
//
// Every bf16 x bf16 product is exact in fp32 and the accumulator is fp32, so the result carries fp32-class error
// (the dropped terms are below one fp32 ulp of each product) while running on v_mfma_f32_32x32x16_bf16 at
// 16/6 = 2.67x the rate of v_mfma_f32_32x32x2_f32.  It is an *evaluation scheme for fp32 data*, not a reduced
// precision mode: inputs, outputs, residual stream and accumulation stay fp32; the parity tests hold it to the
// same tolerances as the exact kernel.  bf16 has fp32's exponent range, so no scaling / overflow handling is needed.
//
// Data path (CI = 16 input channels per chunk = ONE 32x32x16 MFMA k-step per tap):
//   * B (activations): staged global -> registers -> (leaky-ReLU, zero pad, 3-way split) -> LDS as
//     [piece][column][16 channels] bf16 (32 B per column), so a lane's 8-channel fragment at any tap is one
//     ds_read_b128 at (column + tap*dilation); double buffered, one barrier per chunk.
//     The fetch is buffer-addressed: out-of-row offsets return 0 (= the zero padding), no per-element address VALU.
//   * A (weights): split on the host and packed as [m_tile][chunk*tap][piece][lane][8 bf16]; streamed from L2 with
//     one buffer_load_dwordx4 per (m_tile, piece) per step; in the straight-line kernels (K > 0) each piece is
//     re-fetched in place for step + 2 right after its last use, in the generic kernel one step ahead.
//   * accumulator init / epilogue are shared with the exact kernel (same 32x32 C/D layout).
// The kernels are power-limited on random data: the same schedule runs 36 % faster on constant
// operands, so further issue-slot tuning does not pay.
#pragma once
#include <type_traits>

#include "conv_mfma.h"

namespace parrot {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair (x0 in the low half), round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// v_max_f32 without the canonicalising v_max(v, v) that fmaxf() puts in front of it for a freshly loaded operand
__device__ __forceinline__ float max_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x0, x1) -> three packed bf16 pairs with x ~= p0 + p1 + p2 (each subtraction is exact in fp32)
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(p1 << 16), s1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(s0, s1);
}

// K > 0: the tap count is a compile-time constant and a whole chunk (K steps) is one straight-line block;
// K == 0: any tap count, flat two-step walk with uniform branches at the chunk boundaries.
// compile-time loop: f(integral_constant<0>) ... f(integral_constant<N-1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// SUBS > 1 (1x1 convs only): SUBS 16-channel sub-slabs are staged per barrier and no halo columns at all -- a 1x1 conv
// otherwise pays one barrier, one conversion pass over tile + 64 halo columns and one pipeline bubble per MFMA step.
template <int WAVES_M, int WAVES_N, int WM, int WN, int MINW, int K, int SUBS = 1>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_bf16x6_kernel(const ConvParams p) {
    static_assert(SUBS == 1 || K == 1, "sub-slab staging is for 1x1 convs");
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int COLS = BN + (SUBS > 1 ? 0 : CONV_HALO);  // staged columns per chunk
    constexpr int PIECE_BYTES = COLS * 32;         // [col][16 ch] bf16
    constexpr int SUB_BYTES = 3 * PIECE_BYTES;     // one 16-channel sub-slab
    constexpr int BUF_BYTES = SUBS * SUB_BYTES;
    constexpr int LIVE = SUBS * COLS * 2;          // (sub-slab, channel-octet, column) items per chunk
    constexpr int ITEMS = (LIVE + NT - 1) / NT;    // ... per thread
    constexpr int KS = K * SUBS;                   // MFMA steps per chunk (static path)

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [2][3][COLS][16] bf16

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int b = blockIdx.x / p.tiles_n;
    const int tn = blockIdx.x - b * p.tiles_n;
    const int t0 = tn * BN;
    const int W = BN + (p.k - 1) * p.dil;
    const int Tlim = p.row_len ? min(p.Tin, p.row_len[b] * p.row_len_mul) : p.Tin;  // this row's true input length
    const int grp = (p.groups > 1) ? (blockIdx.y * BM) / p.Mg : 0;
    const float* __restrict__ xb = p.x + (size_t)b * p.x_bstride + (size_t)grp * p.Cin * p.Tin;

    // Slab fetch: one buffer descriptor over this batch row's (group's) input.  The per-lane offset is the time index
    // (computed once per kernel; padding / past-the-row columns get an offset beyond num_records, for which the
    // load returns 0 = the conv's zero padding), the channel goes into the wave-uniform scalar offset: no per-element
    // address or predicate VALU work at all.  (COLS is a multiple of 64, so the channel octet of an item is
    // wave-uniform; the host only selects this kernel when Cin % 16 == 0, 0 <= slope <= 1 and a row is < 2 GiB.)
    static_assert(COLS % 64 == 0, "channel octet must be wave-uniform");
    const size_t xaddr = reinterpret_cast<size_t>(xb);
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((size_t)x_hi << 32) | x_lo), 0, 0x7fffffff, 0x00020000);
    const int row_bytes = p.Tin * 4;
    const float slope = (p.pre == PRE_LRELU) ? p.pre_slope : 1.f;
    int voff[ITEMS], soct[ITEMS];
    static_for<ITEMS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int item = tid + NT * i;
        const int so = item / COLS;  // sub-slab * 2 + channel octet (wave-uniform)
        const int col = item - so * COLS;
        const int tin = t0 - p.pad_left + col;
        voff[i] = (item < LIVE && tin >= 0 && tin < Tlim) ? tin * 4 : (int)0x80000000;  // (idle slots read out of range too)
        soct[i] = __builtin_amdgcn_readfirstlane(min(so, 2 * SUBS - 1) * 8) * row_bytes;
    });
    float stage[ITEMS][8];
    auto load_slab = [&](int c) __attribute__((always_inline)) {
        const int cbase = c * (16 * SUBS) * row_bytes;
        static_for<ITEMS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                stage[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff[i], cbase + soct[i] + e * row_bytes, 0));
        });
    };
    auto store_slab = [&](int buf) __attribute__((always_inline)) {
        char* dst = smem_raw + buf * BUF_BYTES;
        static_for<ITEMS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int item = tid + NT * i;
            const int so = item / COLS, sub = so >> 1, oct = so & 1;
            const int col = item - so * COLS;
            unsigned q0[4], q1[4], q2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // leaky ReLU for 0 <= slope <= 1 (slope 1 = none) is max(v, slope * v); lrelu(0) = 0 keeps the padding
                const float v0 = stage[i][2 * e], v1 = stage[i][2 * e + 1];
                split3_pk(max_nc(v0, v0 * slope), max_nc(v1, v1 * slope), q0[e], q1[e], q2[e]);
            }
            if (item < LIVE) {
                const uint4 v0 = {q0[0], q0[1], q0[2], q0[3]}, v1 = {q1[0], q1[1], q1[2], q1[3]}, v2 = {q2[0], q2[1], q2[2], q2[3]};
                // the two channel octets of a column swap places on odd 8-column groups: with a 32 B column
                // stride this makes every ds_read_b128 lane group hit 16 distinct 16 B slots (no 2-way conflict)
                const int off = sub * SUB_BYTES + col * 32 + ((oct ^ ((col >> 3) & 1)) * 16);
                *reinterpret_cast<uint4*>(dst + off) = v0;
                *reinterpret_cast<uint4*>(dst + PIECE_BYTES + off) = v1;
                *reinterpret_cast<uint4*>(dst + 2 * PIECE_BYTES + off) = v2;
            }
        });
    };

    f32x16 acc[WM][WN];
    const int m_wave = blockIdx.y * BM + wm * WM * 32;
    const int n_wave = t0 + wn * WN * 32;
    conv_acc_init<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);

    // A stream: n_it = nchunks * k steps per m-tile, 3 KiB per step ([piece][lane][8 bf16]).
    // Weight stream addressing: one buffer descriptor (SGPR quad) for the whole packed stream, a constant per-lane byte
    // offset (voffset) and a wave-uniform scalar offset per (m-tile, step) (soffset): each fetch is
    // `buffer_load_dwordx4 v, v_lane, s[rsrc], s_off offen` -- no 64-bit VALU address math per step and no
    // load-destination registers recycled as address temporaries.
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wfrag));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wfrag) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    int abase[WM];  // byte offset of each m-tile's stream (host guarantees the packed stream is < 2 GiB)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
        abase[mt] = __builtin_amdgcn_readfirstlane((int)((blockIdx.y * (BM / 32) + wm * WM + mt) * p.n_it * 3072));
    const unsigned lane16 = lane * 16;
    const int nsup = p.nchunks / SUBS;  // chunks of 16 * SUBS channels (the host only picks SUBS > 1 when this divides)
    // (A per-tile rotation of the chunk order, meant to spread neighbouring workgroups over different L2 lines, measured
    //  +2 % before the streams were buffer-addressed and -0.8 % after -- co-resident workgroups now share weight lines in
    //  L1 -- so every workgroup walks the chunks from 0; the loop below still supports any starting chunk.)
    constexpr int rot = 0;  // (% runs on the VALU: pin the result to an SGPR)
    // Two operand register sets in ping-pong: the step after the current one is always fetched straight into the
    // OTHER set, so the loop has no register copies (48 v_mov per step cost as much issue time as the 24 MFMAs).
    s16x8 a0[WM][3], b0[WN][3], a1[WM][3], b1[WN][3];
    auto load_a = [&](s16x8 (&a)[WM][3], int step) {
        const int soff = __builtin_amdgcn_readfirstlane(step) * 3072;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                a[mt][pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, abase[mt] + soff, 0));
    };
    const int colbase = wn * WN * 32 + l31;  // this lane's column at tap 0 (tile nt adds 32*nt: same swizzle bit)
    auto load_b = [&](s16x8 (&bb)[WN][3], const char* __restrict__ xs, int tap) {
        const int col = colbase + tap * p.dil;
        const int off = col * 32 + ((half ^ ((col >> 3) & 1)) * 16);
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bb[nt][pc] = *reinterpret_cast<const s16x8*>(xs + pc * PIECE_BYTES + nt * 1024 + off);
    };
    auto mfmas = [&](const s16x8 (&a)[WM][3], const s16x8 (&bb)[WN][3]) {
        // six terms, smallest first: x3w1, x1w3, x2w2, x2w1, x1w2, x1w1 (a = weight pieces, bb = activation pieces)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mt][PA[t]]),
                                                                          __builtin_bit_cast(bf16x8, bb[nt][PB[t]]), acc[mt][nt], 0, 0, 0);
    };
    if constexpr (K > 0) {
        // ---- static tap count: per chunk, one straight-line block of K steps --------------------------------
        // Weights: two register sets in ping-pong, refilled IN PLACE two steps ahead: the six terms are ordered by
        // weight piece (3rd, 2nd, 2nd, 1st, 1st, 1st), so piece 3 of the current set is dead after the first
        // term, piece 2 after the third; each dead piece is immediately re-fetched for step + 2 (same set).  Every
        // weight fetch then has 1.5-1.8 steps to land, and -- vmcnt retiring in order -- so do the slab loads issued
        // at the top of the chunk: nothing waits on HBM latency inside a chunk.
        s16x8 A[2][WM][3], Bv[2][WN][3];
        constexpr int PA[6] = {2, 1, 1, 0, 0, 0};
        constexpr int PB[6] = {0, 1, 0, 2, 1, 0};
        auto load_a_piece = [&](s16x8 (&a)[WM][3], int pc, int soff) {
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
                a[mt][pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, abase[mt] + soff, 0));
        };
        auto next_chunk = [&](int q) { return (q + 1 == nsup) ? 0 : q + 1; };
        int cc = rot, cn = next_chunk(rot);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            load_a_piece(A[0], pc, cc * KS * 3072);
            load_a_piece(A[1], pc, (KS > 1 ? cc * KS + 1 : cn * KS) * 3072);
        }
        load_slab(rot);
        store_slab(0);
        __syncthreads();
        auto chunk = [&](auto par, int c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            const char* __restrict__ xs = smem_raw + (c & 1) * BUF_BYTES;
            const int cn2 = next_chunk(cn);
#ifndef EXP_NO_SLABLOAD
            load_slab(cn);  // (after the last chunk: a harmless re-read, stored to the idle buffer)
#endif
            load_b(Bv[PAR], xs, 0);
#pragma unroll
            for (int j = 0; j < KS; ++j) {  // step j: sub-slab j / K, tap j % K
                const int cur = (PAR + j) & 1;
                const int tq = (j + 2 < KS) ? cc : (KS == 1 ? cn2 : cn);  // step + 2 in the flat order
                const int tj = (j + 2 < KS) ? j + 2 : (KS == 1 ? 0 : j + 2 - KS);
#ifdef EXP_A_CONST
                const int soff = 0; (void)tq; (void)tj;
#else
                const int soff = (tq * KS + tj) * 3072;
#endif
                if (j + 1 < KS) load_b(Bv[cur ^ 1], xs + ((j + 1) / K) * SUB_BYTES, (j + 1) % K);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[cur][mt][PA[t]]),
                                                                                  __builtin_bit_cast(bf16x8, Bv[cur][nt][PB[t]]), acc[mt][nt], 0, 0, 0);
                    if (t == 0) load_a_piece(A[cur], 2, soff);
                    if (t == 2) load_a_piece(A[cur], 1, soff);
                    if (t == 5) load_a_piece(A[cur], 0, soff);
                }
            }
            // Issue order of the block (hipcc would otherwise sink every prefetch down to its first use): one
            // memory instruction in the shadow of each MFMA -- the re-fetch of a weight piece right after its last
            // use, the next tap's activation fragments after that, and in the chunk's first step the slab loads.
            constexpr int TM = WM * WN, NMF = 6 * TM;
            constexpr int SLAB_SLOTS = NMF - (3 * TM - 1 + WM), SLAB_PER = (ITEMS * 8 + SLAB_SLOTS - 1) / SLAB_SLOTS;
#pragma unroll
            for (int i = 0; i < 3 * WN; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // tap 0 fragments
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                int ds_left = (j + 1 < KS) ? 3 * WN : 0, slab_left = (j == 0) ? ITEMS * 8 : 0;
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m >= TM - 1 && m < TM - 1 + WM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // piece 3
                    if (m >= 3 * TM - 1 && m < 3 * TM - 1 + WM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // piece 2
                    if (m >= TM - 1 + WM && ds_left > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        --ds_left;
                    }
                    if (m >= 3 * TM - 1 + WM)
#pragma unroll
                        for (int q = 0; q < SLAB_PER; ++q)
                            if (slab_left > 0) {
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                                --slab_left;
                            }
                }
#pragma unroll
                for (int i = 0; i < WM; ++i) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // piece 1
            }
            __builtin_amdgcn_sched_barrier(0);
#ifndef EXP_NO_STORE
            store_slab((c + 1) & 1);
#endif
#ifndef EXP_NO_BARRIER
            __syncthreads();
#endif
            cc = cn;
            cn = cn2;
        };
        for (int c = 0; c < nsup; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            if (c + 1 < nsup) chunk(std::integral_constant<int, (KS & 1)>{}, c + 1);
        }
    } else {
        load_a(a0, rot * p.k);
        load_slab(rot);
        store_slab(0);
        __syncthreads();

        // Flat walk over all (chunk, tap) steps, two per iteration: even steps compute from set 0 while set 1 is being
        // filled for the next step, odd steps the other way round.  Chunk boundaries (LDS buffer switch) can fall on
        // either half; the step body handles them with uniform branches.
        int cc = rot, cn = (rot + 1 == p.nchunks) ? 0 : rot + 1;  // current / next chunk in rotated order
        int c = 0, j = 0;                                           // chunks done, tap within the chunk
        const char* __restrict__ xs = smem_raw;
        auto step = [&](s16x8 (&xa)[WM][3], s16x8 (&xb)[WN][3], s16x8 (&ya)[WM][3], s16x8 (&yb)[WN][3]) {
            if (j == 0) {  // first tap of a chunk: its slab is in LDS (barrier passed); start fetching the next one
                if (c + 1 < p.nchunks) load_slab(cn);
                load_b(xb, xs, 0);
            }
            const bool last_tap = (j + 1 == p.k);
            // both fetches are unconditional: a branch here makes hipcc merge wait-count states and stall every MFMA
            // block on the reads issued just before it.  On a chunk's last tap the B fetch reads in-slab values that
            // are discarded (step() reloads B after the barrier); after the very last step A re-reads a valid step.
            load_a(ya, last_tap ? cn * p.k : cc * p.k + j + 1);
            load_b(yb, xs, j + 1);
            mfmas(xa, xb);
            // issue order inside the step: one prefetch instruction after every MFMA pair, so the loads' issue slots and
            // address arithmetic sit in the matrix pipe's shadow instead of in front of the MFMA block
    #pragma unroll
            for (int g = 0; g < WM * 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (weights)
            }
    #pragma unroll
            for (int g = 0; g < WN * 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read (activations)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (last_tap) {
                if (c + 1 < p.nchunks) store_slab((c + 1) & 1);
                __syncthreads();
                ++c;
                xs = smem_raw + (c & 1) * BUF_BYTES;
                cc = cn;
                cn = (cc + 1 == p.nchunks) ? 0 : cc + 1;
                j = 0;
            } else {
                ++j;
            }
        };
        const int total = p.nchunks * p.k;
        for (int st = 0; st < total; st += 2) {
            step(a0, b0, a1, b1);
            if (st + 1 < total) step(a1, b1, a0, b0);
        }
    }
    conv_epilogue<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int MINW, int K, int SUBS = 1>
inline hipError_t launch_conv_bf16x6_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    constexpr int BN = WAVES_N * WN * 32;
    const size_t lds = (size_t)2 * SUBS * 3 * (BN + (SUBS > 1 ? 0 : CONV_HALO)) * 32;
    auto kern = conv_bf16x6_kernel<WAVES_M, WAVES_N, WM, WN, MINW, K, SUBS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    return hipGetLastError();
}

// bf16x6 tile variants: 0 = 128x128 (waves 2x2), 1 = 64x256 (waves 1x4); wave tile 64x64, 2 waves per SIMD;
//                      2 = 128x64 (waves 2x2, wave tile 64x32, 3 waves per SIMD) for 1x1 convs and sequences <= 64;
//                      3 = 32x512 (waves 1x4, wave tile 32x128) for 32-channel layers.
// (Measured and dropped: a 64x128 wave tile at 1 wave per SIMD, -8 %.)
inline void bf16x6_tile(int variant, int& bm, int& bn) {
    if (variant == 2) { bm = 128; bn = 64; return; }   // wave tile 64x32, 3 waves per SIMD
    if (variant == 3) { bm = 32; bn = 512; return; }
    bm = (variant & 1) ? 64 : 128;
    bn = (variant & 1) ? 256 : 128;
}
// the (tile, tap count) pairs of the synthesis path get the straight-line kernel, anything else the generic one
inline hipError_t launch_conv_bf16x6(int variant, const ConvParams& p, hipStream_t s) {
    int bm, bn;
    bf16x6_tile(variant, bm, bn);
    dim3 grid(p.tiles_n * p.B, (p.M + bm - 1) / bm);
    if (variant == 3) return p.k == 3 ? launch_conv_bf16x6_t<1, 4, 1, 4, 2, 3>(p, grid, s) : launch_conv_bf16x6_t<1, 4, 1, 4, 2, 0>(p, grid, s);
    if (variant == 2) {
        if (p.k == 1 && p.nchunks % 4 == 0) return launch_conv_bf16x6_t<2, 2, 2, 1, 3, 1, 4>(p, grid, s);  // 64 channels per barrier
        switch (p.k) {  // (k > 1: sequences of <= 64 steps, i.e. the TTE encoder side)
            case 1: return launch_conv_bf16x6_t<2, 2, 2, 1, 3, 1>(p, grid, s);
            case 3: return launch_conv_bf16x6_t<2, 2, 2, 1, 3, 3>(p, grid, s);
            case 9: return launch_conv_bf16x6_t<2, 2, 2, 1, 3, 9>(p, grid, s);
            default: return launch_conv_bf16x6_t<2, 2, 2, 1, 3, 0>(p, grid, s);
        }
    }
    if (variant & 1) switch (p.k) {
            case 1: return launch_conv_bf16x6_t<1, 4, 2, 2, 2, 1>(p, grid, s);
            case 3: return launch_conv_bf16x6_t<1, 4, 2, 2, 2, 3>(p, grid, s);
            case 7: return launch_conv_bf16x6_t<1, 4, 2, 2, 2, 7>(p, grid, s);
            case 11: return launch_conv_bf16x6_t<1, 4, 2, 2, 2, 11>(p, grid, s);
            default: return launch_conv_bf16x6_t<1, 4, 2, 2, 2, 0>(p, grid, s);
        }
    switch (p.k) {
        case 3: return launch_conv_bf16x6_t<2, 2, 2, 2, 2, 3>(p, grid, s);
        case 7: return launch_conv_bf16x6_t<2, 2, 2, 2, 2, 7>(p, grid, s);
        case 9: return launch_conv_bf16x6_t<2, 2, 2, 2, 2, 9>(p, grid, s);
        case 11: return launch_conv_bf16x6_t<2, 2, 2, 2, 2, 11>(p, grid, s);
        default: return launch_conv_bf16x6_t<2, 2, 2, 2, 2, 0>(p, grid, s);
    }
}

}  // namespace parrot
