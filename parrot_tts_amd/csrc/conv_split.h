// conv_split.h -- the dilated Conv1d implicit GEMM of conv_mfma.h on the 16-bit matrix pipe (v_mfma_f32_32x32x16_{f16,bf16},
// 16x the rate of v_mfma_f32_32x32x2_f32), fp32 data in and out, fp32 accumulation.  Both fp32 operands are split into
// 16-bit pieces on the fly and a product group is evaluated as a few MFMAs; the SCHEME (template parameter) says how:
//
//   SchF16x3  (PARROT_PREC_F16X3, default)   x = x1 + x2, w = w1 + w2 in fp16 (11 + 11 significand bits);
//             x*w ~= x1w2 + x2w1 + x1w1            3 MFMAs; dropped x2w2 <= 2^-22 |xw|
//             Operands are pre-scaled by powers of two so the second pieces stay normal fp16 numbers:
//             activations by 2^3 (range |x| < 8190, full precision down to |x| ~ 2^-5, absolute floor 2^-28 below),
//             weights by the per-layer power of two that puts max|w| in [2^14, 2^15); the accumulator holds the scaled
//             sum and the epilogue multiplies by the exact inverse.  fp16 has no headroom beyond that: |x| >= 8190 turns
//             into inf/NaN in the output (never a silently wrong finite value).
//   SchBf16x6 (PARROT_PREC_BF16X6)           x = x1 + x2 + x3, w likewise in bf16 (3 x 8 bits), no scaling, fp32's range;
//             x*w ~= x3w1 + x2w2 + x2w1 + x1w3 + x1w2 + x1w1     6 MFMAs; dropped terms <= 2^-23 |xw|
//   SchBf16 / SchF16 (PARROT_PREC_BF16 / _F16)   one rounded piece per operand, ONE MFMA per product group: the reduced
//             precision operating point (BASELINE configs[2] "bf16"); residual stream / accumulators / outputs stay fp32.
//
// Every 16-bit x 16-bit product is exact in fp32 and the accumulator is fp32, so the split schemes carry fp32-class
// error (tests/test_gpu_parity.py holds them to the same tolerances as the exact kernel; measured against fp64 they
// are within 1.5x of it), they are an *evaluation scheme for fp32 data*, not a reduced precision mode.
//
// Data path (CI = 16 input channels per chunk = ONE 32x32x16 MFMA k-step per tap), NP = pieces per operand:
//   * B (activations): staged global -> registers -> (leaky-ReLU, scale, NP-way split) -> LDS as
//     [piece][column][16 channels] (32 B per column), so a lane's 8-channel fragment at any tap is one
//     ds_read_b128 at (column + tap*dilation); double buffered, one barrier per chunk.
//     The fetch is buffer-addressed: out-of-row offsets return 0 (= the zero padding), no per-element address VALU.
//   * A (weights): split on the host and packed as [m_tile][chunk*tap][piece][lane][8 x 16 bit]; streamed from L2 with
//     one buffer_load_dwordx4 per (m_tile, piece) per step; in the straight-line kernels (K > 0) each piece is
//     re-fetched in place for step + 2 right after its last use, in the generic kernel one step ahead.
//   * accumulator init / epilogue are shared with the exact kernel (same 32x32 C/D layout).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "conv_mfma.h"

namespace parrot {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair (x0 in the low half), round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two floats -> packed fp16 pair, round-to-nearest-even (overflow -> inf): one v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pk_f16(float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// v_max_f32 without the canonicalising v_max(v, v) that fmaxf() puts in front of it for a freshly loaded operand
__device__ __forceinline__ float max_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- schemes ----------------------------------------------------------------------------------------------------
// NP pieces per operand, NT MFMA terms; term t multiplies weight piece pa(t) with activation piece pb(t).  Terms are
// ordered by weight piece, highest (smallest) first, so weight piece pc is dead after term last_a(pc) and can be
// re-fetched in place.  XS: power-of-two scale applied to activations before the split (the weight scale is per layer).
struct SchBf16x6 {
    static constexpr int ID = 1, NP = 3, NT = 6;
    static constexpr bool F16 = false;
    static constexpr float XS = 1.f;
    static constexpr int pa(int t) { constexpr int v[6] = {2, 1, 1, 0, 0, 0}; return v[t]; }
    static constexpr int pb(int t) { constexpr int v[6] = {0, 1, 0, 2, 1, 0}; return v[t]; }
    static constexpr int last_a(int pc) { return pc == 2 ? 0 : pc == 1 ? 2 : 5; }
    // (x0, x1) -> three packed bf16 pairs with x ~= p0 + p1 + p2 (each subtraction is exact in fp32)
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&q)[3]) {
        q[0] = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(q[0] << 16), r1 = x1 - __uint_as_float(q[0] & 0xffff0000u);
        q[1] = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(q[1] << 16), s1 = r1 - __uint_as_float(q[1] & 0xffff0000u);
        q[2] = pk_bf16(s0, s1);
    }
};
struct SchF16x3 {
    static constexpr int ID = 2, NP = 2, NT = 3;
    static constexpr bool F16 = true;
    static constexpr float XS = 8.f;
    static constexpr int pa(int t) { constexpr int v[3] = {1, 0, 0}; return v[t]; }
    static constexpr int pb(int t) { constexpr int v[3] = {0, 1, 0}; return v[t]; }
    static constexpr int last_a(int pc) { return pc == 1 ? 0 : 2; }
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&q)[2]) {
        const f32x2 v = {x0, x1};
        q[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
        // r = x - float(hi), exact: one v_fma_mix_f32 per element (fp16 source read straight out of the packed pair) instead of
        // v_cvt_f32_f16 + half a v_pk_add_f32 -- hipcc folds fma(hi, -1, x) back into the subtraction, hence the asm
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(q[0]), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(q[0]), "v"(x1));
        q[1] = pk_f16(r0, r1);
    }
};
struct SchBf16 {
    static constexpr int ID = 3, NP = 1, NT = 1;
    static constexpr bool F16 = false;
    static constexpr float XS = 1.f;
    static constexpr int pa(int) { return 0; }
    static constexpr int pb(int) { return 0; }
    static constexpr int last_a(int) { return 0; }
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&q)[1]) { q[0] = pk_bf16(x0, x1); }
};
struct SchF16 {
    static constexpr int ID = 4, NP = 1, NT = 1;
    static constexpr bool F16 = true;
    static constexpr float XS = 8.f;
    static constexpr int pa(int) { return 0; }
    static constexpr int pb(int) { return 0; }
    static constexpr int last_a(int) { return 0; }
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&q)[1]) { q[0] = pk_f16(x0, x1); }
};

template <class SCH>
__device__ __forceinline__ f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
    if constexpr (SCH::F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <class SCH>
__device__ __forceinline__ f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
    if constexpr (SCH::F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// leaky ReLU (0 <= slope <= 1; slope 1 = none) + the scheme's activation scale: max(s*v, s*slope*v); lrelu(0) = 0 keeps padding
template <class SCH>
__device__ __forceinline__ float pre_scale(float v, float slope) {
    if constexpr (SCH::XS != 1.f) return max_nc(v * SCH::XS, v * (SCH::XS * slope));
    else return max_nc(v, v * slope);
}

// compile-time loop: f(integral_constant<0>) ... f(integral_constant<N-1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// K > 0: the tap count is a compile-time constant and a whole chunk (K steps) is one straight-line block;
// K == 0: any tap count, flat two-step walk with uniform branches at the chunk boundaries.
// SUBS > 1 (1x1 convs only): SUBS 16-channel sub-slabs are staged per barrier and no halo columns at all -- a 1x1 conv
// otherwise pays one barrier, one conversion pass over tile + 64 halo columns and one pipeline bubble per MFMA step.
// LEAN: plain convs only (u == 1): the buffer-addressed prologue / epilogue of conv_mfma.h (a separate instantiation: carrying
// both code paths in one kernel costs registers -- spills on the 2x2 wave tile).
template <class SCH, int WAVES_M, int WAVES_N, int WM, int WN, int MINW, int K, int SUBS = 1, bool LEAN = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_split_kernel(const ConvParams p) {
    static_assert(SUBS == 1 || K == 1, "sub-slab staging: 1x1 convs only (no halo)");
    constexpr int NPC = SCH::NP, NTERM = SCH::NT;
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int COLS = BN + ((SUBS > 1 && K == 1) ? 0 : CONV_HALO);  // staged columns per chunk
    constexpr int PIECE_BYTES = COLS * 32;         // [col][16 ch] 16-bit
    constexpr int SUB_BYTES = NPC * PIECE_BYTES;   // one 16-channel sub-slab
    constexpr int BUF_BYTES = SUBS * SUB_BYTES;
    constexpr int LIVE = SUBS * COLS * 2;          // (sub-slab, channel-octet, column) items per chunk
    constexpr int ITEMS = (LIVE + NT - 1) / NT;    // ... per thread
    constexpr int KS = K * SUBS;                   // MFMA steps per chunk (static path)
    constexpr int STEP_BYTES = NPC * 1024;         // weight stream: [piece][lane][8 x 16 bit] per step

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [2][SUBS][NP][COLS][16] 16-bit

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware block order: workgroup ids go round-robin over the 8 XCDs (each with its own L2; tools/probes: id i runs on
    // XCD i % 8).  Each XCD gets a CONTIGUOUS run of column tiles, walked in order with the M-blocks of a tile back to back:
    // the M-blocks reading the same input slab and the neighbouring tiles re-reading its halo columns find them in that
    // XCD's L2 instead of fetching them from HBM again.
    const int n_mb = (p.M + BM - 1) / BM;
    const int n_tiles = p.tiles_n * p.B, tpx = (n_tiles + 7) >> 3;  // column tiles per XCD
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;          // (workgroup id i runs on XCD i % 8, in id order)
    const int mblock = seq % n_mb;
    const int tile_id = xcd * tpx + seq / n_mb;
    if (tile_id >= n_tiles || seq / n_mb >= tpx) return;
    const int b = tile_id / p.tiles_n;
    const int tn = tile_id - b * p.tiles_n;
    const int t0 = tn * BN;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;  // this row's true input length
    const int W = BN + (p.k - 1) * p.dil;  // columns of the slab this layer can reach
    const int grp = (p.groups > 1) ? (mblock * BM) / p.Mg : 0;
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
        // (idle slots and halo columns beyond the layer's own reach (k-1)*dil read out of range too: zeros, no traffic)
        voff[i] = (item < LIVE && col < W && tin >= 0 && tin < Tlim) ? tin * 4 : (int)0x80000000;
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
            unsigned q[4][NPC];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                SCH::split(pre_scale<SCH>(stage[i][2 * e], slope), pre_scale<SCH>(stage[i][2 * e + 1], slope), q[e]);
            if (item < LIVE) {
                // the two channel octets of a column swap places on odd 8-column groups: with a 32 B column
                // stride this makes every ds_read_b128 lane group hit 16 distinct 16 B slots (no 2-way conflict)
                const int off = sub * SUB_BYTES + col * 32 + ((oct ^ ((col >> 3) & 1)) * 16);
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
                    *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES + off) = uint4{q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
            }
        });
    };

    f32x16 acc[WM][WN];
    const int m_wave = mblock * BM + wm * WM * 32;
    const int n_wave = t0 + wn * WN * 32;
    if constexpr (LEAN) conv_acc_init_lean<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
    else conv_acc_init<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);

    // A stream: n_it = nchunks * k steps per m-tile, STEP_BYTES per step ([piece][lane][8 x 16 bit]).
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
        abase[mt] = __builtin_amdgcn_readfirstlane((int)((mblock * (BM / 32) + wm * WM + mt) * p.n_it * STEP_BYTES));
    const unsigned lane16 = lane * 16;
    const int nsup = p.nchunks / SUBS;  // chunks of 16 * SUBS channels (the host only picks SUBS > 1 when this divides)
    const int colbase = wn * WN * 32 + l31;  // this lane's column at tap 0 (tile nt adds 32*nt: same swizzle bit)
    auto load_b = [&](s16x8 (&bb)[WN][NPC], const char* __restrict__ xs, int tap) {
        const int col = colbase + tap * p.dil;
        const int off = col * 32 + ((half ^ ((col >> 3) & 1)) * 16);
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[nt][pc] = *reinterpret_cast<const s16x8*>(xs + pc * PIECE_BYTES + nt * 1024 + off);
    };
    constexpr int TM = WM * WN, NMF = NTERM * TM;
    if constexpr (K > 0) {
        // ---- static tap count: per chunk, one straight-line block of K steps --------------------------------
        // Weights: two register sets in ping-pong, refilled IN PLACE two steps ahead: the terms are ordered by weight
        // piece, smallest first, so each piece of the current set is dead after its last term and is immediately
        // re-fetched for step + 2 (same set).  Every weight fetch then has 1.5-1.8 steps to land, and -- vmcnt retiring
        // in order -- so do the slab loads issued at the top of the chunk: nothing waits on HBM latency inside a chunk.
        s16x8 A[2][WM][NPC], Bv[2][WN][NPC];
        auto load_a_piece = [&](s16x8 (&a)[WM][NPC], int pc, int soff) {
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
                a[mt][pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, abase[mt] + soff, 0));
        };
        auto next_chunk = [&](int q) { return (q + 1 == nsup) ? 0 : q + 1; };
        int cc = 0, cn = next_chunk(0);
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
            load_a_piece(A[0], pc, cc * KS * STEP_BYTES);
            load_a_piece(A[1], pc, (KS > 1 ? cc * KS + 1 : cn * KS) * STEP_BYTES);
        }
        load_slab(0);
        store_slab(0);
        __syncthreads();
        auto chunk = [&](auto par, int c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            const char* __restrict__ xs = smem_raw + (c & 1) * BUF_BYTES;
            const int cn2 = next_chunk(cn);
            load_slab(cn);  // (after the last chunk: a harmless re-read, stored to the idle buffer)
            load_b(Bv[PAR], xs, 0);
#pragma unroll
            for (int j = 0; j < KS; ++j) {  // step j: sub-slab j / K, tap j % K
                const int cur = (PAR + j) & 1;
                const int tq = (j + 2 < KS) ? cc : (KS == 1 ? cn2 : cn);  // step + 2 in the flat order
                const int tj = (j + 2 < KS) ? j + 2 : (KS == 1 ? 0 : j + 2 - KS);
                const int soff = (tq * KS + tj) * STEP_BYTES;
                if (j + 1 < KS) load_b(Bv[cur ^ 1], xs + ((j + 1) / K) * SUB_BYTES, (j + 1) % K);
#pragma unroll
                for (int t = 0; t < NTERM; ++t) {
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            acc[mt][nt] = mfma32<SCH>(A[cur][mt][SCH::pa(t)], Bv[cur][nt][SCH::pb(t)], acc[mt][nt]);
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        if (t == SCH::last_a(pc)) load_a_piece(A[cur], pc, soff);
                }
            }
            // Issue order of the block (hipcc would otherwise sink every prefetch down to its first use): one
            // memory instruction in the shadow of each MFMA -- the re-fetch of a weight piece right after its last
            // use, the next tap's activation fragments after that, and in the chunk's first step the slab loads.
            constexpr int DS_START = (NPC >= 2) ? (SCH::last_a(NPC - 1) + 1) * TM - 1 + WM : 0;
            constexpr int SLAB_START = (NPC >= 2) ? (SCH::last_a(1) + 1) * TM - 1 + WM : 0;
            constexpr int SLAB_SLOTS = (NMF - SLAB_START) > 0 ? (NMF - SLAB_START) : 1, SLAB_PER = (ITEMS * 8 + SLAB_SLOTS - 1) / SLAB_SLOTS;
#pragma unroll
            for (int i = 0; i < NPC * WN; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // tap 0 fragments
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                int ds_left = (j + 1 < KS) ? NPC * WN : 0, slab_left = (j == 0) ? ITEMS * 8 : 0;
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int pc = NPC - 1; pc >= 1; --pc)
                        if (m >= (SCH::last_a(pc) + 1) * TM - 1 && m < (SCH::last_a(pc) + 1) * TM - 1 + WM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (m >= DS_START && ds_left > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        --ds_left;
                    }
                    if (m >= SLAB_START)
#pragma unroll
                        for (int q = 0; q < SLAB_PER; ++q)
                            if (slab_left > 0) {
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                                --slab_left;
                            }
                }
#pragma unroll
                for (int i = 0; i < WM; ++i) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // piece 0
            }
            __builtin_amdgcn_sched_barrier(0);
            store_slab((c + 1) & 1);
            __syncthreads();
            cc = cn;
            cn = cn2;
        };
        for (int c = 0; c < nsup; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            if (c + 1 < nsup) chunk(std::integral_constant<int, (KS & 1)>{}, c + 1);
        }
    } else {
        // Two operand register sets in ping-pong: the step after the current one is always fetched straight into the
        // OTHER set, so the loop has no register copies.
        s16x8 a0[WM][NPC], b0[WN][NPC], a1[WM][NPC], b1[WN][NPC];
        auto load_a = [&](s16x8 (&a)[WM][NPC], int step) {
            const int soff = __builtin_amdgcn_readfirstlane(step) * STEP_BYTES;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
                    a[mt][pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, abase[mt] + soff, 0));
        };
        auto mfmas = [&](const s16x8 (&a)[WM][NPC], const s16x8 (&bb)[WN][NPC]) {
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma32<SCH>(a[mt][SCH::pa(t)], bb[nt][SCH::pb(t)], acc[mt][nt]);
        };
        load_a(a0, 0);
        load_slab(0);
        store_slab(0);
        __syncthreads();

        // Flat walk over all (chunk, tap) steps, two per iteration: even steps compute from set 0 while set 1 is being
        // filled for the next step, odd steps the other way round.  Chunk boundaries (LDS buffer switch) can fall on
        // either half; the step body handles them with uniform branches.
        int cc = 0, cn = (1 == p.nchunks) ? 0 : 1;  // current / next chunk
        int c = 0, j = 0;                           // chunks done, tap within the chunk
        const char* __restrict__ xs = smem_raw;
        auto step = [&](s16x8 (&xa)[WM][NPC], s16x8 (&xb)[WN][NPC], s16x8 (&ya)[WM][NPC], s16x8 (&yb)[WN][NPC]) {
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
            // issue order inside the step: one prefetch instruction after every MPER MFMAs, so the loads' issue slots and
            // address arithmetic sit in the matrix pipe's shadow instead of in front of the MFMA block
            constexpr int MPER = (NMF / (NPC * (WM + WN))) > 0 ? (NMF / (NPC * (WM + WN))) : 1;
#pragma unroll
            for (int g = 0; g < WM * NPC; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read (weights)
            }
#pragma unroll
            for (int g = 0; g < WN * NPC; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
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
    if constexpr (LEAN) conv_epilogue_lean<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
    else conv_epilogue<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
}

template <class SCH, int WAVES_M, int WAVES_N, int WM, int WN, int MINW, int K, int SUBS = 1, bool LEAN = false>
inline hipError_t launch_conv_split_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    constexpr int BN = WAVES_N * WN * 32;
    const size_t lds = (size_t)2 * SUBS * SCH::NP * (BN + ((SUBS > 1 && K == 1) ? 0 : CONV_HALO)) * 32;
    auto kern = conv_split_kernel<SCH, WAVES_M, WAVES_N, WM, WN, MINW, K, SUBS, LEAN>;
    static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    return hipGetLastError();
}

// split-kernel tile variants: 0 = 128x128 (waves 2x2), 1 = 64x256 (waves 1x4); wave tile 64x64, 2 waves per SIMD;
//                             2 = 128x64 (waves 2x2, wave tile 64x32, 3 waves per SIMD) for 1x1 convs and sequences <= 64;
//                             3 = 32x512 (waves 1x4, wave tile 32x128) for 32-channel layers.
// (Measured and dropped: a 64x128 wave tile at 1 wave per SIMD, -8 %.)
inline void split_tile(int variant, int& bm, int& bn) {
    if (variant == 2) { bm = 128; bn = 64; return; }   // wave tile 64x32, 3 waves per SIMD
    if (variant == 3) { bm = 32; bn = 512; return; }
    bm = (variant & 1) ? 64 : 128;
    bn = (variant & 1) ? 256 : 128;
}
inline unsigned split_grid(const ConvParams& p, int bm) {  // 1-D: 8 XCDs x ceil(tiles / 8) x M-blocks, see the kernel's block order
    const unsigned tiles = (unsigned)p.tiles_n * p.B, n_mb = (p.M + bm - 1) / bm;
    return (tiles + 7) / 8 * 8 * n_mb;
}
// The default scheme's 128 x 64 tile (1x1 convs, short sequences) has its four waves stacked along M, each covering all 64 columns:
// every weight fragment is fetched by one wave instead of two (the vector L1 delivers 64 B/clk, LDS 128 B/clk).  TTE 1x1 convs at
// B = 64: 0.77 -> 0.70 ms per step; one utterance: 0.55 -> 0.51 ms (round-3 A/B; the 2 x 2 grid stays for the other schemes).
// the (tile, tap count) pairs of the synthesis path get the straight-line kernel, anything else the generic one
template <class SCH>
inline hipError_t launch_conv_split_s(int variant, const ConvParams& p, hipStream_t s) {
    int bm, bn;
    split_tile(variant, bm, bn);
    dim3 grid(split_grid(p, bm));
    if (variant == 3) return p.k == 3 ? launch_conv_split_t<SCH, 1, 4, 1, 4, 2, 3>(p, grid, s) : launch_conv_split_t<SCH, 1, 4, 1, 4, 2, 0>(p, grid, s);
    const bool lean = conv_lean_ok(p);  // plain conv: the instantiation with the buffer-addressed prologue / epilogue
    if constexpr (SCH::ID == SchF16x3::ID) if (variant == 2) {  // the 128 x 64 tile with its four waves stacked along M (default scheme only: build time)
        if (p.k == 1 && p.nchunks % 4 == 0)
            return lean ? launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 1, 4, true>(p, grid, s) : launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 1, 4>(p, grid, s);
        switch (p.k) {
            case 1: return launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 1>(p, grid, s);
            case 3: return launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 3>(p, grid, s);
            case 9: return launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 9>(p, grid, s);
            default: return launch_conv_split_t<SCH, 4, 1, 1, 2, 3, 0>(p, grid, s);
        }
    }
    if constexpr (SCH::ID != SchF16x3::ID) if (variant == 2) {
        if (p.k == 1 && p.nchunks % 4 == 0)  // 64 channels per barrier
            return lean ? launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 1, 4, true>(p, grid, s) : launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 1, 4>(p, grid, s);
        switch (p.k) {  // (k > 1: sequences of <= 64 steps, i.e. the TTE encoder side)
            case 1: return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 1>(p, grid, s);
            case 3: return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 3>(p, grid, s);
            case 9: return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 9>(p, grid, s);
            default: return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 0>(p, grid, s);
        }
    }
    if (variant & 1) switch (p.k) {
            case 1: return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 1>(p, grid, s);
            case 3: return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 3>(p, grid, s);
            case 7: return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 7>(p, grid, s);
            case 11: return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 11>(p, grid, s);
            default: return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 0>(p, grid, s);
        }
    switch (p.k) {
        case 3:
            return lean ? launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 3, 1, true>(p, grid, s) : launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 3>(p, grid, s);
        case 7: return launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 7>(p, grid, s);
        case 9: return launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 9>(p, grid, s);
        case 11: return launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 11>(p, grid, s);
        default: return launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 0>(p, grid, s);
    }
}
// The single-piece schemes (reduced precision operating point) only get the generic kernels: they are a companion
// mode, and every (scheme, tile, tap count) instantiation costs build time.
template <class SCH>
inline hipError_t launch_conv_split_generic(int variant, const ConvParams& p, hipStream_t s) {
    int bm, bn;
    split_tile(variant, bm, bn);
    dim3 grid(split_grid(p, bm));
    if (variant == 3) return launch_conv_split_t<SCH, 1, 4, 1, 4, 2, 0>(p, grid, s);
    if (variant == 2) {
        if (p.k == 1 && p.nchunks % 4 == 0) return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 1, 4>(p, grid, s);
        return launch_conv_split_t<SCH, 2, 2, 2, 1, 3, 0>(p, grid, s);
    }
    if (variant & 1) return launch_conv_split_t<SCH, 1, 4, 2, 2, 2, 0>(p, grid, s);
    return launch_conv_split_t<SCH, 2, 2, 2, 2, 2, 0>(p, grid, s);
}
// Per-scheme entry points, each defined in its own translation unit (csrc/tu_*.hip) so the library builds in parallel.
hipError_t launch_conv_split_f16x3(int variant, const ConvParams& p, hipStream_t s);
hipError_t launch_conv_split_bf16x6(int variant, const ConvParams& p, hipStream_t s);
hipError_t launch_conv_split_bf16(int variant, const ConvParams& p, hipStream_t s);
hipError_t launch_conv_split_f16(int variant, const ConvParams& p, hipStream_t s);
inline hipError_t launch_conv_split(int scheme, int variant, const ConvParams& p, hipStream_t s) {
    switch (scheme) {
        case SchBf16x6::ID: return launch_conv_split_bf16x6(variant, p, s);
        case SchF16x3::ID: return launch_conv_split_f16x3(variant, p, s);
        case SchBf16::ID: return launch_conv_split_bf16(variant, p, s);
        case SchF16::ID: return launch_conv_split_f16(variant, p, s);
        default: return hipErrorInvalidValue;
    }
}
inline int scheme_pieces(int scheme) { return scheme == SchBf16x6::ID ? 3 : scheme == SchF16x3::ID ? 2 : 1; }
inline bool scheme_is_f16(int scheme) { return scheme == SchF16x3::ID || scheme == SchF16::ID; }
inline float scheme_xs(int scheme) { return scheme_is_f16(scheme) ? 8.f : 1.f; }

}  // namespace parrot
