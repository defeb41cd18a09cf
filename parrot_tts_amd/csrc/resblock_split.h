// resblock_split.h -- HiFi-GAN ResBlock1 layer pairs (reference utils/vocoder/models.py:31-38) for the 16-, 32- and
// 64-channel stages, fused per launch and evaluated with the split schemes of conv_split.h (fp32 data, a few 16-bit
// MFMAs per product group, fp32 accumulate; template parameter SCH: 3 fp16 MFMAs / 6 bf16 MFMAs / 1 MFMA).
// (Written below for three bf16 pieces; NP = SCH::NP pieces in general, pre-scaled operands for the fp16 schemes:
//  the accumulator of conv m holds XS * wsc[m] * sum and is scaled back when it is consumed.)
//
// Layer by layer the 32-channel stage is bandwidth bound (every conv streams (B, 32, T) in, the residual in and
// the result out: 3 passes per conv, 18 convs).  Here a workgroup owns a window of W = 384 columns:
//
//   R  (running residual, fp32)  : REGISTERS, in the MFMA C/D layout -- wave w holds columns [96w, 96w+96), all 32
//                                   channels (3 tiles of 32x32); the residual add is register + register.
//   P  (conv operand, 3 x bf16)  : LDS, [piece][chunk][column][16 channels] = the B-fragment layout of
//                                   conv_bf16x6.h; written straight from the accumulators (leaky ReLU, sequence
//                                   mask, 3-way split), read with one ds_read_b128 per fragment.
//
//   per pair (dilation d):   P = split(lrelu(R));  h = conv_d(P) + b1;  P = split(lrelu(h));  R = conv_1(P) + b2 + R
//
// Each conv shrinks the valid region by its own reach, so after the launch's last conv the central W - 2H columns
// are valid and are written out (columns outside are garbage that never feeds a valid one; fragment reads are
// clamped to the window).  Positions outside the true sequence [0, T) are forced to zero after every conv: that IS
// the reference's per-layer zero padding.  One operand buffer suffices (R never lives in LDS), at two barriers per
// conv; 72 KiB of LDS and <= 256 VGPRs let two workgroups share a CU, so one's conversion phase overlaps the
// other's MFMA phase.
//
// Weights: the conv plans' split streams ([step][piece][lane][8 bf16], step = chunk * K + tap), concatenated in conv
// order into ONE stream per ResBlock, so the in-place two-steps-ahead weight prefetch (see conv_bf16x6.h) runs
// straight across conv boundaries.
#pragma once
#include "conv_split.h"

namespace parrot {

constexpr int RBS_MAX_CONVS = 8;    // convs of one ResBlock (pairs x 2)
constexpr int RBS_MAX_BRANCH = 4;   // MRF branches of a stage (PARROT_MAX_KERNELS)
constexpr int RBS_MAX_ALL = RBS_MAX_CONVS * RBS_MAX_BRANCH;

// Phase trace (experiment builds only, tools/build_exp.sh trace -DRBS_TRACE): one workgroup of the 32-channel kernels stamps the
// shader clock (s_memtime) at every phase boundary -- [wave][stamp] in a device array read back by parrot_debug_rbs_trace.
#ifdef RBS_TRACE
constexpr int RBS_TRACE_STAMPS = 128;
__device__ unsigned long long g_rbs_trace[8 * RBS_TRACE_STAMPS];
#define RBS_STAMP()                                                                                                   \
    do {                                                                                                              \
        if (trace_on && trace_n < RBS_TRACE_STAMPS) {                                                                 \
            const unsigned long long tt = __builtin_amdgcn_s_memtime();                                               \
            if (lane == 0) g_rbs_trace[wave * RBS_TRACE_STAMPS + trace_n] = tt;                                       \
        }                                                                                                             \
        ++trace_n;                                                                                                    \
    } while (0)
#else
#define RBS_STAMP() do {} while (0)
#endif
// 1 (default): a wave converts the columns no other wave reads BEFORE the barrier that ends a conv (see write_p); 0: round-4 order
#ifndef RBS_EARLY_INNER
#define RBS_EARLY_INNER 1
#endif

struct ResblockSplitParams {
    const float* x;           // (B, C, T) input of the first pair of this launch (C = 32 or 16: the kernel's)
    float* y;                 // (B, C, T) output (MRF accumulator or an intermediate buffer)
    const uint16_t* wstream;  // this launch's first step in the ResBlock's concatenated weight stream (padded past the end for the prefetch)
    const float* bias[RBS_MAX_ALL];
    float wsc[RBS_MAX_ALL];  // per-conv weight scale of the stream (power of two; 1 for the bf16 schemes)
    int dil[RBS_MAX_ALL];
    int n_conv;               // convs in this launch (even: whole pairs); whole-MRF launches: convs per branch
    int k;                    // taps of every conv of the block
    // whole-MRF launches (the MRF instantiations): every ResBlock branch of the stage in ONE launch -- conv m of branch j is entry
    // j * n_conv + m of the arrays above, branch j walks its own weight stream with its own tap count; y = sum_j branch_j(x) / div
    int early;                // every conv's reach (k - 1) / 2 * dil is <= 32 columns: the inner tiles may be rewritten before the barrier (write_p)
    int n_branch;
    const uint16_t* bstream[RBS_MAX_BRANCH];
    int bk[RBS_MAX_BRANCH];
    int T, B;
    int H;                    // total reach of this launch's convs
    int TT;                   // output columns per workgroup = W - 2H
    int tiles;                // ceil(T / TT)
    int epi;                  // EPI_STORE / EPI_ADD / EPI_ADD_DIV on y
    float div;
    float slope;
    const int32_t* row_len;
    int row_len_mul;
    int row_len_add;         // true length of a row of n > 0 units at this layer = n * row_len_mul + row_len_add (odd k - u upsampling stages add samples)
};

constexpr int RBS_W = 384;  // 32-channel window (the 64-channel kernel uses half: the same 72 KiB of LDS)

// NCH = channel chunks of 16: 2 -> 32 channels, 4 waves side by side (window 384);
//                              4 -> 64 channels, 2 (row tiles) x 2 (column halves) waves (window 192);
//                              8 -> 128 channels, 4 row tiles, one column group (window 96);
//                             16 -> 256 channels, 8 waves = 8 row tiles, one column group (window 96, 96 KiB of LDS: one workgroup per CU).
// WN = column groups (waves side by side; 0 = the pair kernels' own: 4 waves per workgroup up to 128 channels).  MRF = whole-stage
// launches: all ResBlock branches of an MRF stage per window, the branch sum in registers (x is read once per branch -- from L2
// after the first -- and y written ONCE per stage instead of three launches that each re-read x and read-modify-write the sum).
// The MRF instantiation (32 channels) runs 8 waves on a twice-as-wide window (768 columns; 96 KiB of LDS, one workgroup per CU,
// <= 256 VGPRs): 18 convs per window amortise the window's load / store, which a single resident workgroup cannot hide behind
// another's MFMAs, and the total reach of the k = 11 branch (60 columns) still leaves 84 % of a window.  (The 64-channel
// instantiation -- 384 columns, 69 % -- was built and measured slower than pair launches + layer kernels: profiles/r05b_mrf_ab.jsonl.)
constexpr int rbs_default_wn(int nch) { return (nch / 2 >= 4) ? 1 : 4 / (nch / 2); }
template <class SCH, int NCH, int WN = 0, bool MRF = false>
__global__ __launch_bounds__((NCH / 2) * (WN ? WN : rbs_default_wn(NCH)) * 64,
                             ((NCH / 2) * (WN ? WN : rbs_default_wn(NCH)) * 64 == 512) ? 1 : (SCH::NP == 3 ? 2 : 3)) void resblock_split_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT, STEP_BYTES = SCH::NP * 1024;
    constexpr int C = 16 * NCH, WAVES_M = NCH / 2, WAVES_N = WN ? WN : rbs_default_wn(NCH), NTW = 3, W = WAVES_N * NTW * 32;
    // (with several row tiles the waves of a column group read each other's channels: 32 channels only)
    const bool early = (WAVES_M == 1) && RBS_EARLY_INNER && __builtin_amdgcn_readfirstlane(p.early) != 0;
    int K = MRF ? p.bk[0] : p.k;
    // operand buffer: [piece][chunk][octet][col][8 channels]: 16 bytes per (column, octet), columns contiguous -- both the
    // 16-byte stores of write_p and the ds_read_b128 fragment reads at any tap shift walk consecutive addresses across lanes
    constexpr int OCT_BYTES = W * 16, CH_BYTES = 2 * OCT_BYTES, PIECE_BYTES = NCH * CH_BYTES;
    static_assert(NCH == 2 || NCH == 4 || NCH == 8 || NCH == 16, "32, 64, 128 channels (4 waves: NCH / 2 row tiles x 4 / (NCH / 2) column groups) or 256 (8 waves)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // NP * PIECE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;  // row tile (32 output channels) of this wave
    const int wn = wave % WAVES_N;  // column group (96 columns)
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int b = blockIdx.x / p.tiles;
    const int tile = blockIdx.x - b * p.tiles;
    const int t_base = tile * p.TT - p.H;  // sequence position of window column 0
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    const bool edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);  // window not fully inside the row
#ifdef RBS_TRACE
    const bool trace_on = NCH == 2 && blockIdx.x == gridDim.x / 2;  // one mid-launch workgroup of the 32-channel kernels
    int trace_n = 0;
    RBS_STAMP();  // 0: start
#endif

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
    auto load_r = [&]() __attribute__((always_inline)) {
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
    };
    load_r();

    // ---- accumulator tile -> operand buffer: (leaky ReLU,) sequence mask, 3-way split, two 8-byte stores per
    //      (tile, chunk, piece, row group): rows 8q+4g+{0..3} of this half are channels 16 (2 wm + q) + 8g + 4 half + {0..3}
    // `mul`: (power-of-two) factor taking v to the operand scale XS (v may be a scaled accumulator)
    // `part`: 0 = every tile; 1 = the INNER tile only (columns 32..63 of the wave's 96: with one row tile per workgroup -- 32 channels --
    // no other wave reads them, every conv's reach being <= 25 columns, so they may be rewritten as soon as THIS wave has finished
    // its conv, before the barrier: a wave that would idle at the barrier converts a third of its tile instead); 2 = the outer tiles
    auto write_p = [&](const f32x16 (&v)[NTW], float mul, int part = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            if (part != 0 && (nt == 1) != (part == 1)) continue;
            const float m2 = (SCH::XS != 1.f) ? mul * p.slope : p.slope;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned qq[2][2][NPC];  // [octet g][pair e][piece]
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 vv = {v[nt][8 * q + 4 * g + 2 * e], v[nt][8 * q + 4 * g + 2 * e + 1]};
                        const f32x2 a = (SCH::XS != 1.f) ? vv * mul : vv, bq = vv * m2;  // lrelu(c v) = max(c v, c slope v): packed multiplies
                        SCH::split(max_nc(a[0], bq[0]), max_nc(a[1], bq[1]), qq[g][e]);
                    }
                // A lane holds 4 of the 8 channels of both octets of this column (the other 4 sit in lane ^ 32).  Swap the
                // halves across the two half-waves (v_permlane32_swap): lane (col, half h) then owns octet h completely and
                // stores 16 contiguous bytes -- 8-byte stores at a 32-byte lane stride cost 4.7 -> 4.1 ms in bank conflicts.
                char* dst = smem_raw + (2 * wm + q) * CH_BYTES + half * OCT_BYTES + col[nt] * 16;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                    *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        }
        // Positions outside the true sequence are zero INPUTS of every conv (the reference's per-layer zero padding).  Only the
        // windows that touch a sequence end (`edge`, wave-uniform) pay for it: the lane overwrites the slots of its out-of-sequence
        // columns with zeros -- interior windows (the bulk) carry no per-element mask instructions (2 of 7 VALU per element before).
        if (edge) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                if (!tok[nt] && (part == 0 || (nt == 1) == (part == 1))) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc)
                            *reinterpret_cast<uint4*>(smem_raw + (2 * wm + q) * CH_BYTES + half * OCT_BYTES + col[nt] * 16 + pc * PIECE_BYTES) = uint4{0u, 0u, 0u, 0u};
                }
        }
    };

    // ---- weights: one buffer descriptor, per-lane constant offset, scalar step offset ----------------------------
    // stream order: [conv][row tile][chunk * K + tap]; this wave walks its own row tile's steps
    auto stream_rsrc = [&](const uint16_t* ws) __attribute__((always_inline)) {
        const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(ws));
        const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(ws) >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t wrsrc = stream_rsrc(MRF ? p.bstream[0] : p.wstream);
    const unsigned lane16 = lane * 16;
    s16x8 A[2][NPC], Bv[2][NTW][NPC];
    auto load_a_piece = [&](s16x8 (&a)[NPC], int pc, int gstep) __attribute__((always_inline)) {
        a[pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, gstep * STEP_BYTES, 0));
    };
    int conv_steps = WAVES_M * NCH * K;  // steps of one conv in the stream
    int gbase = wm * NCH * K;            // this wave's first step of the current conv
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
        load_a_piece(A[0], pc, gbase);      // (tap 0, chunk 0)
        load_a_piece(A[1], pc, gbase + K);  // (tap 0, chunk 1)
    }

    RBS_STAMP();  // 1: window loaded (issued), weights prefetched
    write_p(R, SCH::XS);  // (out-of-sequence columns of R are 0 from the range-checked loads)
    RBS_STAMP();  // 2: first conversion done
    __syncthreads();
    RBS_STAMP();  // 3: barrier

    int center = (K - 1) / 2;
    auto load_b = [&](s16x8 (&bb)[NTW][NPC], int ch, int shift) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            // (unclamped: columns outside [0, W) are read only for output columns that are invalid anyway; LDS never faults)
            const char* src = smem_raw + ch * CH_BYTES + half * OCT_BYTES + (col[nt] + shift) * 16;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[nt][pc] = *reinterpret_cast<const s16x8*>(src + pc * PIECE_BYTES);
        }
    };
    // one conv out of P into acc (initialised by the caller).  Tap loop; per tap NCH steps (input-channel chunk ch with
    // register sets ch & 1).  On entry the weight sets hold this conv's (tap 0, chunk 0 / 1); each piece is re-fetched
    // in place for the step two ahead (chunk + 2 of this tap, else chunk + 2 - NCH of the next tap, else of the next
    // conv's tap 0) right after its last use, and the operand fragments of the next step are read during the current one.
    // (A last, discarded fragment read follows the final step; the stream is padded by one conv for the prefetch.)
    auto conv = [&](f32x16 (&acc)[NTW], int dil) __attribute__((always_inline)) {  // (accumulates into the given register set)
        load_b(Bv[0], 0, -center * dil);
        for (int j = 0; j < K; ++j) {
            const int shift = (j - center) * dil;
            const int n0 = (j + 1 < K) ? gbase + j + 1 : gbase + conv_steps;  // (next tap, chunk 0); chunk c is c * K steps further
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int set = ch & 1;
                if (ch + 1 < NCH) load_b(Bv[set ^ 1], ch + 1, shift);
                else load_b(Bv[set ^ 1], 0, shift + dil);
                const int nx = (ch + 2 < NCH) ? gbase + (ch + 2) * K + j : n0 + (ch + 2 - NCH) * K;  // two steps ahead
#pragma unroll
                for (int t = 0; t < NTERM; ++t) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[nt] = mfma32<SCH>(A[set][SCH::pa(t)], Bv[set][nt][SCH::pb(t)], acc[nt]);
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        if (t == SCH::last_a(pc)) load_a_piece(A[set], pc, nx);
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

    f32x16 XSr[MRF ? NTW : 1];  // whole-MRF launches: the branch sum (models.py:100-106), in branch order
    const int n_branch = MRF ? p.n_branch : 1;
    for (int br = 0; br < n_branch; ++br) {
    if (MRF && br > 0) {  // next branch: its own tap count and weight stream, R <- the stage input again (an L2 hit by now)
        K = p.bk[br];
        center = (K - 1) / 2;
        conv_steps = WAVES_M * NCH * K;
        gbase = wm * NCH * K;
        wrsrc = stream_rsrc(p.bstream[br]);
        load_r();
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
            load_a_piece(A[0], pc, gbase);
            load_a_piece(A[1], pc, gbase + K);
        }
        RBS_STAMP();  // branch switch: loads issued
        __syncthreads();  // every wave is done with the previous branch's last conv (reads of P)
        RBS_STAMP();
        write_p(R, SCH::XS);
        RBS_STAMP();
        __syncthreads();
        RBS_STAMP();
    }
    // R is carried in SCALED form between pairs: after a pair's second conv the accumulator IS the new residual times rs = XS * wsc
    // of that conv, and it stays in its registers -- the next second conv starts from fma(R, s2 / rs, b2 * s2) in place, the operand
    // conversion takes XS / rs as its factor.  Every factor is a power of two, so each value equals the unscaled formulation's
    // ((b2 + R) * s2, R = acc / s2) bit for bit; per pair it saves 24 packed multiplies and 24 packed adds of ~560 VALU instructions.
    float rs = 1.f;  // scale of R (1 after the load)
    for (int mm = 0; mm < p.n_conv; mm += 2) {
        const int m = br * p.n_conv + mm;  // (pair kernels: one branch, m = mm)
        float bv[16];
        // h = conv_d(P) + b1   (accumulator scale s1 = XS * wsc[m])
        const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
        bias_rows(p.bias[m], bv);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = (SCH::XS != 1.f) ? bv[r] * s1 : bv[r];
        RBS_STAMP();  // A: accumulators initialised
        conv(acc, p.dil[m]);
        RBS_STAMP();  // B: conv done
        if (early) write_p(acc, 1.f / p.wsc[m], 1);
        RBS_STAMP();  // C: inner tile converted
        __syncthreads();  // every wave is done reading P
        RBS_STAMP();  // D: barrier
        write_p(acc, 1.f / p.wsc[m], early ? 2 : 0);  // P = split(lrelu(h)), masked: acc / s1 * XS
        RBS_STAMP();  // E: outer tiles converted
        __syncthreads();
        RBS_STAMP();  // F: barrier
        // R = conv_1(P) + b2 + R, accumulated IN R's registers
        bias_rows(p.bias[m + 1], bv);
        if constexpr (SCH::XS != 1.f) {
            const float ratio = s2 / rs;
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bv[r] * s2;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[nt][r] = __builtin_fmaf(R[nt][r], ratio, bv[r]);
        } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[nt][r] = bv[r] + R[nt][r];
        }
        // (the residual rides in the accumulator: keeping R live across this conv costs 48 VGPRs, i.e. the third workgroup per
        //  CU or spills -- measured 4.8 -> 8.1 ms per step; in the vocoder |x| ~ |conv sum|, so folding costs no accuracy)
        RBS_STAMP();  // A
        conv(R, p.dil[m + 1]);
        RBS_STAMP();  // B
        rs = (SCH::XS != 1.f) ? s2 : 1.f;
        // (columns outside the sequence now hold garbage in R: they never feed a valid column -- write_p zeroes their operand slots --
        //  and the final store drops them or leaves them in the row's unspecified tail)
        if (mm + 2 < p.n_conv) {
            if (early) write_p(R, SCH::XS / rs, 1);
            RBS_STAMP();  // C
            __syncthreads();
            RBS_STAMP();  // D
            write_p(R, SCH::XS / rs, early ? 2 : 0);
            RBS_STAMP();  // E
            __syncthreads();
            RBS_STAMP();  // F
        }
    }
    if constexpr (SCH::XS != 1.f) {  // back to the plain residual for the branch sum / the store
        const float inv = 1.f / rs;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) R[nt][r] = R[nt][r] * inv;
    }
    if constexpr (MRF) {  // xs = rb_0(x), then xs += rb_j(x): the same adds, in the same order, as the per-branch launches' epilogues
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) XSr[nt][r] = (br == 0) ? 0.f + R[nt][r] : XSr[nt][r] + R[nt][r];
    }
    }  // branches

    RBS_STAMP();  // compute done
    // ---- write the central TT columns (registers -> global, 128-byte runs per row) ---------------------------------
    // buffer-addressed like the layer kernels' RowTile (conv_mfma.h): ONE lane-dependent offset per column tile (0x80000000 for a
    // column outside the window's own output range: loads return 0, stores are dropped), scalar row offsets -- the 64-bit pointer
    // arithmetic and per-element selects of the first version were ~600 VALU instructions per window, two conversion passes' worth
    const RowTile yt = row_tile(p.y + (size_t)b * C * p.T, C, p.T);
    const int row_bytes = p.T * 4;
    const bool has_acc = !MRF && p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = col[nt] - p.H;
        const int t = tile * p.TT + c;
        const bool ok = c >= 0 && c < p.TT && t < p.T;
        const int voff = ok ? (4 * half * p.T + t) * 4 : (int)0x80000000;
        float v[16];
        if constexpr (MRF) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (n_branch > 1) ? XSr[nt][r] / p.div : XSr[nt][r];
        } else if (has_acc) {
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = row_tile_load(yt, voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = yv[r] + R[nt][r];
            if (do_div) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] / p.div;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f + R[nt][r];  // (as the accumulating form with y = 0: -0 comes out as +0 in both)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) row_tile_store(yt, v[r], voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes);
    }
}

// ---- 16-channel variant on v_mfma_f32_16x16x32_bf16 -----------------------------------------------------------------
// One MFMA k-step = 32 = 16 channels x TWO taps (an odd tap count is padded with one zero-weight tap), so a 16-channel
// layer needs no zero-padded MFMA rows or k-slots beyond that.  Window 768 columns (72 KiB of LDS, two workgroups
// per CU); wave w holds columns [192w, 192w+192) as 12 tiles of 16x16.
//   A (weights)  : lane l = row l&15, k-group g = l>>4: channels 8(g&1)..+7 of tap 2*step + (g>>1); the host packs
//                  [conv][step][piece][lane][8 bf16].
//   B (operand)  : lane l = column l&15 (+ tap shift (g>>1)*dil), channels 8(g&1)..+7: one ds_read_b128 per piece.
//                  The two channel octets of a column swap places on odd 4-column groups: a ds_read_b128 lane group
//                  here is {8 columns of octet 0, the 8 columns 8 further on of octet 1}, which land on 16 distinct
//                  16-byte bank slots only if the swap does NOT depend on bit 3 of the column (the (col >> 3) swap of
//                  the 32-column layout cost 0.34 conflict cycles per busy cycle here).
//   C/D          : column l&15, rows 4(l>>4) + r.
// Per step the 12 tiles are walked in pairs (terms outer, the two tiles inner, so dependent MFMAs are one apart);
// the next pair's fragments are fetched during the current pair's MFMAs.
constexpr int RBS16_W = 768;

template <class SCH>
__global__ __launch_bounds__(256, SCH::NP == 3 ? 2 : 3) void resblock16_split_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT, STEP_BYTES = SCH::NP * 1024;
    constexpr int C = 16, NT = 12, W = RBS16_W, NP = NT / 2;
    const int K = p.k, S = (K + 1) / 2;  // tap pairs per conv (even: the host only takes k = 3, 7, 11)
    constexpr int OCT_BYTES = W * 16, PIECE_BYTES = 2 * OCT_BYTES;  // [piece][octet][col][8 channels] (see resblock_split_kernel)
    static_assert(4 * NT * 16 == W, "4 waves x NT tiles cover the window");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // NPC * PIECE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4;  // C/D row group; as an operand lane: channel octet g4 & 1, tap parity g4 >> 1
    const int l15 = lane & 15;
    const int b = blockIdx.x / p.tiles;
    const int tile = blockIdx.x - b * p.tiles;
    const int t_base = tile * p.TT - p.H;
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    const bool edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);
    const int col0 = wave * (NT * 16) + l15;  // this lane's window column in tile 0 (tile nt adds 16 nt)
    const bool early = RBS_EARLY_INNER && __builtin_amdgcn_readfirstlane(p.early) != 0;

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
    // `part`: 0 = every tile pair; 1 = the INNER pairs only (tiles 2..9 = columns 32..159 of the wave's 192: no other wave reads
    // them, every conv's reach being <= 25 columns -- written before the barrier that ends a conv, see resblock_split_kernel);
    // 2 = the outer pairs
    auto write_p = [&](const f32x4 (&v)[NT], float mul, int part = 0) __attribute__((always_inline)) {
        const float m2 = (SCH::XS != 1.f) ? mul * p.slope : p.slope;
#pragma unroll
        for (int np = 0; np < NT; np += 2) {  // tile pairs: lane groups with even g4 end up owning tile np's octet, odd ones tile np+1's
            if (part != 0 && (np >= 2 && np < NT - 2) != (part == 1)) continue;
            unsigned qq[2][2][NPC];           // [tile of the pair][pair e][piece]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x2 vv = {v[np + u][2 * e], v[np + u][2 * e + 1]};
                    const f32x2 a = (SCH::XS != 1.f) ? vv * mul : vv, bq = vv * m2;
                    SCH::split(max_nc(a[0], bq[0]), max_nc(a[1], bq[1]), qq[u][e]);
                }
            }
            // a lane holds 4 of the 8 channels of an octet (the other 4 sit in lane ^ 16): v_permlane16_swap gives the even
            // 16-lane rows the whole octet of tile np and the odd rows that of tile np + 1 -> one 16-byte store per lane
            const int c = col0 + (np + (g4 & 1)) * 16;
            char* dst = smem_raw + (g4 >> 1) * OCT_BYTES + c * 16;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
            }
        }
        if (edge) {  // windows at a sequence end only: zero the operand slots of out-of-sequence columns (see resblock_split_kernel)
#pragma unroll
            for (int np = 0; np < NT; np += 2) {
                if (part != 0 && (np >= 2 && np < NT - 2) != (part == 1)) continue;
                const int c = col0 + (np + (g4 & 1)) * 16;  // the column whose octet this lane stored above
                const int t = t_base + c;
                if (!(t >= 0 && t < Tlim)) {
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        *reinterpret_cast<uint4*>(smem_raw + (g4 >> 1) * OCT_BYTES + c * 16 + pc * PIECE_BYTES) = uint4{0u, 0u, 0u, 0u};
                }
            }
        }
    };

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
    write_p(R, SCH::XS);
    __syncthreads();

    const int center = (K - 1) / 2;
    int gstep = 0;
    // fragments of tile pair pr at step s: column + (2s + tap parity - center) * dil, clamped to the window
    auto load_b = [&](s16x8 (&bb)[2][NPC], int pr, int shift) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // (no clamp: a column outside [0, W) is only ever read for output columns within the conv's reach of the window edge,
            //  which are invalid by construction; the address stays inside this piece's neighbours or out of the allocation, where
            //  LDS reads return 0 -- garbage either way, never a fault.  The clamp was 2 VALU per fragment read: 85 of 183 per step pair)
            const char* src = smem_raw + (g4 & 1) * OCT_BYTES + (col0 + (2 * pr + q) * 16 + shift) * 16;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[q][pc] = *reinterpret_cast<const s16x8*>(src + pc * PIECE_BYTES);
        }
    };
    // one conv: loop over step pairs (weight set 0 / 1), six tile pairs per step alternating the operand sets
    auto conv = [&](int dil) __attribute__((always_inline)) {
        const int tpd = (g4 >> 1) * dil - center * dil;  // this lane's tap-parity shift, centred
        // An odd tap count is padded with ONE zero-weight tap (index K, the odd-parity lanes of the last step).  Its operand must be
        // finite -- 0 x inf = NaN -- and, with unclamped reads, columns beyond the conv's true reach may hold anything: those lanes
        // re-read tap K - 1 (inside the reach, finite wherever the output column is valid) instead of stepping one tap further out.
        const int tpd_last = tpd - (((K & 1) && (g4 >> 1)) ? dil : 0);
        load_b(Bv[0], 0, S == 1 ? tpd_last : tpd);
        for (int s2 = 0; s2 < S; s2 += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int s = s2 + sub;
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    const int cur = pr & 1;  // NP is even: every step starts on operand set 0
                    if (pr + 1 < NP) load_b(Bv[cur ^ 1], pr + 1, (s == S - 1 ? tpd_last : tpd) + 2 * s * dil);
                    else load_b(Bv[cur ^ 1], 0, (s + 1 == S - 1 ? tpd_last : tpd) + 2 * (s + 1) * dil);  // (after the last step: discarded)
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
            // issue order: the next pair's fragment reads in the shadow of the first MFMAs of each pair
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

    for (int m = 0; m < p.n_conv; m += 2) {
        float bv[4];
        const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (SCH::XS != 1.f) ? p.bias[m][4 * g4 + r] * s1 : p.bias[m][4 * g4 + r];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] = bv[r];
        conv(p.dil[m]);
        if (early) write_p(acc, 1.f / p.wsc[m], 1);
        __syncthreads();
        write_p(acc, 1.f / p.wsc[m], early ? 2 : 0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias[m + 1][4 * g4 + r];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] = (SCH::XS != 1.f) ? bv[r] * s2 : bv[r];
        conv(p.dil[m + 1]);
        const float i2 = 1.f / s2;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) R[nt][r] = (SCH::XS != 1.f) ? fmaf(acc[nt][r], i2, R[nt][r]) : acc[nt][r] + R[nt][r];
        if (m + 2 < p.n_conv) {
            if (early) write_p(R, SCH::XS, 1);
            __syncthreads();
            write_p(R, SCH::XS, early ? 2 : 0);
            __syncthreads();
        }
    }

    const RowTile yt = row_tile(p.y + (size_t)b * C * p.T, C, p.T);  // buffer-addressed (see resblock_split_kernel)
    const int row_bytes = p.T * 4;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    if (has_acc) {
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += 4) {  // four column tiles per batch: 16 loads in flight
            float yv[4][4];
            int voff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = col0 + (n0 + i) * 16 - p.H;
                const int t = tile * p.TT + c;
                voff[i] = (c >= 0 && c < p.TT && t < p.T) ? (4 * g4 * p.T + t) * 4 : (int)0x80000000;
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[i][r] = row_tile_load(yt, voff[i], r * row_bytes);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = yv[i][r] + R[n0 + i][r];
                    if (do_div) v = v / p.div;
                    row_tile_store(yt, v, voff[i], r * row_bytes);
                }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = col0 + nt * 16 - p.H;
            const int t = tile * p.TT + c;
            const int voff = (c >= 0 && c < p.TT && t < p.T) ? (4 * g4 * p.T + t) * 4 : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 4; ++r) row_tile_store(yt, 0.f + R[nt][r], voff, r * row_bytes);
        }
    }
}

template <class SCH>
inline hipError_t launch_resblock_split_s(int C, const ResblockSplitParams& p, hipStream_t s) {
    const size_t lds = (size_t)SCH::NP * 2 * RBS_W * 32 * (C == 256 ? 2 : 1);  // the same up to 128 channels (72 KiB with three pieces)
    auto kern = (C == 16) ? resblock16_split_kernel<SCH> : (C == 256) ? resblock_split_kernel<SCH, 16> : (C == 128) ? resblock_split_kernel<SCH, 8>
              : (C == 64) ? resblock_split_kernel<SCH, 4> : resblock_split_kernel<SCH, 2>;
    static DynLdsOnce lds_once[5];
    const int slot = (C == 16) ? 0 : (C == 256) ? 4 : (C == 128) ? 3 : (C == 64) ? 2 : 1;
    {
        hipError_t e = ensure_dyn_lds(lds_once[slot], reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles * p.B), dim3(C == 256 ? 512 : 256), lds, s, p);
    return hipGetLastError();
}
// whole-MRF launches (32 channels): 8 waves, 768-column window, NP x 48 KiB of LDS
constexpr int rbs_mrf_window(int C) { return C == 32 ? 768 : 0; }
inline bool resblock_mrf_has(int C) { return C == 32; }
template <class SCH>
inline hipError_t launch_mrf_split_s(int C, const ResblockSplitParams& p, hipStream_t s) {
    if (!resblock_mrf_has(C)) return hipErrorInvalidValue;
    const size_t lds = (size_t)SCH::NP * 49152;
    auto kern = resblock_split_kernel<SCH, 2, 8, true>;
    static DynLdsOnce lds_once;
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles * p.B), dim3(512), lds, s, p);
    return hipGetLastError();
}
hipError_t launch_mrf_split_f16x3(int C, const ResblockSplitParams& p, hipStream_t s);   // csrc/tu_mrf.hip
hipError_t launch_mrf_split_bf16(int C, const ResblockSplitParams& p, hipStream_t s);    // csrc/tu_mrf_single.hip
hipError_t launch_mrf_split_f16(int C, const ResblockSplitParams& p, hipStream_t s);     // csrc/tu_mrf_single.hip
inline hipError_t launch_mrf_split(int scheme, int C, const ResblockSplitParams& p, hipStream_t s) {
    switch (scheme) {
        case SchF16x3::ID: return launch_mrf_split_f16x3(C, p, s);
        case SchBf16::ID: return launch_mrf_split_bf16(C, p, s);
        case SchF16::ID: return launch_mrf_split_f16(C, p, s);
        default: return hipErrorInvalidValue;  // (three bf16 pieces: R + accumulators + branch sum + fragments do not fit 256 VGPRs)
    }
}
inline bool resblock_mrf_scheme(int scheme) { return scheme == SchF16x3::ID || scheme == SchBf16::ID || scheme == SchF16::ID; }
hipError_t launch_resblock_split_f16x3(int C, const ResblockSplitParams& p, hipStream_t s);   // csrc/tu_resblock.hip
hipError_t launch_resblock_split_bf16x6(int C, const ResblockSplitParams& p, hipStream_t s);  // csrc/tu_resblock.hip
hipError_t launch_resblock_split_bf16(int C, const ResblockSplitParams& p, hipStream_t s);    // csrc/tu_split_single.hip
hipError_t launch_resblock_split_f16(int C, const ResblockSplitParams& p, hipStream_t s);     // csrc/tu_split_single.hip
inline hipError_t launch_resblock_split(int scheme, int C, const ResblockSplitParams& p, hipStream_t s) {
    switch (scheme) {
        case SchBf16x6::ID: return launch_resblock_split_bf16x6(C, p, s);
        case SchF16x3::ID: return launch_resblock_split_f16x3(C, p, s);
        case SchBf16::ID: return launch_resblock_split_bf16(C, p, s);
        case SchF16::ID: return launch_resblock_split_f16(C, p, s);
        default: return hipErrorInvalidValue;
    }
}
// odd tap counts with (k + 1) / 2 even; at 64 channels the 192-column window loses too much to the halo of k = 11
// at 128 channels (window 96) only the k = 3 pairs: the layer kernel is latency-bound there (three MFMA steps per chunk), the
// fused pair reads x once and writes y once instead of five passes, and a pair's reach (<= 6) leaves >= 87 % of the window
inline bool resblock_split_has(int C, int k) {
    if (C == 256 || C == 128) return k == 3;
    if (C == 64) return k == 3 || k == 7;
    return (C == 32 || C == 16) && (k == 3 || k == 7 || k == 11);
}
inline int resblock_split_window(int C) { return C == 16 ? RBS16_W : C >= 128 ? RBS_W / 4 : C == 64 ? RBS_W / 2 : RBS_W; }
inline int resblock_split_steps(int C, int k) { return C == 16 ? (k + 1) / 2 : (C / 32) * (C / 16) * k; }  // weight steps per conv

}  // namespace parrot
