// resblock_split16.h -- the fused ResBlock1 pair kernel of resblock_split.h for 32 and 64 channels on the 16x16x32 MFMA
// shape (as conv_split16.h does for the layer kernel): a k-step holds 32 channels of ONE tap, a wave owns ALL output
// channels of its columns (CM = C / 16 row tiles x NT column tiles of 16), so one column fragment read from LDS feeds CM
// MFMAs per term instead of one -- half (C = 32) / a quarter (C = 64) of the LDS fragment bytes per MAC of the 32x32x16
// version, whose waves own one 32-row tile each -- and the instruction itself moves a quarter of the accumulator registers.
//
//   R (running residual, fp32) : registers, C/D layout of 16x16 tiles: column l & 15, rows 16 tm + 4 (l >> 4) + r
//   P (conv operand, NP pieces): LDS [piece][32-channel chunk][octet 0..3][column][8 channels] (16 B per (column, octet))
//   per pair:  P = split(lrelu(R));  h = conv_d(P) + b1;  P = split(lrelu(h));  R = conv_1(P) + b2 + R
//
// C = 32: window 384 columns (4 waves x 6 tiles), one k-step per tap; C = 64: window 192 (4 x 3), two k-steps per tap.
// Weights: [conv][tap * (C/32) + chunk][row tile][piece][lane][8 halves] (lane = row l & 15, channels 8 (l >> 4)..), fetched
// one step ahead into the other of two register sets; column fragments in ONE set, refilled in place during the last row
// tile's MFMAs of a step.  Everything else (halo / reach bookkeeping, zero padding by masking, ragged rows, epilogue modes,
// accumulator scaling of the fp16 scheme) is resblock_split.h's.
#pragma once
#include "resblock_split.h"

namespace parrot {

template <class SCH, int CM>
__global__ __launch_bounds__(256, 2) void resblock_split16_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT;
    constexpr int C = 16 * CM, NCH = C / 32;       // 32-channel chunks = k-steps per tap
    constexpr int NT = (CM == 2) ? 6 : 3;          // column tiles per wave
    constexpr int W = 4 * NT * 16;                 // window: 384 / 192 columns
    constexpr int OCT_BYTES = W * 16, CH_BYTES = 4 * OCT_BYTES, PIECE_BYTES = NCH * CH_BYTES;
    constexpr int STEP_BYTES = CM * NPC * 1024;    // weight stream per k-step: [row tile][piece][lane][8]
    static_assert(CM == 2 || CM == 4, "32 or 64 channels");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // NPC * PIECE_BYTES
    const int K = p.k;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / p.tiles;
    const int tile = blockIdx.x - b * p.tiles;
    const int t_base = tile * p.TT - p.H;
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    const bool edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);
    const int col0 = wave * (NT * 16) + l15;  // this lane's window column in tile 0 (tile tn adds 16 tn)

    f32x4 R[CM][NT], acc[CM][NT];
    {
        const size_t xaddr = reinterpret_cast<size_t>(p.x + (size_t)b * C * p.T);
        const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)x_hi << 32) | x_lo), 0, 0x7fffffff, 0x00020000);
        const int row_bytes = p.T * 4;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
            const int t = t_base + col0 + tn * 16;
            const int voff = (t >= 0 && t < Tlim) ? t * 4 + 4 * g4 * row_bytes : (int)0x80000000;  // out of range -> 0
#pragma unroll
            for (int tm = 0; tm < CM; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    R[tm][tn][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, (16 * tm + r) * row_bytes, 0));
        }
    }
    // accumulator tiles -> operand buffer.  A lane holds rows 16 tm + 4 g4 + r: 4 of the 8 channels of octet 2 tm + (g4 >> 1)
    // (the other 4 sit in lane ^ 16).  v_permlane16_swap over a PAIR of row tiles (tm, tm + 1) gives the even 16-lane rows
    // the whole octet of tile tm and the odd rows that of tile tm + 1: one 16-byte store per lane, consecutive columns.
    auto write_p = [&](const f32x4 (&v)[CM][NT], float mul) __attribute__((always_inline)) {
        const float m2 = (SCH::XS != 1.f) ? mul * p.slope : p.slope;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
            const int c = col0 + tn * 16;
            const int t = t_base + c;
            const bool tk = t >= 0 && t < Tlim;
#pragma unroll
            for (int tp = 0; tp < CM; tp += 2) {
                unsigned qq[2][2][NPC];  // [row tile of the pair][pair e][piece]
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 vv = {v[tp + u][tn][2 * e], v[tp + u][tn][2 * e + 1]};
                        const f32x2 a = (SCH::XS != 1.f) ? vv * mul : vv, bq = vv * m2;
                        float v0 = max_nc(a[0], bq[0]), v1 = max_nc(a[1], bq[1]);
                        if (edge) {
                            v0 = tk ? v0 : 0.f;
                            v1 = tk ? v1 : 0.f;
                        }
                        SCH::split(v0, v1, qq[u][e]);
                    }
                const int o = 2 * (tp + (g4 & 1)) + (g4 >> 1);  // octet of the whole channel range this lane stores
                char* dst = smem_raw + (o >> 2) * CH_BYTES + (o & 3) * OCT_BYTES + c * 16;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                    *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        }
    };

    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wstream));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wstream) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    s16x8 A[2][CM][NPC], Bv[NT][NPC];
    auto load_a = [&](s16x8 (&a)[CM][NPC], int gstep) __attribute__((always_inline)) {
#pragma unroll
        for (int tm = 0; tm < CM; ++tm)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc)
                a[tm][pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + (tm * NPC + pc) * 1024, gstep * STEP_BYTES, 0));
    };
    auto load_b_tile = [&](s16x8 (&bb)[NPC], int tn, int ch, int shift) __attribute__((always_inline)) {
        const int cc = min(max(col0 + tn * 16 + shift, 0), W - 1);  // clamped: only garbage columns ever read a clamped one
        const char* src = smem_raw + ch * CH_BYTES + g4 * OCT_BYTES + cc * 16;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) bb[pc] = *reinterpret_cast<const s16x8*>(src + pc * PIECE_BYTES);
    };
    load_a(A[0], 0);
    write_p(R, SCH::XS);
    __syncthreads();

    int gstep = 0;  // k-step in the launch's weight stream
    // one k-step: weights of the NEXT step into the other register set, CM x NTERM x NT MFMAs, column fragments of the next
    // step (chunk chn, tap shift shn) re-read in place during the last row tile's last term
    auto step = [&](auto par, int chn, int shn) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        load_a(A[PAR ^ 1], gstep + 1);
#pragma unroll
        for (int tm = 0; tm < CM; ++tm)
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    acc[tm][tn] = mfma16<SCH>(A[PAR][tm][SCH::pa(t)], Bv[tn][SCH::pb(t)], acc[tm][tn]);
                    if (tm == CM - 1 && t == NTERM - 1) load_b_tile(Bv[tn], tn, chn, shn);
                }
        // issue order: the weight fetches in the shadow of the first MFMAs, a fragment refill after each of the last NT
        constexpr int NMF = CM * NTERM * NT;
        int vm_left = CM * NPC;
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (m >= NMF - NT) {
#pragma unroll
                for (int i = 0; i < NPC; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else if (vm_left > 0) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                --vm_left;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ++gstep;
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // one conv out of P into acc (initialised by the caller).  PAR = weight register set of its first step.
    auto conv = [&](auto par, int dil) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        const int center = (K - 1) / 2;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) load_b_tile(Bv[tn], tn, 0, -center * dil);
        if constexpr (NCH == 2) {  // two k-steps per tap: every tap starts on the same register set
            for (int j = 0; j < K; ++j) {
                const int sh = (j - center) * dil;
                step(std::integral_constant<int, PAR>{}, 1, sh);
                step(std::integral_constant<int, PAR ^ 1>{}, 0, sh + dil);  // (after the last tap: a discarded read)
            }
        } else {                   // one k-step per tap, K odd: taps in pairs + a last one; the next conv starts on the other set
            int j = 0;
            for (; j + 1 < K; j += 2) {
                step(std::integral_constant<int, PAR>{}, 0, (j + 1 - center) * dil);
                step(std::integral_constant<int, PAR ^ 1>{}, 0, (j + 2 - center) * dil);
            }
            step(std::integral_constant<int, PAR>{}, 0, (K - center) * dil);  // (discarded read)
        }
    };
    auto bias_rows = [&](const float* __restrict__ bias, float scale, float (&bv)[CM][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int tm = 0; tm < CM; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[tm][r] = bias[16 * tm + 4 * g4 + r] * scale;
    };
    auto pair = [&](auto par, int m) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr int PAR2 = (NCH == 2) ? PAR : (PAR ^ 1);  // an odd number of steps per conv flips the set
        float bv[CM][4];
        const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
        bias_rows(p.bias[m], (SCH::XS != 1.f) ? s1 : 1.f, bv);
#pragma unroll
        for (int tm = 0; tm < CM; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tm][tn][r] = bv[tm][r];
        conv(std::integral_constant<int, PAR>{}, p.dil[m]);
        __syncthreads();  // every wave is done reading P
        write_p(acc, 1.f / p.wsc[m]);
        __syncthreads();
        bias_rows(p.bias[m + 1], 1.f, bv);
#pragma unroll
        for (int tm = 0; tm < CM; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tm][tn][r] = (SCH::XS != 1.f) ? (bv[tm][r] + R[tm][tn][r]) * s2 : bv[tm][r] + R[tm][tn][r];
        conv(std::integral_constant<int, PAR2>{}, p.dil[m + 1]);
        const float i2 = 1.f / s2;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
            const int t = t_base + col0 + tn * 16;
            const bool tk = !edge || (t >= 0 && t < Tlim);
#pragma unroll
            for (int tm = 0; tm < CM; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) R[tm][tn][r] = tk ? ((SCH::XS != 1.f) ? acc[tm][tn][r] * i2 : acc[tm][tn][r]) : 0.f;
        }
        if (m + 2 < p.n_conv) {
            __syncthreads();
            write_p(R, SCH::XS);
            __syncthreads();
        }
    };
    // (two convs per pair: with an odd step count per conv the pair ends on the set it started on, so every pair starts on set 0)
    for (int m = 0; m < p.n_conv; m += 2) pair(P0{}, m);
    (void)sizeof(P1);

    float* __restrict__ yb = p.y + (size_t)b * C * p.T;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
        const int c = col0 + tn * 16 - p.H;
        const int t = tile * p.TT + c;
        const bool ok = c >= 0 && c < p.TT && t < p.T;
#pragma unroll
        for (int tm = 0; tm < CM; ++tm) {
            float yv[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_acc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = yb[ok ? (size_t)(16 * tm + 4 * g4 + r) * p.T + t : 0];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = yv[r] + R[tm][tn][r];
                if (do_div) v = v / p.div;
                if (ok) yb[(size_t)(16 * tm + 4 * g4 + r) * p.T + t] = v;
            }
        }
    }
}

template <class SCH>
inline hipError_t launch_resblock_split16_s(int C, const ResblockSplitParams& p, hipStream_t s) {
    const size_t lds = (size_t)SCH::NP * (C / 32) * 4 * (C == 32 ? 384 : 192) * 16;
    auto kern = (C == 64) ? resblock_split16_kernel<SCH, 4> : resblock_split16_kernel<SCH, 2>;
    static DynLdsOnce lds_once[2];
    const int slot = (C == 64) ? 1 : 0;
    {
        hipError_t e = ensure_dyn_lds(lds_once[slot], reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles * p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}
inline bool resblock_split16_has(int scheme, int C) { return scheme == SchF16x3::ID && (C == 32 || C == 64); }
hipError_t launch_resblock_split16_f16x3(int C, const ResblockSplitParams& p, hipStream_t s);  // csrc/tu_split16.hip
inline hipError_t launch_resblock_split16(int scheme, int C, const ResblockSplitParams& p, hipStream_t s) {
    if (scheme == SchF16x3::ID) return launch_resblock_split16_f16x3(C, p, s);
    return hipErrorInvalidValue;
}
inline int resblock_split16_steps(int C, int k) { return k * (C / 32); }  // k-steps per conv; each CM * NP KiB

}  // namespace parrot
