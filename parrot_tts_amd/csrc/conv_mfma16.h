// conv_mfma16.h -- exact-fp32 implicit-GEMM Conv1d for layers with <= 16 output channels (HiFi-GAN stage 4,
// conv_post, the duration projection): same slab / fragment-stream structure as conv_mfma.h, but on
// v_mfma_f32_16x16x4_f32 (16 rows x 16 columns x 4 k per instruction, 32-cycle issue, same 64 FLOP/clk/SIMD rate),
// so a 16-channel layer no longer pays for 32 MFMA rows of which half multiply zero weights.
//
//   A (weights)  : lane l holds row l&15, k = l>>4 of each 16x4 step; the host packs four consecutive steps
//                  (= all 16 channels of a chunk at one tap) per lane: [chunk][tap][lane][4] floats, one
//                  global_load_dwordx4 per tap.
//   B (slab)     : lane l reads channel (4*step + (l>>4)), column (l&15) of its 16-column tile; the LDS row stride
//                  is 16 mod 32 dwords, so the two channel rows inside each 32-lane half land on disjoint banks.
//   C/D          : col = lane&15, row = 4*(lane>>4) + r  (r = 0..3).
// Block = 4 waves side by side, each 16 x (NT16*16) columns.
#pragma once
#include "conv_mfma.h"

namespace parrot {

template <int NT16, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_mfma16_kernel(const ConvParams p) {
    constexpr int WCOLS = NT16 * 16;        // columns per wave
    constexpr int BN = 4 * WCOLS;
    constexpr int RS = BN + CONV_HALO + 16;  // == 16 (mod 32): rows ch and ch+1 hit different bank halves
    constexpr int CI = 16;
    constexpr int ROWS_PW = CI / 4;
    constexpr int COLS_IT = (RS + 63) / 64;
    static_assert(RS % 32 == 16, "row stride must be 16 mod 32 dwords");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][CI][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4;   // k group 0..3
    const int l15 = lane & 15;

    const int b = blockIdx.x / p.tiles_n;
    const int tn = blockIdx.x - b * p.tiles_n;
    const int t0 = tn * BN;
    const int W = BN + (p.k - 1) * p.dil;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const float* __restrict__ xb = p.x + (size_t)b * p.x_bstride;

    float stage[ROWS_PW][COLS_IT];
    auto load_slab = [&](int c) {
#pragma unroll
        for (int r = 0; r < ROWS_PW; ++r) {
            const int ch = c * CI + wave * ROWS_PW + r;
            const bool chok = ch < p.Cin;
            const float* __restrict__ row = xb + (size_t)(chok ? ch : 0) * p.Tin;
#pragma unroll
            for (int i = 0; i < COLS_IT; ++i) {
                const int col = lane + 64 * i;
                const int tin = t0 - p.pad_left + col;
                const bool ok = chok && col < W && tin >= 0 && tin < Tlim;
                float v = row[ok ? tin : 0];
                v = ok ? v : 0.f;
                const float vs = v * p.pre_slope;
                stage[r][i] = (p.pre == PRE_LRELU && v < 0.f) ? vs : v;
            }
        }
    };
    auto store_slab = [&](int buf) {
        float* dst = smem + buf * (CI * RS);
#pragma unroll
        for (int r = 0; r < ROWS_PW; ++r)
#pragma unroll
            for (int i = 0; i < COLS_IT; ++i) {
                const int col = lane + 64 * i;
                if (col < RS) dst[(wave * ROWS_PW + r) * RS + col] = stage[r][i];
            }
    };

    // accumulators = bias (+ residual), fetched in the prologue (see conv_mfma.h)
    float* __restrict__ yb = p.y + (size_t)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const bool fold_res = rb != nullptr && p.act == ACT_NONE;
    f32x4 acc[NT16];
    const int n_wave = t0 + wave * WCOLS;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
        const int n = n_wave + nt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kg + r;
            acc[nt][r] = fold_res ? rb[(n < p.Ncols && m < p.M) ? m * p.Tout + n : 0] : 0.f;
        }
    }
    if (p.bias) {
        float bsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bsv[r] = p.bias[min(4 * kg + r, p.M - 1)];
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] += bsv[r];
    }

    const f32x4* __restrict__ ap = reinterpret_cast<const f32x4*>(p.wfrag) + lane;
    f32x4 a_nxt = ap[0], a_cur;

    load_slab(0);
    store_slab(0);
    __syncthreads();

    const int bbase = kg * RS + wave * WCOLS + l15;
    int it = 0;
    float bv_cur[4][NT16], bv_nxt[4][NT16];
    for (int c = 0; c < p.nchunks; ++c) {
        const float* __restrict__ xs = smem + (c & 1) * (CI * RS) + bbase;
        const bool more = (c + 1 < p.nchunks);
        if (more) load_slab(c + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt) bv_cur[e][nt] = xs[(4 * e) * RS + nt * 16];
        for (int j = 0; j < p.k; ++j) {
            ++it;
            const float* __restrict__ xn = xs + (j + 1) * p.dil;  // next tap (after the last: in-row, discarded)
            a_cur = a_nxt;
            a_nxt = ap[(size_t)it * 64];  // one padding group at the end of the stream
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) bv_nxt[e][nt] = xn[(4 * e) * RS + nt * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[e], bv_cur[e][nt], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) bv_cur[e][nt] = bv_nxt[e][nt];
        }
        if (more) store_slab((c + 1) & 1);
        __syncthreads();
    }

    // epilogue (plain conv only): remaining loads batched per tile, branch-free finish, masked stores
    const bool late_res = rb != nullptr && !fold_res;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    const float act_lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
        const int n = n_wave + nt * 16 + l15;
        int off[4];
        float rv[4], yv[4], v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kg + r;
            off[r] = (n < p.Ncols && m < p.M) ? m * p.Tout + n : -1;
            rv[r] = 0.f;
            yv[r] = 0.f;
        }
        if (late_res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = rb[off[r] < 0 ? 0 : off[r]];
        }
        if (has_acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = yb[off[r] < 0 ? 0 : off[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = yv[r] + (fmaxf(acc[nt][r], act_lo) + rv[r]);
        if (do_div) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / p.div;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (off[r] >= 0) yb[off[r]] = v[r];
    }
}

constexpr int MFMA16_NT16 = 8;  // 16 x 128 per wave, 16 x 512 per workgroup
inline hipError_t launch_conv_mfma16(const ConvParams& p, hipStream_t s) {
    constexpr int BN = 4 * MFMA16_NT16 * 16;
    const size_t lds = (size_t)2 * 16 * (BN + CONV_HALO + 16) * sizeof(float);
    auto kern = conv_mfma16_kernel<MFMA16_NT16, 2>;
    static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_n * p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}

}  // namespace parrot
