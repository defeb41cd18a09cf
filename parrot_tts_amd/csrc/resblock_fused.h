// resblock_fused.h -- one whole HiFi-GAN ResBlock (reference utils/vocoder/models.py:31-38 / :58-62) per launch
// for the low-channel stages (C <= 32), exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Layer by layer these stages are HBM / latency bound: every conv streams (B, C, T) in and out (15 passes per
// ResBlock1) and each launch pays its own prologue and epilogue per 32x512 tile.  Here a workgroup loads one
// [C][W0] window (output tile + the block's total receptive-field halo H) into LDS once, runs all 2*n_dil convs
// out of LDS (ping-pong between a running-residual buffer R and a scratch buffer S, both fp32), and writes the
// central W0 - 2H columns once: 1 read + 1 (read-)modify-write per ResBlock instead of 15 passes.
//
//   ResBlock1, per dilation d:   S = lrelu(conv_d(lrelu(R)) + b1);   R = conv_1(S) + b2 + R
//   ResBlock2, per dilation d:   S = conv_d(lrelu(R)) + b + R;       swap(R, S)
//
// Each conv shrinks the valid region by its own halo, so after the last conv exactly the central columns are
// valid (columns outside are garbage that never feeds a valid one).  Positions outside the true sequence
// [0, T) are forced to zero after every conv: that IS the reference's per-layer zero padding.
// Weights are the conv plans' fragment-ordered fp32 packs (tile config 2: one 32-row m-tile, 16-channel chunks),
// streamed from L2 as in conv_mfma.h; B operands are ds_read_b32 of the LDS rows (conflict free for every
// dilation); 8 waves x NTW column tiles of 32 cover the W0 window.
#pragma once
#include "conv_mfma.h"

namespace parrot {

constexpr int RB_MAX_CONVS = 8;
constexpr int RB_PAD = 32;  // LDS columns either side of the window (reads reach at most (k-1)/2*dil <= 25 outside)

struct ResblockParams {
    const float* x;   // (B, C, T) stage input
    float* y;         // (B, C, T) MRF accumulator
    const float* wfrag[RB_MAX_CONVS];
    const float* bias[RB_MAX_CONVS];
    int dil[RB_MAX_CONVS];
    int n_conv;       // 2*n_dil (type 1) or n_dil (type 2)
    int type;         // 1 / 2
    int k, C, T, B;
    int H;            // total halo of the block
    int TT;           // output columns per workgroup = W0 - 2H
    int tiles;        // ceil(T / TT)
    int epi;          // EPI_STORE / EPI_ADD / EPI_ADD_DIV on y
    float div;
    float slope;      // leaky-relu slope (0.1)
    const int32_t* row_len;  // optional per-row true length in base units (x row_len_mul = samples at this stage)
    int row_len_mul;
    int row_len_add;         // true length of a row of n > 0 units at this layer = n * row_len_mul + row_len_add (odd k - u upsampling stages add samples)
};

template <int NTW>
__global__ __launch_bounds__(512, 1) void resblock_fused_kernel(const ResblockParams p) {
    constexpr int W0 = 8 * NTW * 32;       // window columns
    constexpr int RS = W0 + 2 * RB_PAD;    // LDS row stride (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // R[C][RS], S[C][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int b = blockIdx.x / p.tiles;
    const int tile = blockIdx.x - b * p.tiles;
    const int t_base = tile * p.TT - p.H;  // sequence position of window column 0
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;  // this row's true length
    const int C = p.C;
    const int nchunks = (C + 15) / 16;

    float* R = smem;
    float* S = smem + C * RS;
    const float* __restrict__ xb = p.x + (size_t)b * C * p.T;

    // ---- load the window (zero outside the sequence), clear the pads of both buffers ------------------------
    for (int idx = tid; idx < C * W0; idx += 512) {
        const int row = idx / W0, c = idx - row * W0;
        const int t = t_base + c;
        const bool ok = t >= 0 && t < Tlim;
        const float v = xb[(size_t)row * p.T + (ok ? t : 0)];
        R[row * RS + RB_PAD + c] = ok ? v : 0.f;
    }
    for (int idx = tid; idx < C * 2 * RB_PAD; idx += 512) {
        const int row = idx / (2 * RB_PAD), c = idx - row * (2 * RB_PAD);
        const int col = c < RB_PAD ? c : W0 + c;  // [0, PAD) and [PAD + W0, RS)
        R[row * RS + col] = 0.f;
        S[row * RS + col] = 0.f;
    }
    __syncthreads();

    const int col0 = wave * (NTW * 32) + l31;  // this lane's window column in tile nt = 0

    // one conv out of LDS: dst = f(conv(src) + bias (+ res)), masked to the sequence
    auto conv_phase = [&](const float* src, float* dst, const float* res,  // dst may alias res (in-place residual)
                          const float* __restrict__ wfrag, const float* __restrict__ bias, int dil, bool lrelu_in, bool lrelu_out) {
        f32x16 acc[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const f32x4* __restrict__ ap = reinterpret_cast<const f32x4*>(wfrag) + lane;
        const int center = (p.k - 1) / 2;
        f32x4 a_nxt = ap[0], a_cur;
        float bv_cur[4][NTW], bv_nxt[4][NTW];
        auto fetch_b = [&](const float* __restrict__ s0, float (&bv)[4][NTW]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[e][nt] = s0[(2 * e) * RS + nt * 32];
        };
        int it = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            const float* __restrict__ sb = src + (ch * 16 + half) * RS + RB_PAD + col0 - center * dil;
            fetch_b(sb, bv_cur);
            for (int j = 0; j < p.k; ++j) {
                const float* __restrict__ sj = sb + j * dil;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ++it;
                    // software pipeline: next iteration's A group and B values are requested before this iteration's
                    // MFMAs (after the chunk's last iteration the B prefetch reads in-row values that are discarded)
                    a_cur = a_nxt;
                    a_nxt = ap[(size_t)it * 64];  // plans carry one padding group at the end
                    fetch_b(q == 0 ? sj + 8 * RS : sj + dil, bv_nxt);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            const float v = bv_cur[e][nt];
                            const float vs = v * p.slope;
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[e], (lrelu_in && v < 0.f) ? vs : v, acc[nt], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) bv_cur[e][nt] = bv_nxt[e][nt];
                }
            }
        }
        // epilogue into LDS: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int c = col0 + nt * 32;
            const int t = t_base + c;
            const bool tok = t >= 0 && t < Tlim;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < C) {
                    float v = acc[nt][r] + bias[m];
                    if (res) v += res[m * RS + RB_PAD + c];
                    const float vs = v * p.slope;
                    v = (lrelu_out && v < 0.f) ? vs : v;
                    dst[m * RS + RB_PAD + c] = tok ? v : 0.f;
                }
            }
        }
    };

    float* cur = R;
    float* oth = S;
    if (p.type == 1) {
        for (int m = 0; m < p.n_conv; m += 2) {
            conv_phase(cur, oth, nullptr, p.wfrag[m], p.bias[m], p.dil[m], true, true);
            __syncthreads();
            conv_phase(oth, cur, cur, p.wfrag[m + 1], p.bias[m + 1], p.dil[m + 1], false, false);
            __syncthreads();
        }
    } else {
        for (int m = 0; m < p.n_conv; ++m) {
            conv_phase(cur, oth, cur, p.wfrag[m], p.bias[m], p.dil[m], true, false);
            __syncthreads();
            float* tmp = cur; cur = oth; oth = tmp;
        }
    }

    // ---- write the central TT columns into the MRF accumulator -------------------------------------------------
    float* __restrict__ yb = p.y + (size_t)b * C * p.T;
    for (int idx = tid; idx < C * p.TT; idx += 512) {
        const int row = idx / p.TT, c = idx - row * p.TT;
        const int t = tile * p.TT + c;
        if (t < p.T) {
            float v = cur[row * RS + RB_PAD + p.H + c];
            const size_t o = (size_t)row * p.T + t;
            if (p.epi == EPI_ADD) v = yb[o] + v;
            else if (p.epi == EPI_ADD_DIV) v = (yb[o] + v) / p.div;
            yb[o] = v;
        }
    }
}

// ---- 16-channel variant on v_mfma_f32_16x16x4_f32 (no zero-padded MFMA rows; see conv_mfma16.h for the fragment
// layouts).  Window 1024 columns, 8 waves x 8 tiles of 16; LDS row stride 16 (mod 32) dwords.
constexpr int RB16_PAD = 40;
static __global__ __launch_bounds__(512, 1) void resblock_fused16_kernel(const ResblockParams p) {
    constexpr int NT16 = 8;
    constexpr int W0 = 8 * NT16 * 16;      // 1024
    constexpr int RS = W0 + 2 * RB16_PAD;  // 1104 = 16 (mod 32)
    static_assert(RS % 32 == 16, "row stride must be 16 mod 32 dwords");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // R[16][RS], S[16][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4;
    const int l15 = lane & 15;
    const int b = blockIdx.x / p.tiles;
    const int tile = blockIdx.x - b * p.tiles;
    const int t_base = tile * p.TT - p.H;
    const int Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
    constexpr int C = 16;

    float* R = smem;
    float* S = smem + C * RS;
    const float* __restrict__ xb = p.x + (size_t)b * C * p.T;
    for (int idx = tid; idx < C * W0; idx += 512) {
        const int row = idx / W0, c = idx - row * W0;
        const int t = t_base + c;
        const bool ok = t >= 0 && t < Tlim;
        const float v = xb[(size_t)row * p.T + (ok ? t : 0)];
        R[row * RS + RB16_PAD + c] = ok ? v : 0.f;
    }
    for (int idx = tid; idx < C * 2 * RB16_PAD; idx += 512) {
        const int row = idx / (2 * RB16_PAD), c = idx - row * (2 * RB16_PAD);
        const int col = c < RB16_PAD ? c : W0 + c;
        R[row * RS + col] = 0.f;
        S[row * RS + col] = 0.f;
    }
    __syncthreads();

    const int col0 = wave * (NT16 * 16) + l15;
    auto conv_phase = [&](const float* src, float* dst, const float* res, const float* __restrict__ wfrag,
                          const float* __restrict__ bias, int dil, bool lrelu_in, bool lrelu_out) {
        f32x4 acc[NT16];
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] = 0.f;
        const f32x4* __restrict__ ap = reinterpret_cast<const f32x4*>(wfrag) + lane;
        const int center = (p.k - 1) / 2;
        f32x4 a_nxt = ap[0], a_cur;
        float bv_cur[4][NT16], bv_nxt[4][NT16];
        const float* __restrict__ sb = src + kg * RS + RB16_PAD + col0 - center * dil;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt) bv_cur[e][nt] = sb[(4 * e) * RS + nt * 16];
        for (int j = 0; j < p.k; ++j) {
            const float* __restrict__ sn = sb + (j + 1) * dil;  // next tap (after the last: in-row, discarded)
            a_cur = a_nxt;
            a_nxt = ap[(size_t)(j + 1) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) bv_nxt[e][nt] = sn[(4 * e) * RS + nt * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) {
                    const float v = bv_cur[e][nt];
                    const float vs = v * p.slope;
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[e], (lrelu_in && v < 0.f) ? vs : v, acc[nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) bv_cur[e][nt] = bv_nxt[e][nt];
        }
        // C/D layout of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) {
            const int c = col0 + nt * 16;
            const int t = t_base + c;
            const bool tok = t >= 0 && t < Tlim;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * kg + r;
                float v = acc[nt][r] + bias[m];
                if (res) v += res[m * RS + RB16_PAD + c];
                const float vs = v * p.slope;
                v = (lrelu_out && v < 0.f) ? vs : v;
                dst[m * RS + RB16_PAD + c] = tok ? v : 0.f;
            }
        }
    };

    float* cur = R;
    float* oth = S;
    if (p.type == 1) {
        for (int m = 0; m < p.n_conv; m += 2) {
            conv_phase(cur, oth, nullptr, p.wfrag[m], p.bias[m], p.dil[m], true, true);
            __syncthreads();
            conv_phase(oth, cur, cur, p.wfrag[m + 1], p.bias[m + 1], p.dil[m + 1], false, false);
            __syncthreads();
        }
    } else {
        for (int m = 0; m < p.n_conv; ++m) {
            conv_phase(cur, oth, cur, p.wfrag[m], p.bias[m], p.dil[m], true, false);
            __syncthreads();
            float* tmp = cur; cur = oth; oth = tmp;
        }
    }
    float* __restrict__ yb = p.y + (size_t)b * C * p.T;
    for (int idx = tid; idx < C * p.TT; idx += 512) {
        const int row = idx / p.TT, c = idx - row * p.TT;
        const int t = tile * p.TT + c;
        if (t < p.T) {
            float v = cur[row * RS + RB16_PAD + p.H + c];
            const size_t o = (size_t)row * p.T + t;
            if (p.epi == EPI_ADD) v = yb[o] + v;
            else if (p.epi == EPI_ADD_DIV) v = (yb[o] + v) / p.div;
            yb[o] = v;
        }
    }
}

template <int NTW>
inline hipError_t launch_resblock_fused_t(const ResblockParams& p, hipStream_t s) {
    constexpr int RS = 8 * NTW * 32 + 2 * RB_PAD;
    const size_t lds = (size_t)2 * p.C * RS * sizeof(float);
    auto kern = resblock_fused_kernel<NTW>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles * p.B), dim3(512), lds, s, p);
    return hipGetLastError();
}

// window width for a channel count: the two fp32 buffers must fit 160 KiB of LDS
inline int resblock_window(int C) { return C <= 16 ? 1024 : 512; }
// C == 16: the plans are 16-row packs (tile config 6) and the 16x16x4 kernel runs; C == 32: 32-row packs (config 2)
inline hipError_t launch_resblock_fused(const ResblockParams& p, hipStream_t s) {
    if (p.C == 16) {
        const size_t lds = (size_t)2 * 16 * (1024 + 2 * RB16_PAD) * sizeof(float);
        static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
        {
            hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(resblock_fused16_kernel), (size_t)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(resblock_fused16_kernel, dim3(p.tiles * p.B), dim3(512), lds, s, p);
        return hipGetLastError();
    }
    return launch_resblock_fused_t<2>(p, s);
}

}  // namespace parrot
