// conv_valu.h -- the two narrowest layers of the vocoder as plain fp32 FMA streaming kernels.  conv_post (16 -> 1
// channel, k = 7) and the last upsampling ConvTranspose1d (32 -> 16 channels, k = 4, stride 2) move 0.36 / 0.67 GB
// for 1.2 / 10.7 GFLOP: on the MFMA tile kernels they ran at 1.1-1.2 TB/s because 15/16 resp. half of every MFMA
// multiplied zero rows.  Here a lane owns output time steps, inputs arrive through buffer loads whose out-of-row
// offsets return 0 (= zero padding), weights are wave-uniform (scalar loads), and tanh is fused.
#pragma once
#include "conv_mfma.h"

namespace parrot {

struct ConvValuParams {
    const float* x;      // (B, Cin, Tin)
    const float* w;      // Conv1d: (1, Cin, K);  ConvTranspose1d: (Cin, COUT, K)
    const float* bias;
    float* y;            // (B, COUT, Tout)
    int B, Cin, Tin, Tout;
    float slope;         // leaky-ReLU on the input (1 = none)
    int act;             // ACT_NONE / ACT_TANH
    const int32_t* row_len;
    int row_len_mul;
    int row_len_add;         // true length of a row of n > 0 units at this layer = n * row_len_mul + row_len_add (odd k - u upsampling stages add samples)
    int* err;            // optional device flag: set to 5 when a tanh layer's pre-activation is NaN / inf (it reached the waveform)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t valu_row_rsrc(const float* base) {
    const size_t a = reinterpret_cast<size_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// Linear(C -> 1) over (B, C, T) (the duration predictor's projection, reference modules/duration.py:39): lane = time step
// (coalesced 256-byte rows), the four waves of a workgroup take a quarter of the channels each and their partial sums are added
// in wave order; the generic kernel below gives a short sequence (S = 64) sixteen busy threads with 256-deep load chains (76 us).
__global__ __launch_bounds__(256) void linear1_valu_kernel(const ConvValuParams p) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + lane;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const __amdgpu_buffer_rsrc_t xr = valu_row_rsrc(p.x + (size_t)b * p.Cin * p.Tin);
    const int voff = (t < Tlim) ? t * 4 : (int)0x80000000;
    const int per = (p.Cin + 3) / 4, c0 = wave * per, c1 = min(p.Cin, c0 + per);
    const int row_bytes = p.Tin * 4;
    float acc = 0.f;
    for (int c = c0; c < c1; ++c) {
        const float xv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff, c * row_bytes, 0));
        acc = fmaf(p.w[c], fmaxf(xv, xv * p.slope), acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && t < p.Tout) {
        const float v = (((p.bias ? p.bias[0] : 0.f) + part[0][lane]) + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (p.act == ACT_TANH && p.err && !(fabsf(v) < INFINITY)) atomicExch(p.err, 5);
        p.y[(size_t)b * p.Tout + t] = (p.act == ACT_TANH) ? tanhf(v) : v;
    }
}

// Conv1d with ONE output channel, dilation 1, padding (K-1)/2: four consecutive outputs per thread.
template <int K>
__global__ __launch_bounds__(256) void conv1_valu_kernel(const ConvValuParams p) {
    constexpr int NV = 4 + K - 1;
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const __amdgpu_buffer_rsrc_t xr = valu_row_rsrc(p.x + (size_t)b * p.Cin * p.Tin);
    int voff[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int t = t0 - (K - 1) / 2 + i;
        voff[i] = (t >= 0 && t < Tlim) ? t * 4 : (int)0x80000000;
    }
    const float b0 = p.bias ? p.bias[0] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    const int row_bytes = p.Tin * 4;
    for (int c = 0; c < p.Cin; ++c) {
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float xv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff[i], c * row_bytes, 0));
            v[i] = fmaxf(xv, xv * p.slope);  // leaky ReLU for 0 <= slope <= 1
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float wj = p.w[c * K + j];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(wj, v[o + j], acc[o]);
        }
    }
    float* yb = p.y + (size_t)b * p.Tout;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float r = (p.act == ACT_TANH) ? tanhf(acc[o]) : acc[o];
        if (p.act == ACT_TANH && p.err && !(fabsf(acc[o]) < INFINITY)) atomicExch(p.err, 5);  // (pre-tanh: tanhf(inf) = 1 would pass)
        if (t0 + o < p.Tout) yb[t0 + o] = r;
    }
}

// The same for K = 7 with 16-byte loads (rows that are a multiple of 4 samples long): the ten inputs of a thread's four
// outputs are the three aligned quads at t0 - 4, t0, t0 + 4 -- 3 buffer_load_dwordx4 per channel instead of 10 strided
// dword loads (the texture-address path, not HBM, bounded the dword version at 1.9 TB/s).  A quad is either inside
// the row or entirely outside it (-> 0); only a ragged row end (Tlim % 4 != 0) needs per-sample masking.
static __global__ __launch_bounds__(256) void conv1_valu7_vec_kernel(const ConvValuParams p) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const __amdgpu_buffer_rsrc_t xr = valu_row_rsrc(p.x + (size_t)b * p.Cin * p.Tin);
    int voff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int t = t0 + 4 * (q - 1);
        voff[q] = (t >= 0 && t < Tlim) ? t * 4 : (int)0x80000000;
    }
    const bool partial = (Tlim & 3) != 0;  // the quad that straddles the row end carries samples of the padding region
    const float b0 = p.bias ? p.bias[0] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    const int row_bytes = p.Tin * 4;
    for (int c = 0; c < p.Cin; ++c) {
        float v[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const f32x4v xv = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[q], c * row_bytes, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xe = xv[e];
                if (partial) xe = (t0 + 4 * (q - 1) + e < Tlim) ? xe : 0.f;
                v[4 * q + e] = fmaxf(xe, xe * p.slope);
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float wj = p.w[c * 7 + j];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(wj, v[o + j + 1], acc[o]);  // input t0 + o + j - 3 = v[(o + j - 3) + 4]
        }
    }
    f32x4v r;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        r[o] = (p.act == ACT_TANH) ? tanhf(acc[o]) : acc[o];
        if (p.act == ACT_TANH && p.err && !(fabsf(acc[o]) < INFINITY)) atomicExch(p.err, 5);  // (pre-tanh: tanhf(inf) = 1 would pass)
    }
    if (t0 + 3 < p.Tout) *reinterpret_cast<f32x4v*>(p.y + (size_t)b * p.Tout + t0) = r;
    else
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (t0 + o < p.Tout) p.y[(size_t)b * p.Tout + t0 + o] = r[o];
}

// ConvTranspose1d, COUT output channels, kernel K, stride U, padding PAD, in polyphase form: a lane owns input position
// n and produces outputs t = n*U + ph, ph < U, for every output channel:
//     y[o][n*U + ph] = b[o] + sum_c sum_{kap : (ph + PAD - kap) % U == 0} w[c][o][kap] * pre(x[c][n + (ph + PAD - kap) / U])
template <int COUT, int K, int U, int PAD>
__global__ __launch_bounds__(256) void convt_valu_kernel(const ConvValuParams p) {
    constexpr int DLO = -((K - 1 - PAD + U - 1) / U), DHI = (U - 1 + PAD) / U, ND = DHI - DLO + 1;  // input offsets used
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;
    const __amdgpu_buffer_rsrc_t xr = valu_row_rsrc(p.x + (size_t)b * p.Cin * p.Tin);
    int voff[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int t = n + DLO + i;
        voff[i] = (t >= 0 && t < Tlim) ? t * 4 : (int)0x80000000;
    }
    float acc[COUT][U];
#pragma unroll
    for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int ph = 0; ph < U; ++ph) acc[o][ph] = p.bias ? p.bias[o] : 0.f;
    const int row_bytes = p.Tin * 4;
    for (int c = 0; c < p.Cin; ++c) {
        float v[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const float xv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff[i], c * row_bytes, 0));
            v[i] = fmaxf(xv, xv * p.slope);
        }
        const float* __restrict__ wc = p.w + (size_t)c * COUT * K;  // wave-uniform: scalar loads
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
            for (int ph = 0; ph < U; ++ph)
#pragma unroll
                for (int kap = 0; kap < K; ++kap)
                    if ((ph + PAD - kap) % U == 0) acc[o][ph] = fmaf(wc[o * K + kap], v[(ph + PAD - kap) / U - DLO], acc[o][ph]);
    }
    if (n < p.Tin) {
        float* yb = p.y + (size_t)b * COUT * p.Tout;
        if (U == 2 && (p.Tout & 1) == 0 && n * 2 + 1 < p.Tout) {  // the two phases of a lane are adjacent samples: 8-byte stores
            typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int o = 0; o < COUT; ++o) *reinterpret_cast<f32x2v*>(yb + (size_t)o * p.Tout + 2 * n) = f32x2v{acc[o][0], acc[o][U - 1]};
        } else {
#pragma unroll
            for (int o = 0; o < COUT; ++o)
#pragma unroll
                for (int ph = 0; ph < U; ++ph) {
                    const int t = n * U + ph;
                    if (t < p.Tout) yb[(size_t)o * p.Tout + t] = acc[o][ph];
                }
        }
    }
}

}  // namespace parrot
