// attn.h -- batched fp32 MFMA GEMM for the attention score (Q K^T) and context (P V) products of the
// torch MHA math path (reference modules/fft.py:56 -> F.multi_head_attention_forward).
//
//   C[z][m][n] = sum_k (alpha * A[z](k,m)) * B[z](k,n),  z = (batch, head)
// Operands are addressed with explicit (k, m|n) strides so the same kernel serves
//   scores : A = q (channel-first, k = head channel, m = t_q),  B = k (k = channel, n = t_k)
//   context: A = v (m = head channel, k = t_k),                 B = P (n = t_q, k = t_k)
// without any transposed copies.  Tiles are staged K-major in LDS ([k][m], [k][n], row stride 65 so the
// transposing writes of the second form are bank-conflict free) and fed to v_mfma_f32_32x32x2_f32.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"

namespace parrot {

struct BgemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long a_sk, a_sm, b_sk, b_sn;      // element strides
    long a_zb, a_zh, b_zb, b_zh;      // batch / head offsets
    long c_zb, c_zh, ldc;
    int H;
    float alpha;
};

__global__ __launch_bounds__(256) void bgemm_mfma_kernel(const BgemmParams p) {
    constexpr int KC = 32, RS = 65;
    __shared__ float As[KC][RS];
    __shared__ float Bs[KC][RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int z = blockIdx.z, zb = z / p.H, zh = z - zb * p.H;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* __restrict__ A = p.A + zb * p.a_zb + zh * p.a_zh;
    const float* __restrict__ B = p.B + zb * p.b_zb + zh * p.b_zh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const bool a_kfast = (p.a_sk == 1), b_kfast = (p.b_sk == 1);
    for (int k0 = 0; k0 < p.K; k0 += KC) {
#pragma unroll
        for (int i = 0; i < (KC * 64) / 256; ++i) {
            const int idx = i * 256 + tid;
            int kk, mm;
            if (a_kfast) { mm = idx / KC; kk = idx % KC; } else { kk = idx / 64; mm = idx % 64; }
            float v = 0.f;
            if (k0 + kk < p.K && m0 + mm < p.M) v = A[(long)(k0 + kk) * p.a_sk + (long)(m0 + mm) * p.a_sm] * p.alpha;
            As[kk][mm] = v;
            int kb, nn;
            if (b_kfast) { nn = idx / KC; kb = idx % KC; } else { kb = idx / 64; nn = idx % 64; }
            float w = 0.f;
            if (k0 + kb < p.K && n0 + nn < p.N) w = B[(long)(k0 + kb) * p.b_sk + (long)(n0 + nn) * p.b_sn];
            Bs[kb][nn] = w;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC; ks += 2) {
            const float a = As[ks + half][wm * 32 + l31];
            const float b = Bs[ks + half][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* __restrict__ C = p.C + zb * p.c_zb + zh * p.c_zh;
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M && n < p.N) C[(long)m * p.ldc + n] = acc[r];
    }
}

// MFMA fragment-layout probe: D = A(32x2) * B(2x32) with A[i][k] = i + 100k, B[k][j] = (k ? 1000 : 1) * (j+1)
// dumps the 16 accumulator registers of every lane so the host can check the assumed C/D mapping.
__global__ void mfma_probe_kernel(float* out) {
    const int lane = threadIdx.x;
    const int i = lane & 31, k = lane >> 5;
    const float a = (float)(i + 100 * k);
    const float b = (k ? 1000.f : 1.f) * (float)(i + 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

}  // namespace parrot
