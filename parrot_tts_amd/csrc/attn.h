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
#include "kernels_misc.h"

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

// ---------------------------------------------------------------------------------------------
// Fused attention core for sequences of <= 256 steps: scores, key-padding softmax and context in ONE launch per
// (batch, head, QT queries) instead of bgemm -> softmax_mask -> bgemm with the (B, H, T, T) score tensor going
// through HBM twice.  Same arithmetic as the three-kernel path, in the same order (v_mfma_f32_32x32x2_f32 over
// ascending channel / key pairs; the softmax row reductions of softmax_mask_kernel), so the results are bit-identical:
//   phase 1  S[tq][tk] = sum_c (alpha q[c][tq]) k[c][tk]  -> LDS, 64 x T            (32x32 tiles dealt round-robin to waves)
//   softmax  P = softmax(S + key mask) per query row, in LDS                        (16 rows per wave)
//   phase 2  ctx[c][tq] = sum_tk v[c][tk] P[tq][tk]                                 (wave w: channels 32w.., both query tiles)
// q / k / v / ctx are channel-first (hd, T) slices of the (B, 3, D, T) projection buffer / (B, D, T) context buffer.
struct AttnParams {
    const float* qkv;      // (B, 3, D, T)
    const uint8_t* valid;  // (B, T) 1 = real key
    float* ctx;            // (B, D, T)
    int T, H, D, hd;
    float alpha;
};
constexpr int ATTN_TMAX = 256;

template <int HD, int QT>
__global__ __launch_bounds__(256) void attn_fused_kernel(const AttnParams p) {
    constexpr int MT = QT / 32;    // query tiles of 32 per workgroup (QT = 32: 33 KiB of LDS -> 4 workgroups per CU)
    extern __shared__ float Ps[];  // [QT][TP]
    constexpr int KB = 8;          // MFMA k-steps per operand batch (operands are fetched a batch ahead)
    const int T = p.T, ntile = (T + 31) / 32, TP = ntile * 32 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int z = blockIdx.y, b = z / p.H, h = z - b * p.H;
    const int tq0 = blockIdx.x * QT;
    const long DT = (long)p.D * T;
    const float* __restrict__ q = p.qkv + (long)b * 3 * DT + (long)h * HD * T;
    const float* __restrict__ k = q + DT;
    const float* __restrict__ v = q + 2 * DT;

    // ---- phase 1: 2 query tiles x ntile key tiles of 32x32, round-robin over the four waves ---------------------
    for (int pair = wave; pair < MT * ntile; pair += 4) {
        const int mt = pair % MT, nt = pair / MT;
        const int tq = tq0 + mt * 32 + l31, tk = nt * 32 + l31;
        const bool qok = tq < T, kok = tk < T;
        const float* __restrict__ qp = q + (qok ? tq : 0) + (long)half * T;
        const float* __restrict__ kp = k + (kok ? tk : 0) + (long)half * T;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float a[2][KB], bb[2][KB];
        auto fetch = [&](int set, int c0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                a[set][i] = qp[(long)(c0 + 2 * i) * T];
                bb[set][i] = kp[(long)(c0 + 2 * i) * T];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int it = 0; it < HD / (2 * KB); ++it) {
            if (it + 1 < HD / (2 * KB)) fetch((it + 1) & 1, (it + 1) * 2 * KB);
#pragma unroll
            for (int i = 0; i < KB; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qok ? a[it & 1][i] * p.alpha : 0.f, kok ? bb[it & 1][i] : 0.f, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ps[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TP + tk] = acc[r];
    }
    __syncthreads();

    // ---- softmax over the keys, masked by key validity (as softmax_mask_kernel) -------------------------------
    const uint8_t* __restrict__ kv = p.valid + (long)b * T;
    for (int row = wave * (QT / 4); row < (wave + 1) * (QT / 4); ++row) {
        if (tq0 + row >= T) break;
        float* sr = Ps + row * TP;
        float mx = -INFINITY;
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, kv[t] ? sr[t] : -INFINITY);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float e = kv[t] ? expf(sr[t] - mx) : 0.f;
            sr[t] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int t = lane; t < T; t += 64) sr[t] = sr[t] / sum;
        for (int t = T + lane; t < ntile * 32; t += 64) sr[t] = 0.f;  // padded keys contribute nothing
    }
    __syncthreads();

    // ---- phase 2: channel tiles of 32, both query tiles; keys in batches of 2*KB (the padded P columns are 0) -----
    float* __restrict__ ctx = p.ctx + (long)b * DT + (long)h * HD * T;
    const int nbatch = (T + 2 * KB - 1) / (2 * KB);
    for (int ct = wave; ct < HD / 32; ct += 4) {
        const float* __restrict__ vp = v + (long)(ct * 32 + l31) * T + half;
        const float* __restrict__ p0 = Ps + l31 * TP + half;
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        float a[2][KB];
        auto fetch = [&](int set, int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int t = t0 + 2 * i;
                a[set][i] = (t + half < T) ? vp[t] : 0.f;
            }
        };
        fetch(0, 0);
        for (int it = 0; it < nbatch; it += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int t0 = (it + sub) * 2 * KB;
                if (t0 < T) {
                    fetch(sub ^ 1, t0 + 2 * KB);  // (past the end: zeros)
#pragma unroll
                    for (int i = 0; i < KB; ++i)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sub][i], p0[m * 32 * TP + t0 + 2 * i], acc[m], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = (long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * T;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (tq0 + m * 32 + l31 < T) ctx[row + tq0 + m * 32 + l31] = acc[m][r];
        }
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
