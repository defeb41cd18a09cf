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
#include "conv_split.h"
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

static __global__ __launch_bounds__(256) void bgemm_mfma_kernel(const BgemmParams p) {
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

// ---------------------------------------------------------------------------------------------
// Flash-style attention core for ANY sequence length on the fp16 split pipe (SchF16x3: 3 fp16 MFMAs per product group,
// fp32 accumulate -- conv_split.h): softmax(q k^T * hd^-1/2 + key mask) v of the torch MHA math path (fft.py:56)
// without ever materialising a (T, T) score tensor.  One workgroup = 128 queries of one (batch, head); wave w owns
// queries [32w, 32w+32) and walks the keys in tiles of 32 with an online softmax:
//
//   S^T[key][query] = sum_c K[c][key] (alpha q[c][query])       A = K tile (LDS, [key][channel] fp16 pieces), B = Q (registers)
//   m, l, O updated per tile:  m' = max(m, max_key S);  O *= e^(m-m');  l = l e^(m-m') + sum_key e^(S-m')
//   O[c][query] += sum_key V[c][key] P[query][key]              A = V tile (LDS, [channel][key] fp16 pieces), B = P (registers)
//
// With queries on the MFMA column axis a lane owns ONE query: its 16 accumulator registers of S^T are 16 keys of that
// query (the other 16 sit in lane ^ 32), so the softmax statistics are per-lane scalars, and the probabilities feed the
// second GEMM straight from registers -- the C/D layout of S^T is the B-operand layout of P up to a fixed permutation of
// the keys inside a k-step, which the V fragments are read with as well.  K / V tiles are staged (global -> fp16 pieces
// -> LDS) once per workgroup and shared by its four waves, double buffered, one barrier per tile.
// Scales (powers of two): q, k, v pieces carry 2^3, P pieces 2^11; undone on the scores / in the final normalisation.
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_flash_kernel(const AttnParams p) {
    using SCH = SchF16x3;
    constexpr int KS = HD / 16;            // k-steps over the head channels (scores)
    constexpr int MT = (HD + 31) / 32;     // 32-channel tiles of the context
    constexpr int KT = 32;                 // keys per tile
    constexpr int KROW = HD * 2 + 16;      // bytes per key row of a K piece (+16: conflict-free ds_read_b128 across keys)
    constexpr int VROW = KT * 2 + 8;       // bytes per channel row of a V piece
    constexpr int KBYTES = KT * KROW, VBYTES = MT * 32 * VROW;
    constexpr int BUF = 2 * (KBYTES + VBYTES);  // [K p0][K p1][V p0][V p1]
    constexpr float PS = 2048.f;           // scale of the probability pieces
    static_assert(HD % 16 == 0 && HD <= 128, "head dim: multiple of 16, <= 128");
    extern __shared__ __attribute__((aligned(16))) char fsm[];  // [2][BUF]

    const int T = p.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int z = blockIdx.y, b = z / p.H, h = z - b * p.H;
    const int tq = blockIdx.x * 128 + wave * 32 + l31;  // this lane's query
    const long DT = (long)p.D * T;
    const float* __restrict__ q = p.qkv + (long)b * 3 * DT + (long)h * HD * T;
    const float* __restrict__ k = q + DT;
    const float* __restrict__ v = q + 2 * DT;
    const uint8_t* __restrict__ kv = p.valid + (long)b * T;

    // ---- Q fragments (B operand: lane = query, 8 channels 16 ks + 8 half ..): registers for the whole kernel --------
    s16x8 Q[KS][2];
    {
        const float* __restrict__ qp = q + min(tq, T - 1);
        const float qs = p.alpha * SCH::XS;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned w[4][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 16 * ks + 8 * half + 2 * e;
                SCH::split(qp[(long)c * T] * qs, qp[(long)(c + 1) * T] * qs, w[e]);
            }
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) Q[ks][pc] = __builtin_bit_cast(s16x8, uint4{w[0][pc], w[1][pc], w[2][pc], w[3][pc]});
        }
    }

    // ---- K / V tile staging ---------------------------------------------------------------------------------------
    // K: thread (t = tid & 31, channel pair tid >> 5 + 8 i): two coalesced row loads -> packed fp16 pair at [t][c, c+1]
    // V: thread (key pair tid & 15, channel tid >> 4 + 16 i): two loads -> packed pair at [c][t, t+1]
    constexpr int KI = HD / 16, VI = MT * 2;
    float kraw[KI][2], vraw[VI][2];
    auto load_k = [&](int t0) __attribute__((always_inline)) {
        const int t = t0 + (tid & 31);
        const bool ok = t < T;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int c = 2 * ((tid >> 5) + 8 * i);
            kraw[i][0] = ok ? k[(long)c * T + t] : 0.f;
            kraw[i][1] = ok ? k[(long)(c + 1) * T + t] : 0.f;
        }
    };
    auto store_k = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int c = 2 * ((tid >> 5) + 8 * i);
            unsigned w[2];
            SCH::split(kraw[i][0] * SCH::XS, kraw[i][1] * SCH::XS, w);
            char* dst = buf + (tid & 31) * KROW + c * 2;
            *reinterpret_cast<unsigned*>(dst) = w[0];
            *reinterpret_cast<unsigned*>(dst + KBYTES) = w[1];
        }
    };
    auto load_v = [&](int t0) __attribute__((always_inline)) {
        const int t = t0 + 2 * (tid & 15);
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int c = (tid >> 4) + 16 * i;
            const bool cok = c < HD;
            vraw[i][0] = (cok && t < T) ? v[(long)c * T + t] : 0.f;
            vraw[i][1] = (cok && t + 1 < T) ? v[(long)c * T + t + 1] : 0.f;
        }
    };
    auto store_v = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int c = (tid >> 4) + 16 * i;
            unsigned w[2];
            SCH::split(vraw[i][0] * SCH::XS, vraw[i][1] * SCH::XS, w);
            char* dst = buf + 2 * KBYTES + c * VROW + (tid & 15) * 4;
            *reinterpret_cast<unsigned*>(dst) = w[0];
            *reinterpret_cast<unsigned*>(dst + VBYTES) = w[1];
        }
    };

    f32x16 O[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[mt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    load_k(0);
    load_v(0);
    store_k(fsm);
    store_v(fsm);
    __syncthreads();

    const int ntile = (T + KT - 1) / KT;
    for (int kt = 0; kt < ntile; ++kt) {
        const char* __restrict__ cur = fsm + (kt & 1) * BUF;
        char* nxt = fsm + ((kt + 1) & 1) * BUF;
        const int t0 = kt * KT;
        const bool more = kt + 1 < ntile;
        // key validity of this tile as a wave-uniform bit mask (key padding mask of the batch row; keys past T)
        const unsigned long long vb = __ballot((lane < KT) && (t0 + lane < T) && kv[min(t0 + lane, T - 1)] != 0);
        const unsigned vmask = (unsigned)vb;
        if (more) load_k(t0 + KT);
        // ---- scores: S^T = K^T-tile x Q ---------------------------------------------------------------------------
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {   // K fragments one k-step ahead (explicit double buffer; the sched_barrier keeps hipcc from hoisting all KS fetches)
            const char* src = cur + l31 * KROW + 8 * half * 2;
            s16x8 kf[2][2];
            kf[0][0] = *reinterpret_cast<const s16x8*>(src);
            kf[0][1] = *reinterpret_cast<const s16x8*>(src + KBYTES);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    kf[(ks + 1) & 1][0] = *reinterpret_cast<const s16x8*>(src + (ks + 1) * 32);
                    kf[(ks + 1) & 1][1] = *reinterpret_cast<const s16x8*>(src + (ks + 1) * 32 + KBYTES);
                }
                S = mfma32<SCH>(kf[ks & 1][1], Q[ks][0], S);
                S = mfma32<SCH>(kf[ks & 1][0], Q[ks][1], S);
                S = mfma32<SCH>(kf[ks & 1][0], Q[ks][0], S);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            store_k(nxt);
            load_v(t0 + KT);
        }
        // ---- online softmax (per lane = per query; the two half-waves hold complementary keys) ----------------------
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
            S[r] = ((vmask >> key) & 1u) ? S[r] * (1.f / (SCH::XS * SCH::XS)) : -INFINITY;
            mx = fmaxf(mx, S[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;  // (no valid key yet: every exp below is e^-inf = 0)
        const float resc = expf(m_run - m_safe);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = expf(S[r] - m_safe);  // (in place: S now holds the unnormalised probabilities)
            ps += S[r];
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * resc + ps;
        m_run = m_new;
        if (__any(resc != 1.f)) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[mt][r] *= resc;
        }
        // probabilities -> B fragments of the second GEMM: k-step s holds registers 8s..8s+7, i.e. the keys
        // 16 s + 8 (e >> 2) + 4 half + (e & 3): the V fragments below are read in the same order
        s16x8 P[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            unsigned w[4][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) SCH::split(S[8 * s2 + 2 * e] * PS, S[8 * s2 + 2 * e + 1] * PS, w[e]);
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) P[s2][pc] = __builtin_bit_cast(s16x8, uint4{w[0][pc], w[1][pc], w[2][pc], w[3][pc]});
        }
        // ---- context: O += V-tile x P ------------------------------------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* src = cur + 2 * KBYTES + (mt * 32 + l31) * VROW + (16 * s2 + 4 * half) * 2;
                const uint2 a0 = *reinterpret_cast<const uint2*>(src), a1 = *reinterpret_cast<const uint2*>(src + 16);
                const uint2 c0 = *reinterpret_cast<const uint2*>(src + VBYTES), c1 = *reinterpret_cast<const uint2*>(src + VBYTES + 16);
                const s16x8 v0 = __builtin_bit_cast(s16x8, uint4{a0.x, a0.y, a1.x, a1.y});
                const s16x8 v1 = __builtin_bit_cast(s16x8, uint4{c0.x, c0.y, c1.x, c1.y});
                O[mt] = mfma32<SCH>(v1, P[s2][0], O[mt]);
                O[mt] = mfma32<SCH>(v0, P[s2][1], O[mt]);
                O[mt] = mfma32<SCH>(v0, P[s2][0], O[mt]);
            }
            if (mt & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_v(nxt);
        __syncthreads();
    }

    // ---- ctx[h*HD + c][tq] = O / (l * scales): lanes run along the queries -> 128-byte row segments --------------------
    if (tq < T) {
        const float inv = 1.f / (l_run * (PS * SCH::XS));
        float* __restrict__ ctx = p.ctx + (long)b * DT + (long)h * HD * T + tq;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < HD) ctx[(long)c * T] = O[mt][r] * inv;
            }
    }
}

inline bool attn_flash_has(int hd) { return hd == 128 || hd == 64 || hd == 32 || hd == 16; }
template <int HD>
inline hipError_t launch_attn_flash_t(const AttnParams& p, int B, hipStream_t s) {
    constexpr int MT = (HD + 31) / 32;
    const size_t lds = (size_t)2 * 2 * (32 * (HD * 2 + 16) + MT * 32 * (32 * 2 + 8));
    auto kern = attn_flash_kernel<HD>;
    static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((p.T + 127) / 128, B * p.H), dim3(256), lds, s, p);
    return hipGetLastError();
}
hipError_t launch_attn_flash(const AttnParams& p, int B, hipStream_t s);  // csrc/tu_attn.hip

// MFMA fragment-layout probe: D = A(32x2) * B(2x32) with A[i][k] = i + 100k, B[k][j] = (k ? 1000 : 1) * (j+1)
// dumps the 16 accumulator registers of every lane so the host can check the assumed C/D mapping.
static __global__ void mfma_probe_kernel(float* out) {
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
