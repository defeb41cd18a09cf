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
//   * A (weights): split on the host and packed as [m_tile][chunk*tap][piece][lane][8 bf16]; streamed from L2 with
//     one global_load_dwordx4 per (m_tile, piece) per step, prefetched one step ahead.
//   * accumulator init / epilogue are shared with the exact kernel (same 32x32 C/D layout).
#pragma once
#include "conv_mfma.h"

namespace parrot {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {  // round-to-nearest-even, finite inputs
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// x -> three bf16 pieces (bit patterns) with x ~= p0 + p1 + p2
__device__ __forceinline__ void split3(float x, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = bf16_rn_bits(x);
    const float r1 = x - __uint_as_float(p0 << 16);
    p1 = bf16_rn_bits(r1);
    const float r2 = r1 - __uint_as_float(p1 << 16);
    p2 = bf16_rn_bits(r2);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int MINW>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_bf16x6_kernel(const ConvParams p) {
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int COLS = BN + CONV_HALO;           // staged columns per chunk
    constexpr int PIECE_BYTES = COLS * 32;         // [col][16 ch] bf16
    constexpr int BUF_BYTES = 3 * PIECE_BYTES;
    constexpr int ITEMS = (COLS * 2 + NT - 1) / NT;  // (column, channel-octet) items per thread

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
    const int grp = (p.groups > 1) ? (blockIdx.y * BM) / p.Mg : 0;
    const float* __restrict__ xb = p.x + (size_t)b * p.x_bstride + (size_t)grp * p.Cin * p.Tin;

    float stage[ITEMS][8];
    auto load_slab = [&](int c) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int item = tid + NT * i;
            const int oct = item / COLS;          // 0 / 1 (>= 2: idle slot)
            const int col = item - oct * COLS;
            const int tin = t0 - p.pad_left + col;
            const bool colok = oct < 2 && col < W && tin >= 0 && tin < p.Tin;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c * 16 + oct * 8 + e;
                const bool ok = colok && ch < p.Cin;
                float v = xb[ok ? (size_t)ch * p.Tin + tin : 0];
                v = ok ? v : 0.f;
                const float vs = v * p.pre_slope;
                stage[i][e] = (p.pre == PRE_LRELU && v < 0.f) ? vs : v;
            }
        }
    };
    auto store_slab = [&](int buf) {
        char* dst = smem_raw + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int item = tid + NT * i;
            const int oct = item / COLS;
            const int col = item - oct * COLS;
            unsigned q0[8], q1[8], q2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split3(stage[i][e], q0[e], q1[e], q2[e]);
            if (oct < 2) {
                uint4 v0, v1, v2;
                v0.x = q0[0] | (q0[1] << 16); v0.y = q0[2] | (q0[3] << 16); v0.z = q0[4] | (q0[5] << 16); v0.w = q0[6] | (q0[7] << 16);
                v1.x = q1[0] | (q1[1] << 16); v1.y = q1[2] | (q1[3] << 16); v1.z = q1[4] | (q1[5] << 16); v1.w = q1[6] | (q1[7] << 16);
                v2.x = q2[0] | (q2[1] << 16); v2.y = q2[2] | (q2[3] << 16); v2.z = q2[4] | (q2[5] << 16); v2.w = q2[6] | (q2[7] << 16);
                const int off = col * 32 + oct * 16;
                *reinterpret_cast<uint4*>(dst + off) = v0;
                *reinterpret_cast<uint4*>(dst + PIECE_BYTES + off) = v1;
                *reinterpret_cast<uint4*>(dst + 2 * PIECE_BYTES + off) = v2;
            }
        }
    };

    f32x16 acc[WM][WN];
    const int m_wave = blockIdx.y * BM + wm * WM * 32;
    const int n_wave = t0 + wn * WN * 32;
    conv_acc_init<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);

    // A stream: n_it = nchunks * k steps per m-tile, 3 KiB per step ([piece][lane][8 bf16])
    const char* __restrict__ aptr[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int mtg = blockIdx.y * (BM / 32) + wm * WM + mt;
        aptr[mt] = reinterpret_cast<const char*>(p.wfrag) + ((size_t)mtg * p.n_it) * 3072 + lane * 16;
    }
    s16x8 a_cur[WM][3], a_nxt[WM][3], b_cur[WN][3], b_nxt[WN][3];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a_nxt[mt][pc] = *reinterpret_cast<const s16x8*>(aptr[mt] + pc * 1024);

    load_slab(0);
    store_slab(0);
    __syncthreads();

    const int bbase = (wn * WN * 32 + l31) * 32 + half * 16;  // byte offset of this lane's fragment at tap 0
    int it = 0;
    for (int c = 0; c < p.nchunks; ++c) {
        const char* __restrict__ xs = smem_raw + (c & 1) * BUF_BYTES + bbase;
        const bool more = (c + 1 < p.nchunks);
        if (more) load_slab(c + 1);
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) b_cur[nt][pc] = *reinterpret_cast<const s16x8*>(xs + pc * PIECE_BYTES + nt * 1024);
        for (int j = 0; j < p.k; ++j) {
            ++it;
            // prefetch the next step's operands (after the chunk's last tap the B prefetch reads in-slab garbage
            // that is discarded; wfrag carries one padding step at the end)
            const char* __restrict__ xn = xs + (j + 1) * p.dil * 32;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    a_cur[mt][pc] = a_nxt[mt][pc];
                    a_nxt[mt][pc] = *reinterpret_cast<const s16x8*>(aptr[mt] + (size_t)it * 3072 + pc * 1024);
                }
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b_nxt[nt][pc] = *reinterpret_cast<const s16x8*>(xn + pc * PIECE_BYTES + nt * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt) {
                    // smallest terms first
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
                    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[mt][PA[t]]),
                                                                              __builtin_bit_cast(bf16x8, b_cur[nt][PB[t]]), acc[mt][nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b_cur[nt][pc] = b_nxt[nt][pc];
        }
        if (more) store_slab((c + 1) & 1);
        __syncthreads();
    }
    conv_epilogue<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
}

// bf16x6 tile table: index = the exact kernel's tile_cfg it stands in for (0: 128x128, 1: 64x256)
template <int WAVES_M, int WAVES_N, int WM, int WN, int MINW>
inline hipError_t launch_conv_bf16x6_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    constexpr int BN = WAVES_N * WN * 32;
    const size_t lds = (size_t)2 * 3 * (BN + CONV_HALO) * 32;
    auto kern = conv_bf16x6_kernel<WAVES_M, WAVES_N, WM, WN, MINW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    return hipGetLastError();
}

inline hipError_t launch_conv_bf16x6(int cfg, const ConvParams& p, hipStream_t s) {
    const TileCfg t = tile_cfg(cfg);
    dim3 grid(p.tiles_n * p.B, (p.M + t.bm - 1) / t.bm);
    if (cfg == 1) return launch_conv_bf16x6_t<1, 4, 2, 2, 2>(p, grid, s);
    return launch_conv_bf16x6_t<2, 2, 2, 2, 2>(p, grid, s);
}

}  // namespace parrot
