// conv_mfma.h -- the dilated Conv1d implicit-GEMM kernel (fp32 in, fp32 accumulate) for gfx950.
//
// One kernel family covers every Conv1d / ConvTranspose1d / Linear on the Parrot-TTS hot path
// (reference utils/vocoder/models.py:17-28,75,81-83,91; modules/fft.py:48-50,65-76;
// modules/duration.py:64-72).  Layout is channel-first (B, C, T), T contiguous, as in the
// reference's vocoder.
//
//   GEMM view per batch row:  Y[M x N] = W[M x K] * X~[K x N],  M = C_out, N = T, K = C_in*k,
//   X~[(i,j)][t] = pre(x[i][t + j*dil - pad])  (never materialised: an x slab with halo sits in LDS
//   and every tap is a shifted ds_read of the same slab).
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = the fp32 peak
//     (157 TF); A and B operands are ONE f32 VGPR per lane, so LDS/L2 operand traffic per MFMA is tiny
//     and the matrix pipe is the only saturated resource.
//   * A (weights) never touches LDS: the host packs it into MFMA fragment order, so each wave
//     streams its 32-row slice with one fully coalesced 1 KiB global_load_dwordx4 per 4 k-steps
//     (L2-resident: the largest layer is 2.9 MB), register double-buffered one group ahead.
//   * B (activations): a [CI channels][BN + halo] slab is staged global->reg->(leaky-relu)->LDS with
//     coalesced 256 B rows, double-buffered in LDS; one barrier per CI-channel chunk.  Within a
//     k-step the two 32-lane halves of the wave read two adjacent channels at the same tap, so every
//     ds_read_b32 is 32 consecutive dwords per half: bank-conflict free for every dilation.
//   * 256-thread workgroups (4 waves), 2 per CU; wave tile 64x64 (2x2 MFMA tiles) or 32x128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

namespace parrot {

// Samples (at the current layer) of a batch row that holds n real units: n * mul + add for n > 0 -- `add` is what the
// upsampling stages with odd kernel_size - rate contribute (ConvTranspose1d then yields T u + 1 samples, reference
// utils/vocoder/models.py:80-83) -- and 0 for an empty row.
__host__ __device__ __forceinline__ int row_true_len(int n, int mul, int add) { return n > 0 ? n * mul + add : 0; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: remember it per (kernel instantiation, device), not per
// process -- a second device in the same process would otherwise launch with the 64 KiB default and fail.
struct DynLdsOnce {
    bool done[64] = {};
};
inline hipError_t ensure_dyn_lds(DynLdsOnce& st, const void* kern, size_t lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && st.done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && tracked) st.done[dev] = true;
    return e;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PRE_NONE = 0, PRE_LRELU = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_ADD_DIV = 2 };

constexpr int CONV_HALO = 64;  // max (k-1)*dil the fast path supports (MRF needs 50)

struct ConvParams {
    const float* x;      // (B, Cin, Tin)
    const float* wfrag;  // packed A fragments [mtile][it][lane][4]
    const float* bias;   // [Cout] or null
    const float* res;    // (B, Cout, Tout) or null
    float* y;            // (B, Cout, Tout)
    int B, Cin, Tin;
    int M;               // GEMM rows: Cout (conv) or Cout*u (transposed, row = o*u + phase)
    int Cout;
    int Ncols;           // GEMM columns per batch row
    int Tout;            // output length per batch row
    int k, dil, pad_left;
    int nchunks;         // ceil(Cin / CI)
    int n_it;            // nchunks * k * (CI/8): A-fragment groups per m-tile
    int pre;
    float pre_slope;
    int act;
    int epi;
    float div;
    int u;               // transposed stride (1 for plain conv)
    int tiles_n;
    long x_bstride;      // elements between batch rows of x / y / res (lets callers address channel slices)
    long y_bstride;
    long res_bstride;
    int groups;          // grouped conv: Cin is PER GROUP, rows [g*Mg, (g+1)*Mg) read channels [g*Cin, (g+1)*Cin)
    int Mg;              // rows per group (multiple of the block's BM when groups > 1)
    int u_inv16;         // ceil(65536 / u)
    const int32_t* row_len;  // optional per-batch-row true length (in base units); input positions >= row_len[b]*row_len_mul
    int row_len_mul;         // read as zero: each row then sees its OWN sequence edge (ragged batches); null = Tin for all
    int row_len_add;         // true length of a row of n > 0 units at this layer = n * row_len_mul + row_len_add (odd k - u upsampling stages add samples)
    int fold_res;            // 1: the residual tile initialises the accumulator (prologue latency hiding); 0: added once in the
                             // epilogue -- what the reference computes (conv, THEN + res): when |res| >> |conv sum| (the TTE's residual
                             // stream) folding makes every accumulation step round at ulp(|res|) instead of ulp(|sum|)
    int lean;                // conv_split_kernel: buffer-addressed prologue / epilogue for plain convs (conv_acc_init_lean)
    int n_cus;           // CUs of the device (workgroup slots per round = n_cus x workgroups per CU)
    int epi16;           // conv_split16: the output / residual rows are 16-byte aligned (Tout % 4 == 0, aligned bases): the epilogue
                         // transposes each 16-row tile through a wave-private LDS region and moves 16 bytes per lane and instruction
    float acc_scale;         // split schemes with pre-scaled operands (conv_split.h): the accumulator holds acc_scale * sum
    float out_scale;         // = 1 / acc_scale (both exact powers of two; 1 for every other kernel)
    // Operand planes (conv_split16.h): an activation tensor as the CONSUMER's ready-made MFMA operand -- leaky ReLU, scale and the
    // scheme's 16-bit split applied once by the producer's epilogue -- laid out per batch row as [piece][C / 8][T][8 x 16 bit].
    const void* xplane;      // consumer: read the conv input from this plane instead of x (same values, no conversion in the kernel)
    void* yplane;            // producer: also write split(pre(y)) for the next layer
    long xplane_bstride;     // bytes between batch rows
    long yplane_bstride;
    float yplane_slope;      // the NEXT layer's leaky-ReLU slope (1: none)
    int plane_only;          // producer: the fp32 output is not needed (the tensor between the two convs of a ResBlock pair)
};

// ACT_TANH (conv_post only) is applied by tanh_inplace_kernel right after the conv launch: inlining tanhf
// 64x into this epilogue costs ~40 VGPRs and a wave of occupancy.
__device__ __forceinline__ float apply_act(float v, int act) { return (act == ACT_RELU && v < 0.f) ? 0.f : v; }

static __global__ void tanh_inplace_kernel(float* __restrict__ y, size_t n, int* __restrict__ err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = y[i];
        y[i] = tanhf(v);
        // NaN / inf reached the waveform (fp16 split range exceeded).  The PRE-activation is tested: tanhf(+-inf) = +-1 would pass
        if (err && !(fabsf(v) < INFINITY)) atomicExch(err, 5);
    }
}

// ---- shared by every conv kernel variant: accumulator init (bias + folded residual) and epilogue --------------
// m_wave / n_wave: first GEMM row / column of this wave's tile block; C/D layout of the 32x32 MFMA shapes:
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) (identical for the f32 and bf16 instructions).
template <int WM, int WN>
__device__ __forceinline__ void conv_acc_init(const ConvParams& p, f32x16 (&acc)[WM][WN], int b, int m_wave, int n_wave, int half,
                                              int l31) {
    // Accumulators start at bias (+ residual): the residual tile is fetched here, in the prologue, where its
    // latency hides behind the first slab load, instead of in the epilogue where every co-resident workgroup
    // would sit in a memory phase at the same time (they run in lockstep).  Folding needs act == none.
    float* __restrict__ yb = p.y + (size_t)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const bool plain = p.u == 1;
    const bool fold_res = plain && rb != nullptr && p.act == ACT_NONE && p.fold_res != 0;
    // pass 1: every residual element goes straight into its own accumulator register, so all WM*WN*16 loads are
    // in flight together (clamped addresses; masked lanes are never stored); pass 2 adds the bias.
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int mbase = m_wave + mt * 32 + 4 * half;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int n = n_wave + nt * 32 + l31;
            const bool colok = n < p.Ncols;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                acc[mt][nt][r] = fold_res ? rb[(colok && m < p.M) ? m * p.Tout + n : 0] : 0.f;
            }
        }
    }
    if (plain && p.bias) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int mbase = m_wave + mt * 32 + 4 * half;
            float bsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bsv[r] = p.bias[min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1)];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bsv[r];
        }
    }
    if (p.acc_scale != 1.f) {  // (uniform branch; a power of two: exact)
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= p.acc_scale;
    }
}

template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[WM][WN], int b, int m_wave, int n_wave, int half,
                                              int l31) {
    float* __restrict__ yb = p.y + (size_t)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const bool plain = p.u == 1;
    const bool fold_res = plain && rb != nullptr && p.act == ACT_NONE && p.fold_res != 0;
    if (p.out_scale != 1.f) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= p.out_scale;
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    if ((p.u == 4 || p.u == 2) && p.Tout == p.Ncols * p.u) {
        // transposed conv with stride 4 / 2: the four consecutive rows a lane holds per register group (C/D layout) are
        // the 4 phases of ONE output channel (or 2 phases of two), i.e. consecutive output samples: 16-/8-byte stores
        // instead of 16 strided 4-byte ones, one bias load per group.
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int mbase = m_wave + mt * 32 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int n = n_wave + nt * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mbase + 8 * q;
                    const bool ok = m < p.M && n < p.Ncols;
                    if (p.u == 4) {
                        const int o = m >> 2;
                        const float bs = (p.bias && ok) ? p.bias[o] : 0.f;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[mt][nt][4 * q + e] + bs, p.act);
                        if (ok) *reinterpret_cast<f32x4*>(yb + (size_t)o * p.Tout + 4 * n) = v;
                    } else {
                        const int o = m >> 1;
                        const float b0 = (p.bias && ok) ? p.bias[o] : 0.f, b1 = (p.bias && ok && m + 2 < p.M) ? p.bias[o + 1] : 0.f;
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        f32x2_t v0, v1;
                        v0[0] = apply_act(acc[mt][nt][4 * q + 0] + b0, p.act);
                        v0[1] = apply_act(acc[mt][nt][4 * q + 1] + b0, p.act);
                        v1[0] = apply_act(acc[mt][nt][4 * q + 2] + b1, p.act);
                        v1[1] = apply_act(acc[mt][nt][4 * q + 3] + b1, p.act);
                        if (ok) *reinterpret_cast<f32x2_t*>(yb + (size_t)o * p.Tout + 2 * n) = v0;
                        if (ok && m + 2 < p.M) *reinterpret_cast<f32x2_t*>(yb + (size_t)(o + 1) * p.Tout + 2 * n) = v1;
                    }
                }
            }
        }
        return;
    }
    if (p.u > 1) {
        // transposed conv: GEMM row m = o*u + phase, column n = input step -> out[o][n*u + phase]; stores only.
        // o = m / u via a 16-bit reciprocal of the small in-tile remainder (exact for rem < 64, u <= 64).
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int mbase = m_wave + mt * 32 + 4 * half;
            const int o0 = mbase / p.u, rem0 = mbase - o0 * p.u;
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int n = n_wave + nt * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2);
                    const int rem = rem0 + cr;
                    const int dq = (rem * p.u_inv16) >> 16;
                    const int o = o0 + dq;
                    const int tau = n * p.u + (rem - dq * p.u);
                    const bool ok = (mbase + cr < p.M) && (n < p.Ncols) && (tau < p.Tout);
                    const float v = acc[mt][nt][r] + ((p.bias && ok) ? p.bias[o] : 0.f);
                    if (ok) yb[o * p.Tout + tau] = apply_act(v, p.act);
                }
            }
        }
        return;
    }
    // plain conv (bias and, when folded, the residual are already inside acc): per half tile (8 rows) issue the
    // remaining loads together (masked lanes read element 0, always mapped), finish branch-free, store.
    const bool late_res = rb != nullptr && !fold_res;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    const float act_lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int mbase = m_wave + mt * 32 + 4 * half;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int n = n_wave + nt * 32 + l31;
            const bool colok = n < p.Ncols;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                int off[8];
                float rv[8], yv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = g * 8 + q;
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    off[q] = (colok && m < p.M) ? m * p.Tout + n : -1;
                    rv[q] = 0.f;
                    yv[q] = 0.f;
                }
                if (late_res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = rb[off[q] < 0 ? 0 : off[q]];
                }
                if (has_acc) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) yv[q] = yb[off[q] < 0 ? 0 : off[q]];
                }
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = yv[q] + (fmaxf(acc[mt][nt][g * 8 + q], act_lo) + rv[q]);
                if (do_div) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = v[q] / p.div;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (off[q] >= 0) yb[off[q]] = v[q];
            }
        }
    }
}

// ---- lean accumulator init / epilogue for PLAIN convs (u == 1) -----------------------------------------------------------
// The generic versions above address every element with 64-bit pointer arithmetic and per-element bounds selects: ~25 VALU
// instructions per accumulator register in the prologue and the epilogue together -- more VALU time than the main loop's MFMAs
// on the short-K layers (128 channels: 4 chunks), and VALU beside another wave's MFMAs costs 2-3x its stand-alone rate
// (tools/probes/valu_rates.hip).  Here a row-major (M, Tout) fp32 tile is accessed through buffer instructions: the
// lane-dependent part of an element's byte offset is ONE VGPR per column tile (0x80000000 when the column is out of range:
// loads return 0, stores are dropped), the row part is a scalar offset, rows >= M fall beyond num_records = M * Tout * 4.
// Same arithmetic in the same order as the generic code (results equal; an exact zero may come out as +0 instead of -0).
struct RowTile {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ RowTile row_tile(const float* base, int rows, int row_elems) {
    const size_t a = reinterpret_cast<size_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    RowTile t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(rows * row_elems * 4), 0x00020000);
    return t;
}
__device__ __forceinline__ float row_tile_load(const RowTile& t, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rsrc, voff, soff, 0));
}
__device__ __forceinline__ void row_tile_store(const RowTile& t, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), t.rsrc, voff, soff, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 row_tile_load4(const RowTile& t, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t.rsrc, voff, soff, 0));
}
__device__ __forceinline__ void row_tile_store4(const RowTile& t, const f32x4& v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), t.rsrc, voff, soff, 0);
}
// plain convs whose (M, Tout) tile of one batch row fits 31-bit byte offsets (every layer of the path) take the lean code
__host__ __device__ __forceinline__ bool conv_lean_ok(const ConvParams& p) { return p.lean != 0 && p.u == 1 && (long)p.M * p.Tout * 4 < 0x7fffffffL; }

// 32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
template <int WM, int WN>
__device__ __forceinline__ void conv_acc_init_lean(const ConvParams& p, f32x16 (&acc)[WM][WN], int b, int m_wave, int n_wave, int half, int l31) {
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const bool fold_res = rb != nullptr && p.act == ACT_NONE && p.fold_res != 0;
    const int row_bytes = p.Tout * 4;
    int vo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int n = n_wave + nt * 32 + l31;
        vo[nt] = (n < p.Ncols) ? (4 * half * p.Tout + n) * 4 : (int)0x80000000;
    }
    const RowTile rt = row_tile(fold_res ? rb : p.x, p.M, p.Tout);  // (p.x: never read when !fold_res)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m0 = m_wave + mt * 32;  // wave-uniform
        if (fold_res) {  // (the residual loads go out first: the longest latency of the prologue)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = row_tile_load(rt, vo[nt], (m0 + (r & 3) + 8 * (r >> 2)) * row_bytes);
        }
        float bs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bs[r] = p.bias ? p.bias[min(m0 + 4 * half + (r & 3) + 8 * (r >> 2), p.M - 1)] * p.acc_scale : 0.f;
        if (fold_res) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], p.acc_scale, bs[r]);  // = (res + bias) * scale, one rounding
        } else {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bs[r];
        }
    }
}

template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue_lean(const ConvParams& p, f32x16 (&acc)[WM][WN], int b, int m_wave, int n_wave, int half, int l31) {
    float* __restrict__ yb = p.y + (size_t)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bstride : nullptr;
    const bool fold_res = rb != nullptr && p.act == ACT_NONE && p.fold_res != 0;
    const bool late_res = rb != nullptr && !fold_res;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    const bool relu = p.act == ACT_RELU;
    const int row_bytes = p.Tout * 4;
    int vo[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int n = n_wave + nt * 32 + l31;
        vo[nt] = (n < p.Ncols) ? (4 * half * p.Tout + n) * 4 : (int)0x80000000;
    }
    const RowTile yt = row_tile(yb, p.M, p.Tout), rt = row_tile(late_res ? rb : yb, p.M, p.Tout);
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m0 = m_wave + mt * 32;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[mt][nt][r] * p.out_scale;
            if (relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (late_res) {
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = row_tile_load(rt, vo[nt], (m0 + (r & 3) + 8 * (r >> 2)) * row_bytes);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] + rv[r];
            }
            if (has_acc) {
                float yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = row_tile_load(yt, vo[nt], (m0 + (r & 3) + 8 * (r >> 2)) * row_bytes);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = yv[r] + v[r];
                if (do_div) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] / p.div;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) row_tile_store(yt, v[r], vo[nt], (m0 + (r & 3) + 8 * (r >> 2)) * row_bytes);
        }
    }
}

// WAVES_M x WAVES_N waves, each owning WM x WN MFMA tiles of 32x32; CI input channels per LDS slab;
// MINW = workgroups per CU the register allocation must leave room for (LDS caps the 32x512 tile at 2).
template <int WAVES_M, int WAVES_N, int WM, int WN, int CI, int MINW>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_mfma_kernel(const ConvParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int RS = BN + CONV_HALO;      // LDS row stride (floats)
    constexpr int QN = CI / 8;              // A groups (4 k-steps x 2 channels) per tap per chunk
    constexpr int ROWS_PW = CI / NW;        // slab rows staged by each wave
    constexpr int COLS_IT = (RS + 63) / 64; // 64-column strips per row
    static_assert(CI % 8 == 0 && CI % NW == 0, "CI must be a multiple of 8 and of the wave count");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][CI][RS]

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
    const int W = BN + (p.k - 1) * p.dil;  // live slab width (<= RS, checked on the host)
    const int Tlim = p.row_len ? min(p.Tin, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.Tin;  // this row's true input length
    const int grp = (p.groups > 1) ? (blockIdx.y * BM) / p.Mg : 0;
    const float* __restrict__ xb = p.x + (size_t)b * p.x_bstride + (size_t)grp * p.Cin * p.Tin;

    float stage[ROWS_PW][COLS_IT];

    // Branch-free slab load: the address is clamped into the row (always mapped), masked values become 0.
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
        for (int r = 0; r < ROWS_PW; ++r) {
#pragma unroll
            for (int i = 0; i < COLS_IT; ++i) {
                const int col = lane + 64 * i;
                if (col < RS) dst[(wave * ROWS_PW + r) * RS + col] = stage[r][i];
            }
        }
    };

    f32x16 acc[WM][WN];
    const int m_wave = blockIdx.y * BM + wm * WM * 32;
    const int n_wave = t0 + wn * WN * 32;
    conv_acc_init<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);

    // A-fragment stream: one 1 KiB group (64 lanes x float4) per (chunk, tap, q) per m-tile.
    const f32x4* __restrict__ aptr[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int mtg = blockIdx.y * (BM / 32) + wm * WM + mt;
        aptr[mt] = reinterpret_cast<const f32x4*>(p.wfrag) + (size_t)mtg * p.n_it * 64 + lane;
    }
    f32x4 a_nxt[WM], a_cur[WM];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) a_nxt[mt] = aptr[mt][0];

    load_slab(0);
    store_slab(0);
    __syncthreads();

    // per-lane LDS read base: row (half) of the channel pair, column of this lane in the wave's N range
    const int bbase = half * RS + wn * (WN * 32) + l31;
    int it = 0;
    float bv_cur[4][WN], bv_nxt[4][WN];
    for (int c = 0; c < p.nchunks; ++c) {
        const float* __restrict__ xs = smem + (c & 1) * (CI * RS) + bbase;
        const bool more = (c + 1 < p.nchunks);
        if (more) load_slab(c + 1);
        // operands of the chunk's first iteration (one exposed LDS latency per chunk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) bv_cur[e][nt] = xs[(2 * e) * RS + nt * 32];
        for (int j = 0; j < p.k; ++j) {
            const float* __restrict__ xj = xs + j * p.dil;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                ++it;
                // software pipeline: fetch the NEXT iteration's A group (L2) and B values (LDS) before this
                // iteration's 16*WM*WN/4 MFMAs; the sched_barrier keeps hipcc from sinking them to their use.
                // (after the chunk's last iteration the B prefetch reads in-row garbage that is discarded)
                const float* __restrict__ xn = (q + 1 < QN) ? xj + 8 * (q + 1) * RS : xj + p.dil;
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) {
                    a_cur[mt] = a_nxt[mt];
                    a_nxt[mt] = aptr[mt][(size_t)it * 64];  // wfrag is padded by one group at the end
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt) bv_nxt[e][nt] = xn[(2 * e) * RS + nt * 32];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][e], bv_cur[e][nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < WN; ++nt) bv_cur[e][nt] = bv_nxt[e][nt];
            }
        }
        if (more) store_slab((c + 1) & 1);
        __syncthreads();
    }

    conv_epilogue<WM, WN>(p, acc, b, m_wave, n_wave, half, l31);
}

// tile table (index = parrot_conv_desc.tile_cfg)
struct TileCfg {
    int bm, bn, ci, threads;
};
constexpr int NUM_TILE_CFGS = 7;
__host__ inline TileCfg tile_cfg(int id) {
    switch (id) {
        case 0: return {128, 128, 16, 256};  // waves 2x2, wave 64x64
        case 1: return {64, 256, 16, 256};   // waves 1x4, wave 64x64
        case 2: return {32, 512, 16, 256};   // waves 1x4, wave 32x128
        case 3: return {128, 128, 32, 256};  // as 0 with 32-channel slabs (fewer barriers for k<=3)
        case 6: return {16, 512, 16, 256};   // conv_mfma16.h: 16-row MFMA (v_mfma_f32_16x16x4_f32) for <= 16 output channels
        case 5: return {32, 256, 16, 256};   // waves 1x4, wave 32x64: small footprint -> 4 workgroups per CU (HBM-bound layers)
        default: return {128, 64, 16, 256};  // waves 2x2, wave 64x32 (short sequences)
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int CI, int MINW>
inline hipError_t launch_conv_t(const ConvParams& p, dim3 grid, size_t lds, hipStream_t s) {
    auto kern = conv_mfma_kernel<WAVES_M, WAVES_N, WM, WN, CI, MINW>;
    static DynLdsOnce lds_once;  // (> 64 KiB of dynamic LDS needs an explicit opt-in, per device)
    {
        hipError_t e = ensure_dyn_lds(lds_once, reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    return hipGetLastError();
}

// (a template so that the six kernels are instantiated only in the translation unit that calls it, not in every one that includes this header)
template <class = void>
inline hipError_t launch_conv(int cfg, const ConvParams& p, hipStream_t s) {
    const TileCfg t = tile_cfg(cfg);
    dim3 grid(p.tiles_n * p.B, (p.M + t.bm - 1) / t.bm);
    const size_t lds = (size_t)2 * t.ci * (t.bn + CONV_HALO) * sizeof(float);
    switch (cfg) {
        case 0: return launch_conv_t<2, 2, 2, 2, 16, 3>(p, grid, lds, s);
        case 1: return launch_conv_t<1, 4, 2, 2, 16, 3>(p, grid, lds, s);
        case 2: return launch_conv_t<1, 4, 1, 4, 16, 2>(p, grid, lds, s);
        case 3: return launch_conv_t<2, 2, 2, 2, 32, 3>(p, grid, lds, s);
        case 5: return launch_conv_t<1, 4, 1, 2, 16, 4>(p, grid, lds, s);
        default: return launch_conv_t<2, 2, 2, 1, 16, 3>(p, grid, lds, s);
    }
}

}  // namespace parrot
