// kernels_misc.h -- the HBM-bound glue kernels of the Parrot-TTS path (gathers, LayerNorm, softmax,
// duration rounding, length regulator, argmax).  All activations are channel-first (B, C, T).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace parrot {

// ---------------------------------------------------------------------------------------------
// vocoder input: x[b, c, t] = c < E ? dict[code[b,t]][c] : spkr[spkr_id[b]][c - E]
// (utils/vocoder/models.py:155-160 + _upsample :132-151: the speaker vector is repeated over time)
// grid (ceil(U/64), C/?, B): each block = 64 time steps x 64 channels via an LDS transpose so that
// both the embedding-row reads (contiguous in c) and the (B,C,U) writes (contiguous in t) coalesce.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void voc_embed_kernel(const int64_t* __restrict__ code, const int64_t* __restrict__ spkr,
                                                        const float* __restrict__ dict, const float* __restrict__ spk_tab,
                                                        float* __restrict__ x, int U, int E, int C, int Cx, int n_emb, int n_spk,
                                                        int* __restrict__ err, int code_stride) {  // C embedding channels of the Cx input channels; code rows code_stride apart
    __shared__ float tile[64][65];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows of 64
    // load: rows = time, cols = channel (contiguous reads along the embedding row)
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, c = c0 + tx;
        float v = 0.f;
        if (t < U && c < C) {
            if (c < E) {
                int64_t id = code[(size_t)b * code_stride + t];
                if (id < 0 || id >= n_emb) { atomicExch(err, 1); id = 0; }
                v = dict[(size_t)id * E + c];
            } else {
                int64_t s = spkr[b];
                if (s < 0 || s >= n_spk) { atomicExch(err, 2); s = 0; }
                v = spk_tab[(size_t)s * E + (c - E)];
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < U) x[((size_t)b * Cx + c) * U + t] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------
// TTE input: x[b, c, s] = tok_emb[phones[b,s]][c] + pe[S][c]   (parrot.py:94-95, fft.py:17-19, Q1)
// row_len (nullable): row-exact mode, pe[row_len[b]] -- the single row the reference adds when it runs that utterance alone.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void tte_embed_kernel(const int64_t* __restrict__ phones, const float* __restrict__ emb,
                                                        const float* __restrict__ pe, const int32_t* __restrict__ row_len,
                                                        float* __restrict__ x, int S, int D, int vocab, int* __restrict__ err) {
    __shared__ float tile[64][65];
    const int s0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const float* __restrict__ pe_row = pe + (size_t)(row_len ? min(max(row_len[b], 0), S) : S) * D;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + r, c = c0 + tx;
        float v = 0.f;
        if (s < S && c < D) {
            int64_t id = phones[(size_t)b * S + s];
            if (id < 0 || id >= vocab) { atomicExch(err, 3); id = 0; }
            v = pe_row[c] + emb[(size_t)id * D + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, s = s0 + tx;
        if (c < D && s < S) x[((size_t)b * D + c) * S + s] = tile[tx][r];
    }
}

// x[b, c, t] += tab[id[b]][c]      (speaker embedding add, parrot.py:98-99, Q6: all positions incl. pads)
static __global__ void add_channel_vec_kernel(float* __restrict__ x, const int64_t* __restrict__ id, const float* __restrict__ tab,
                                       int C, int T, int n_rows, size_t total, int* __restrict__ err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((size_t)T * C));
    int64_t s = id[b];
    if (s < 0 || s >= n_rows) { atomicExch(err, 4); s = 0; }
    x[i] = x[i] + tab[(size_t)s * C + c];
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels of a channel-first tensor (fft.py:91-92,98-99; duration.py:32,36):
// biased variance, eps inside the sqrt, affine.  Optional ReLU on the input (duration predictor:
// Conv -> ReLU -> LayerNorm).  Two passes (mean, then centred variance), cross-wave combine through LDS.
// ---------------------------------------------------------------------------------------------
// NW waves per block split the channels (c = wave, wave + NW, ...); lane = time step, so every load / store is a
// coalesced 256-byte row segment.  For C <= NW * 16 a thread's channel values stay in registers (one global read);
// wider layers re-read them per pass.  (NW = 16: 16 waves per (batch row, 64 steps) keep a CU busy where 4 left it
// latency-bound.)
template <int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_cf_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y, int C, int T,
                                                                float eps, int relu_in) {
    __shared__ float red[NW][64];
    constexpr int MAXV = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool ok = t < T;
    const float* xb = x + (size_t)b * C * T + (ok ? t : 0);
    const bool in_regs = C <= NW * MAXV;
    float vals[MAXV];
    float s = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = wave + NW * i;
            float v = (c < C) ? xb[(size_t)c * T] : 0.f;
            if (relu_in) v = v > 0.f ? v : 0.f;
            vals[i] = v;
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += (wave + NW * i < C) ? vals[i] : 0.f;
    } else {
        for (int c = wave; c < C; c += NW) {
            float v = xb[(size_t)c * T];
            if (relu_in) v = v > 0.f ? v : 0.f;
            s += v;
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w][lane];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const float d = vals[i] - mean;
            q += (wave + NW * i < C) ? d * d : 0.f;
        }
    } else {
        for (int c = wave; c < C; c += NW) {
            float v = xb[(size_t)c * T];
            if (relu_in) v = v > 0.f ? v : 0.f;
            const float d = v - mean;
            q += d * d;
        }
    }
    red[wave][lane] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qt += red[w][lane];
    const float var = qt / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (ok) {
        float* yb = y + (size_t)b * C * T + t;
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = wave + NW * i;
                if (c < C) yb[(size_t)c * T] = (vals[i] - mean) * rstd * gamma[c] + beta[c];
            }
        } else {
            for (int c = wave; c < C; c += NW) {
                float v = xb[(size_t)c * T];
                if (relu_in) v = v > 0.f ? v : 0.f;
                yb[(size_t)c * T] = (v - mean) * rstd * gamma[c] + beta[c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax of attention scores with a key-padding mask (torch MHA math path, Q3):
// P[r, :] = softmax(S[r, :] + (-inf where key masked)).  One wave per row, wave-shuffle reductions.
// rows = B*H*T, row r belongs to batch r / (H*T).  key_valid (B,T) u8 1 = attend.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static __global__ __launch_bounds__(256) void softmax_mask_kernel(float* __restrict__ s, const uint8_t* __restrict__ key_valid, int rows,
                                                           int T, int HT) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / HT;
    float* sr = s + (size_t)row * T;
    const uint8_t* kv = key_valid + (size_t)b * T;
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) {
        const float v = kv[t] ? sr[t] : -INFINITY;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float e = kv[t] ? expf(sr[t] - mx) : 0.f;
        sr[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int t = lane; t < T; t += 64) sr[t] = sr[t] / sum;
}

// ---------------------------------------------------------------------------------------------
// Durations (parrot.py:82-86, duration.py:46-47): ld = masked_fill(ld, pad, 0);
// dur = max(rint(exp(ld) - 1), 0) as int64 (torch.round = round-half-even = rintf);
// one block per batch row also produces out_len[b] = sum(dur) and the exclusive prefix sums.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void duration_kernel(const float* __restrict__ ld_raw, const uint8_t* __restrict__ src_valid,
                                                       float* __restrict__ log_dur, int64_t* __restrict__ dur,
                                                       int32_t* __restrict__ cum /* (B,S) inclusive */, int32_t* __restrict__ out_len,
                                                       int S, const int32_t* __restrict__ src_len = nullptr) {
    __shared__ int32_t part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (S + 255) / 256;
    const int s_begin = tid * per, s_end = min(S, s_begin + per);
    int32_t local = 0;
    for (int s = s_begin; s < s_end; ++s) {
        // (row-exact: the row's own token count decides, not the key mask -- a row of NO tokens keeps key 0 attendable so that its
        //  softmax is defined, and must still expand to nothing)
        const bool real = src_len ? s < src_len[b] : src_valid[(size_t)b * S + s] != 0;
        float v = real ? ld_raw[(size_t)b * S + s] : 0.0f;
        log_dur[(size_t)b * S + s] = v;
        float d = rintf(expf(v) - 1.0f);
        d = d > 0.f ? d : 0.f;
        const int64_t di = (int64_t)d;
        dur[(size_t)b * S + s] = di;
        local += (int32_t)di;
    }
    part[tid] = local;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const int32_t v = part[i];
            part[i] = run;
            run += v;
        }
        out_len[b] = run;
    }
    __syncthreads();
    int32_t run = part[tid];
    for (int s = s_begin; s < s_end; ++s) {
        run += (int32_t)dur[(size_t)b * S + s];
        cum[(size_t)b * S + s] = run;
    }
}

// per-row unit counts re-based to a chunk [lo, lo + n): clamp(len - lo, 0, n)   (parrot_voc_forward_chunked)
static __global__ void rebase_lens_kernel(const int32_t* __restrict__ lens, int32_t* __restrict__ out, int B, int lo, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = min(max(lens[b] - lo, 0), n);
}

// inclusive prefix sums + totals of given durations (parrot_length_regulator: the standalone entry point)
static __global__ __launch_bounds__(256) void dur_prefix_kernel(const int64_t* __restrict__ dur, int32_t* __restrict__ cum, int32_t* __restrict__ out_len, int S) {
    __shared__ int32_t part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (S + 255) / 256;
    const int s_begin = min(S, tid * per), s_end = min(S, s_begin + per);
    int32_t local = 0;
    for (int s = s_begin; s < s_end; ++s) local += (int32_t)max(dur[(size_t)b * S + s], (int64_t)0);
    part[tid] = local;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const int32_t v = part[i];
            part[i] = run;
            run += v;
        }
        out_len[b] = run;
    }
    __syncthreads();
    int32_t run = part[tid];
    for (int s = s_begin; s < s_end; ++s) {
        run += (int32_t)max(dur[(size_t)b * S + s], (int64_t)0);
        cum[(size_t)b * S + s] = run;
    }
}

constexpr int TIE_GUARD_MAX = 256;  // guarded positions per batch (argmax_cf_kernel / tie_guard_refine_kernel below)

// ---------------------------------------------------------------------------------------------
// Length regulator + positional row (duration.py:6-24, parrot.py:106, data.py:8-20):
// y[b, c, t] = (t < len_b ? enc[b, c, src(t)] : 0) + pe[L][c], src(t) = first s with cum[b,s] > t;
// tgt_mask[b,t] = t <= len_b (Q2).   grid (ceil(L/64), B).  idx is recomputed per (b,t) once and
// reused over channels.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void length_regulate_kernel(const float* __restrict__ enc, const int32_t* __restrict__ cum,
                                                              const int32_t* __restrict__ out_len, const float* __restrict__ pe,
                                                              float* __restrict__ y, uint8_t* __restrict__ tgt_mask, int S, int L, int D,
                                                              int* __restrict__ gstat = nullptr, int gstat_reset = 1, int row_exact = 0, int pe_stride = -1) {
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6;  // lane = frame, the four waves split the channels
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    // (first kernel of a decode: restart the tie-guard statistics {count, ids changed, min margin bits = +inf, first list entry of
    //  this row group} of argmax_cf_kernel; a later row group of the same batch keeps them and notes where its entries start)
    if (gstat && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) {
        if (gstat_reset) gstat[threadIdx.x] = threadIdx.x == 2 ? 0x7f800000 : 0;
        else if (threadIdx.x == 3) gstat[3] = min(gstat[0], TIE_GUARD_MAX);
    }
    if (t >= L) return;
    const int len = out_len[b];
    // padded batch: pe[L] of the batch-max length and the `<=` mask (one extra frame per shorter row, quirk Q2); row-exact: what the
    // row's own B = 1 run sees -- pe[len] and exactly len frames (get_mask_from_lengths(len, max_len = len) is all True)
    const float* __restrict__ pe_row = pe + (size_t)(row_exact ? min(len, L) : L) * (pe_stride < 0 ? D : pe_stride);  // (pe_stride 0: one row for every length)
    // (row-exact rows of length 0 keep key 0 valid: a softmax over no keys at all is 0 / 0, and its NaN would raise the batch's
    //  non-finite flag for a row that emits nothing -- the host recomputes the returned mask from the lengths)
    if (wave == 0) tgt_mask[(size_t)b * L + t] = (row_exact ? t < max(len, 1) : t <= len) ? 1 : 0;
    int src = -1;
    if (t < len) {
        const int32_t* cb = cum + (size_t)b * S;
        int lo = 0, hi = S - 1;  // smallest s with cum[s] > t
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cb[mid] > t) hi = mid; else lo = mid + 1;
        }
        src = lo;
    }
    const float* eb = enc + (size_t)b * D * S;
    float* yb = y + (size_t)b * D * L + t;
    for (int c = wave; c < D; c += 4) {
        const float v = (src >= 0) ? eb[(size_t)c * S + src] : 0.0f;
        yb[(size_t)c * L] = pe_row[c] + v;
    }
}

// argmax over channels of (B, V, L) logits -> ids (B, L); first maximal index wins (torch.argmax).
// Tie guard (reference modules/parrot.py:115 takes the argmax of 1000 fp32 logits): every position also yields its top-2
// margin; positions whose margin is below `guard` are appended to `glist` ((b, t) pairs, at most TIE_GUARD_MAX) and counted in
// gstat[0], the smallest margin of the call lands in gstat[2] (float bits; positive floats order like ints) --
// tie_guard_refine_kernel then re-evaluates the head for exactly those positions in fp64.
constexpr int ARGMAX_WAVES = 16;  // waves per 64 positions: 62-63 codes each at V = 1000 (sixteen loads in flight per position)
static __global__ __launch_bounds__(64 * ARGMAX_WAVES) void argmax_cf_kernel(const float* __restrict__ logits, int64_t* __restrict__ ids, int V, int L,
                                                               int* __restrict__ err, float guard, int* __restrict__ glist,
                                                               int* __restrict__ gstat, int row0 = 0) {
    // lane = time step; the waves scan a slice of the vocabulary each, then the first maximum wins
    // (strict > inside a range, lower range first on ties: torch.argmax's first-occurrence rule)
    __shared__ float bv[ARGMAX_WAVES][64], sv[ARGMAX_WAVES][64];
    __shared__ int bix[ARGMAX_WAVES][64], bad[ARGMAX_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < L;
    const int per = (V + ARGMAX_WAVES - 1) / ARGMAX_WAVES, v0 = wave * per, v1 = min(V, v0 + per);
    const float* lb = logits + (size_t)b * V * L + (ok ? t : 0);
    float best = -INFINITY, second = -INFINITY;  // second: the largest value of the range that is not `best`'s element
    int bi = v0 < V ? v0 : 0, nonfinite = 0;
    auto take = [&](float x, int v) {
        nonfinite |= !(fabsf(x) < INFINITY);  // NaN / inf anywhere in the column (a NaN never wins a `>` comparison)
        if (x > best) { second = best; best = x; bi = v; }
        else if (x > second) second = x;
    };
    if (v0 < V) {
        best = lb[(size_t)v0 * L];
        nonfinite |= !(fabsf(best) < INFINITY);
    }
    int v = v0 + 1;
    for (; v + 3 < v1; v += 4) {
        const float x0 = lb[(size_t)v * L], x1 = lb[(size_t)(v + 1) * L], x2 = lb[(size_t)(v + 2) * L], x3 = lb[(size_t)(v + 3) * L];
        take(x0, v); take(x1, v + 1); take(x2, v + 2); take(x3, v + 3);
    }
    for (; v < v1; ++v) take(lb[(size_t)v * L], v);
    bv[wave][lane] = best;
    sv[wave][lane] = second;
    bix[wave][lane] = bi;
    bad[wave][lane] = nonfinite;
    __syncthreads();
    if (wave == 0 && ok) {
        float m = bv[0][lane], s2 = sv[0][lane];
        int mi = bix[0][lane], nf = bad[0][lane];
#pragma unroll
        for (int w = 1; w < ARGMAX_WAVES; ++w) {
            const float bw = bv[w][lane];
            nf |= bad[w][lane];
            if (bw > m) { s2 = fmaxf(fmaxf(s2, m), sv[w][lane]); m = bw; mi = bix[w][lane]; }
            else s2 = fmaxf(s2, fmaxf(bw, sv[w][lane]));
        }
        ids[(size_t)b * L + t] = mi;
        if (err && nf) atomicExch(err, 5);  // NaN / inf logits (an activation left the fp16 split range)
        if (gstat) {
            const float margin = m - s2;  // >= 0, or NaN when the column holds non-finite logits (flagged above)
            if (margin >= 0.f) atomicMin(gstat + 2, __float_as_int(margin));
            if (!(margin >= guard)) {  // (a NaN margin is guarded too: a non-finite logit never silently skips the re-evaluation)
                const int slot = atomicAdd(gstat, 1);
                if (slot < TIE_GUARD_MAX) { glist[2 * slot] = row0 + b; glist[2 * slot + 1] = t; }  // (batch row; b is the row inside this group)
            }
        }
    }
}

// One workgroup per guarded position: logits[v] = head_b[v] + sum_c head_w[v][c] * x[b][c][t] in fp64 (exact products, fp64
// sums in channel order: the argmax no longer depends on the accumulation order of the fp32 head), first maximum wins.
// `hwt` is the head weight TRANSPOSED to (D, V): thread v reads hwt[c][v], coalesced across the workgroup.
// Deep form (f != NULL; round 4): the position's activation is first re-evaluated one layer earlier, from the last decoder block's
// own fp32 intermediates -- x[c] = h[c] + b2[c] + sum_j w2t[j][c] * f[j] in fp64 (f = relu(conv1) (B, F, L), h = x + attn (B, D, L),
// w2t the 1x1 conv2's weight as (F, D): reference modules/fft.py:81,99) -- which removes the fp32 accumulation error of that
// K = F sum and of the residual add from the refined logits.  `gref` (TIE_GUARD_MAX x V) receives the refined logits.
static __global__ __launch_bounds__(256) void tie_guard_refine_kernel(const float* __restrict__ x, const float* __restrict__ hwt,
                                                                      const float* __restrict__ hb, int64_t* __restrict__ ids, int D, int V,
                                                                      int L, const int* __restrict__ glist, int* __restrict__ gstat,
                                                                      const float* __restrict__ f, const float* __restrict__ h,
                                                                      const float* __restrict__ w2t, const float* __restrict__ b2, int F,
                                                                      float* __restrict__ gref, int row0 = 0) {
    // this row group's entries are [gstat[3], gstat[0]) of the lane's list; x / f / h / ids are the group's own buffers (row b - row0)
    const int n = min(gstat[0], TIE_GUARD_MAX), entry = gstat[3] + (int)blockIdx.x;
    if (entry >= n) return;
    extern __shared__ double xs[];  // D activations of the position (+ F intermediates in the deep form)
    __shared__ double rv[256];
    __shared__ int ri[256];
    const int b = glist[2 * entry] - row0, t = glist[2 * entry + 1];
    if (f) {
        double* fs = xs + D;
        for (int j = threadIdx.x; j < F; j += 256) fs[j] = (double)f[((size_t)b * F + j) * L + t];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            // (a thread walks F = 1024 weights with one dependent fp64 chain: what bounds it is the number of loads in flight, not
            //  the arithmetic -- 8 per thread took 96 us for a full list, one position per workgroup; 32 at a time keep the sum's order)
            double a = 0.0;
#pragma unroll 32
            for (int j = 0; j < F; ++j) a = fma((double)w2t[(size_t)j * D + c], fs[j], a);
            xs[c] = (a + (b2 ? (double)b2[c] : 0.0)) + (double)h[((size_t)b * D + c) * L + t];  // (conv + bias) + residual, fft.py:99
        }
    } else {
        for (int c = threadIdx.x; c < D; c += 256) xs[c] = (double)x[((size_t)b * D + c) * L + t];
    }
    __syncthreads();
    double best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v0 = 0; v0 < V; v0 += 1024) {  // four codes per thread and pass: v0 + threadIdx.x + {0, 256, 512, 768}
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int vv[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vv[q] = v0 + threadIdx.x + 256 * q;
            ok[q] = vv[q] < V;
            vv[q] = min(vv[q], V - 1);  // (unconditional loads: a predicated load per element would serialise them; the surplus is ignored below)
        }
#pragma unroll 8
        for (int c = 0; c < D; ++c) {
            const float* w = hwt + (size_t)c * V;
            const double xc = xs[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fma((double)w[vv[q]], xc, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double a = acc[q] + (hb ? (double)hb[vv[q]] : 0.0);
            if (ok[q] && gref) gref[(size_t)entry * V + vv[q]] = (float)a;
            if (ok[q] && a > best) { best = a; bi = vv[q]; }  // (codes ascending per thread: the first maximum is kept)
        }
    }
    rv[threadIdx.x] = best;
    ri[threadIdx.x] = bi;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) {
            const double o = rv[threadIdx.x + sft];
            const int oi = ri[threadIdx.x + sft];
            if (o > rv[threadIdx.x] || (o == rv[threadIdx.x] && oi < ri[threadIdx.x])) { rv[threadIdx.x] = o; ri[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (ids[(size_t)b * L + t] != ri[0]) atomicAdd(gstat + 1, 1);  // positions whose id the refinement changed
        ids[(size_t)b * L + t] = ri[0];
    }
}

// (B, C, T) -> (B, T, C) transpose (tests / optional logits export)
static __global__ __launch_bounds__(256) void transpose_cf_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T) {
    __shared__ float tile[64][65];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < C && t < T) ? x[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, c = c0 + tx;
        if (t < T && c < C) y[((size_t)b * T + t) * C + c] = tile[tx][r];
    }
}

// wav fp32 -> int16 exactly like numpy's `(x * 32768).astype('int16')` for in-range values
// (C cast: truncation toward zero; utils/vocoder/inference.py:71-73).
static __global__ void wav_to_int16_kernel(const float* __restrict__ w, int16_t* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (int16_t)(int32_t)(w[i] * 32768.0f);
}

// max |x[i]| -> atomic max into dst[0] (non-negative floats order like their bit patterns); NaN / inf count as +inf
static __global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ dst) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = fabsf(x[i]);
        m = (v > m || !(v == v)) ? (v == v ? v : INFINITY) : m;
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(m));
}

static __global__ void copy_kernel(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

}  // namespace parrot
