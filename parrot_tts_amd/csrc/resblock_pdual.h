// resblock_pdual.h -- the dual-window anti-phase fused ResBlock1 pair kernel of resblock_dual.h as a PERSISTENT workgroup.
//
// resblock_dual.h showed (profiles/r03c_rbd_phase_trace.log) that two windows in anti-phase on the same SIMDs run a pair's
// phases faster than three lock-stepped workgroups (32 channels, per window and conv: k = 3 3800 vs 4670 clocks, k = 11
// 8300 vs 11100) -- and lose it all again to the prologue and epilogue of a workgroup that is alone on its CU (loading the
// window, first conversion, final store: 12-17 k of 54-78 k clocks with nothing to overlap them).  Here ONE workgroup per CU
// walks a list of window pairs:
//   * the next window's input is fetched into the running-residual registers R at the end of the current window's last conv
//     (R is dead from that conv's accumulator init on: the residual rides in the accumulator);
//   * the current window's result is stored at the start of the next V phase (beside the other slot's conv), the first
//     conversion of the next window follows in the same phase;
//   * barriers only drain the LDS counter (s_waitcnt lgkmcnt(0); s_barrier): weight prefetches and the window fetch stay in
//     flight across them.
// Phases of a slot: ... M | S V0 | M | V | M | V | M | V ... strictly alternating, slot B one barrier behind slot A for the
// whole launch.  Arithmetic, data layout and weight stream: resblock_dual.h / resblock_split.h (results bit-identical).
#pragma once
#include "resblock_dual.h"

namespace parrot {

// LDS-only barrier: every ds operation of this wave has completed, global loads / stores may stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class SCH, int NCH>
__global__ __launch_bounds__(512, 1) void resblock_pdual_kernel(const ResblockSplitParams p) {
    constexpr int NPC = SCH::NP, NTERM = SCH::NT, STEP_BYTES = SCH::NP * 1024;
    constexpr int C = 16 * NCH, WAVES_M = NCH / 2, WAVES_N = 4 / WAVES_M, NTW = 3, W = WAVES_N * NTW * 32;
    constexpr int OCT_BYTES = W * 16, CH_BYTES = 2 * OCT_BYTES, PIECE_BYTES = NCH * CH_BYTES;
    constexpr int WIN_BYTES = NPC * PIECE_BYTES + 2 * RBD_GUARD;
    static_assert(NCH == 2 || NCH == 4, "32 or 64 channels");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];  // 2 x WIN_BYTES
    const int K = p.k;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int win = wave >> 2, w4 = wave & 3;
    const int wm = w4 / WAVES_N, wn = w4 % WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;
    char* const smem_raw = smem_all + win * WIN_BYTES + RBD_GUARD;
    {   // guard bytes: zero
        const int t256 = tid & 255;
        char* g = smem_all + win * WIN_BYTES + (t256 < 128 ? 0 : NPC * PIECE_BYTES);
        *reinterpret_cast<uint2*>(g + t256 * 8) = uint2{0u, 0u};
    }
    int col[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) col[nt] = wn * (NTW * 32) + nt * 32 + l31;
    RBD_TRACE_INIT();
    RBD_MARK();  // 0: start

    // ---- the workgroup's list of window pairs: pair index blockIdx.x + it * gridDim.x, window 2 * pair + win ----------------
    const int total = p.tiles * p.B, npairs = (total + 1) >> 1;
    const int n_it = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    // state of the CURRENT window (scalars) and what the store of the PREVIOUS one needs
    int b = 0, tile = 0, t_base = 0, Tlim = 0;
    bool live = false, edge = false, tok[NTW];
    int pb = 0, ptile = 0;
    bool plive = false;
    auto set_window = [&](int it) __attribute__((always_inline)) {
        const int wid_raw = 2 * ((int)blockIdx.x + it * (int)gridDim.x) + win;
        live = wid_raw < total;
        const int wid = live ? wid_raw : total - 1;  // (an odd window count: the last slot re-computes the last window and stores nothing)
        b = wid / p.tiles;
        tile = wid - b * p.tiles;
        t_base = tile * p.TT - p.H;
        Tlim = p.row_len ? min(p.T, row_true_len(p.row_len[b], p.row_len_mul, p.row_len_add)) : p.T;
        edge = __builtin_amdgcn_readfirstlane((t_base < 0 || t_base + W > Tlim) ? 1 : 0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int t = t_base + col[nt];
            tok[nt] = t >= 0 && t < Tlim;
        }
    };

    f32x16 R[NTW], acc[NTW];
    const int row_bytes = p.T * 4;
    // R <- x window of the current window state (C/D layout: row 32 wm + (r & 3) + 8 (r >> 2) + 4 half, column lane & 31)
    auto fetch_window = [&]() __attribute__((always_inline)) {
        const RowTile xt = row_tile(p.x + (size_t)b * C * p.T, C, p.T);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int voff = tok[nt] ? (t_base + col[nt]) * 4 + 4 * half * row_bytes : (int)0x80000000;  // out of range -> 0
#pragma unroll
            for (int r = 0; r < 16; ++r) R[nt][r] = row_tile_load(xt, voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes);
        }
    };

    // ---- weights ----------------------------------------------------------------------------------------------------------------
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(p.wstream));
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<size_t>(p.wstream) >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)w_hi << 32) | w_lo), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    s16x8 A[2][NPC], Bv[2][NTW][NPC];
    auto load_a_piece = [&](s16x8 (&a)[NPC], int pc, int gstep) __attribute__((always_inline)) {
        a[pc] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16 + pc * 1024, gstep * STEP_BYTES, 0));
    };
    const int conv_steps = WAVES_M * NCH * K;
    int gbase = wm * NCH * K;

    const int center = (K - 1) / 2;
    const char* fbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) fbase[nt] = smem_raw + half * OCT_BYTES + col[nt] * 16;
    auto load_b = [&](s16x8 (&bb)[NTW][NPC], int ch, int shift16) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const char* src = fbase[nt] + shift16;
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bb[nt][pc] = *reinterpret_cast<const s16x8*>(src + (pc * PIECE_BYTES + ch * CH_BYTES));
        }
    };
    auto conv = [&](int dil) __attribute__((always_inline)) {
        load_b(Bv[0], 0, -center * dil * 16);
        for (int j = 0; j < K; ++j) {
            const int shift16 = (j - center) * dil * 16;
            const int n0 = (j + 1 < K) ? gbase + j + 1 : gbase + conv_steps;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int set = ch & 1;
                if (ch + 1 < NCH) load_b(Bv[set ^ 1], ch + 1, shift16);
                else load_b(Bv[set ^ 1], 0, shift16 + dil * 16);
                const int nx = (ch + 2 < NCH) ? gbase + (ch + 2) * K + j : n0 + (ch + 2 - NCH) * K;
#pragma unroll
                for (int t = 0; t < NTERM; ++t) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[nt] = mfma32<SCH>(A[set][SCH::pa(t)], Bv[set][nt][SCH::pb(t)], acc[nt]);
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        if (t == SCH::last_a(pc)) load_a_piece(A[set], pc, nx);
                }
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                int ds_left = NPC * NTW;
#pragma unroll
                for (int m = 0; m < NTERM * NTW; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    bool refetch = false;
#pragma unroll
                    for (int pc = 1; pc < NPC; ++pc) refetch = refetch || (m == (SCH::last_a(pc) + 1) * NTW - 1);
                    if (refetch) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else if (ds_left > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        --ds_left;
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        gbase += conv_steps;
    };
    auto bias_rows = [&](const float* __restrict__ bias, float (&bv)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half];
    };

    // store the result of the PREVIOUS window (held in acc, scale rs): the central TT columns, buffer-addressed
    float rs = 1.f;
    const bool has_acc = p.epi != EPI_STORE;
    const bool do_div = p.epi == EPI_ADD_DIV;
    auto store_prev = [&]() __attribute__((always_inline)) {
        const RowTile yt = row_tile(p.y + (size_t)pb * C * p.T, C, p.T);
        const float irs = 1.f / rs;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int c = col[nt] - p.H;
            const int t = ptile * p.TT + c;
            const bool ok = plive && c >= 0 && c < p.TT && t < p.T;
            const int voff = ok ? (4 * half * p.T + t) * 4 : (int)0x80000000;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[nt][r] * irs;
            if (has_acc) {
                float yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = row_tile_load(yt, voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = yv[r] + v[r];
                if (do_div) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] / p.div;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) row_tile_store(yt, v[r], voff, (32 * wm + (r & 3) + 8 * (r >> 2)) * row_bytes);
        }
    };

    // one window: V0 M V M [V M V M ...] -- ends after the last conv (the result stays in acc, scale rs); `more`: another
    // window follows in this slot (its input is fetched into R at the end of the last conv)
    auto window = [&](auto edge_c, int it, bool more) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        auto write_p = [&](const f32x16 (&v)[NTW], float mul) __attribute__((always_inline)) {
            const float m2 = mul * p.slope;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned qq[2][2][NPC];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x2 vv = {v[nt][8 * q + 4 * g + 2 * e], v[nt][8 * q + 4 * g + 2 * e + 1]};
                            const f32x2 a = vv * mul, bq = vv * m2;
                            float v0 = max_nc(a[0], bq[0]), v1 = max_nc(a[1], bq[1]);
                            if constexpr (EDGE) {
                                v0 = tok[nt] ? v0 : 0.f;
                                v1 = tok[nt] ? v1 : 0.f;
                            }
                            split_fast<SCH>(v0, v1, qq[g][e]);
                        }
                    char* dst = smem_raw + (2 * wm + q) * CH_BYTES + half * OCT_BYTES + col[nt] * 16;
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(qq[0][0][pc], qq[1][0][pc], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(qq[0][1][pc], qq[1][1][pc], false, false);
                        *reinterpret_cast<uint4*>(dst + pc * PIECE_BYTES) = uint4{r0[0], r1[0], r0[1], r1[1]};
                    }
                }
            }
        };
        RBD_MARK();  // (after the store of the previous window)
        write_p(R, SCH::XS);  // V0 (R freshly fetched: scale 1; out-of-range columns were loaded as zeros)
        RBD_MARK();
        lds_barrier();
        RBD_MARK();
        float rsc = 1.f;      // scale R is held in
        for (int m = 0; m < p.n_conv; m += 2) {
            float bv[16];
            const float s1 = SCH::XS * p.wsc[m], s2 = SCH::XS * p.wsc[m + 1];
            bias_rows(p.bias[m], bv);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = bv[r] * s1;
            conv(p.dil[m]);
            RBD_MARK();
            lds_barrier();
            RBD_MARK();
            write_p(acc, SCH::XS / s1);
            RBD_MARK();
            lds_barrier();
            RBD_MARK();
            bias_rows(p.bias[m + 1], bv);
            const float rr = s2 / rsc;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = fmaf(R[nt][r], rr, bv[r] * s2);
            conv(p.dil[m + 1]);
            RBD_MARK();
            rsc = s2;
            if (m + 2 < p.n_conv) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (EDGE) R[nt][r] = tok[nt] ? acc[nt][r] : 0.f;
                        else R[nt][r] = acc[nt][r];
                    }
                lds_barrier();
                RBD_MARK();
                write_p(R, SCH::XS / rsc);
                RBD_MARK();
                lds_barrier();
                RBD_MARK();
            }
        }
        // the result stays in acc; positions outside the row's true length are stored as zeros, as the one-window kernels do
        if constexpr (EDGE) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = tok[nt] ? acc[nt][r] : 0.f;
        }
        rs = rsc;
        pb = b; ptile = tile; plive = live;
        if (more) {  // R is dead: fetch the next window of this slot into it (lands during the barrier and the store)
            set_window(it + 1);
            fetch_window();
        }
    };

    set_window(0);
    fetch_window();
    if (win) lds_barrier();  // slot B runs one barrier behind slot A from here on
    for (int it = 0; it < n_it; ++it) {
        if (it > 0) store_prev();
        gbase = wm * NCH * K;  // the window starts at the launch's first conv again
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
            load_a_piece(A[0], pc, gbase);      // (tap 0, chunk 0)
            load_a_piece(A[1], pc, gbase + K);  // (tap 0, chunk 1)
        }
        const bool more = it + 1 < n_it;
        const bool e_now = edge;  // (set_window(it + 1) inside window() changes `edge`)
        if (e_now) window(std::true_type{}, it, more);
        else window(std::false_type{}, it, more);
        RBD_MARK();  // (next window's fetch issued)
        if (more) lds_barrier();  // end of the window's last M phase
        RBD_MARK();
    }
    store_prev();
    if (!win) lds_barrier();  // A's share of the barrier B spent at the start
}

template <class SCH>
inline hipError_t launch_resblock_pdual_s(int C, const ResblockSplitParams& p, int n_cus, hipStream_t s) {
    const size_t lds = 2 * ((size_t)SCH::NP * 2 * RBS_W * 32 + 2 * RBD_GUARD);
    auto kern = (C == 64) ? resblock_pdual_kernel<SCH, 4> : resblock_pdual_kernel<SCH, 2>;
    static DynLdsOnce lds_once[2];
    const int slot = (C == 64) ? 1 : 0;
    {
        hipError_t e = ensure_dyn_lds(lds_once[slot], reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    const int npairs = (p.tiles * p.B + 1) / 2;
    hipLaunchKernelGGL(kern, dim3(npairs < n_cus ? npairs : n_cus), dim3(512), lds, s, p);
    return hipGetLastError();
}
inline bool resblock_pdual_has(int scheme, int C) { return scheme == SchF16x3::ID && (C == 32 || C == 64); }
hipError_t launch_resblock_pdual_f16x3(int C, const ResblockSplitParams& p, int n_cus, hipStream_t s);  // csrc/tu_resblock_dual.hip

}  // namespace parrot
