// parrot_hip.hip -- C ABI + host orchestration of the MI355X Parrot-TTS synthesis path.
// See include/parrot_hip.h for the contract and the reference lines each entry point replaces.
#include "../../include/parrot_hip_debug.h"  // (parrot_hip.h + the test / profiling entry points)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "attn.h"
#include "conv_split.h"
#include "conv_split16.h"
#include "conv_mfma.h"
#include "conv_mfma16.h"
#include "conv_valu.h"
#include "kernels_misc.h"
#include "resblock_split.h"
#include "resblock_fused.h"

using namespace parrot;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(PARROT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)
#define TRY(expr)                \
    do {                         \
        int _r = (expr);         \
        if (_r != PARROT_OK) return _r; \
    } while (0)

extern "C" int parrot_abi_version(void) { return PARROT_ABI_VERSION; }
extern "C" const char* parrot_last_error(void) { return g_err.c_str(); }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char* base;
    size_t cap, off;
    bool ok;
    Arena(void* p, size_t n) : base((char*)p), cap(n), off(0), ok(true) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        size_t bytes = n * sizeof(T);
        if (base && off + bytes > cap) ok = false;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// Poison mode (tests only): PARROT_POISON_WS = nan | inf | 7f fills every caller-provided workspace / state / output buffer -- and
// the graph cache's staging buffers -- with that bit pattern at the top of each compute entry point, on the caller's stream.  A
// kernel that reads a byte nobody wrote then fails deterministically (NaN / 0 x inf / 3.4e38 in the result) instead of depending
// on what the allocator happened to leave behind.  Unset: no cost, no launches.
// ---------------------------------------------------------------------------------------------
static uint32_t poison_word() {
    static const uint32_t w = [] {
        const char* e = getenv("PARROT_POISON_WS");
        if (!e || !*e || !strcmp(e, "0")) return 0u;
        if (!strcmp(e, "inf")) return 0x7f800000u;
        if (!strcmp(e, "7f")) return 0x7f7f7f7fu;
        return 0x7fc00000u;  // "nan", "1", anything else
    }();
    return w;
}
static int poison(void* p, size_t bytes, hipStream_t s) {
    const uint32_t w = poison_word();
    if (!w || !p || bytes == 0) return PARROT_OK;
    if ((uintptr_t)p & 3) {  // (an unaligned view: bytes)
        HIP_TRY(hipMemsetAsync(p, 0x7f, bytes, s));
        return PARROT_OK;
    }
    const size_t words = bytes / 4;
    if (words) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p, (int)w, words, s));
    if (bytes & 3) HIP_TRY(hipMemsetAsync((char*)p + 4 * words, 0x7f, bytes & 3, s));
    return PARROT_OK;
}

// ---------------------------------------------------------------------------------------------
// optional per-launch timing of the conv kernel (HIP events on the launch stream), aggregated per
// tile configuration: feeds bench.py's roofline object.  Off by default.
// ---------------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    int cfg;
    double flops, bytes;
};
// (process-wide profiler: one mutex around its state; the flag is an atomic so un-profiled launches never take the lock)
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

static std::atomic<int> g_prof_row{-1};  // >= 0: only launches of this table row are timed (parrot_prof_begin_row)
static int prof_open(ProfRec& rec, int row, double flops, double bytes, hipStream_t s) {
    rec.a = rec.b = nullptr;
    const int only = g_prof_row.load();
    if (only >= 0 && row != only) return PARROT_OK;
    bool fresh = false;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_pool.empty()) fresh = true;
        else {
            rec.a = g_prof_pool.back().first;
            rec.b = g_prof_pool.back().second;
            g_prof_pool.pop_back();
        }
    }
    if (fresh) {
        HIP_TRY(hipEventCreate(&rec.a));
        HIP_TRY(hipEventCreate(&rec.b));
    }
    rec.cfg = row;
    rec.flops = flops;
    rec.bytes = bytes;
    HIP_TRY(hipEventRecord(rec.a, s));
    return PARROT_OK;
}
static int prof_close(ProfRec& rec, hipStream_t s) {
    if (!rec.a) return PARROT_OK;  // (row filtered out)
    HIP_TRY(hipEventRecord(rec.b, s));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(rec);
    return PARROT_OK;
}

extern "C" int parrot_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) g_prof_pool.push_back({r.a, r.b});
    g_prof.clear();
    g_prof_row = -1;
    g_prof_on = true;
    return PARROT_OK;
}
// The same, timing only the launches of ONE row of the table (the dominant kernel): a pair of event records around every launch
// of a step costs 0.6 ms at B = 64 and 0.4 ms of a 2 ms single-utterance step (they keep consecutive kernels from overlapping
// their ramp-up / drain), which is measurement overhead, not work of the path.
extern "C" int parrot_prof_begin_row(int32_t row) {
    if (row < 0) return fail(PARROT_E_INVALID, "prof_begin_row: row must be >= 0");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) g_prof_pool.push_back({r.a, r.b});
    g_prof.clear();
    g_prof_row = row;
    g_prof_on = true;
    return PARROT_OK;
}
// out[cfg*4 + {0,1,2,3}] = {launches, total ms, algorithmic flops, algorithmic bytes}; n_cfg rows.
extern "C" int parrot_prof_end(double* out, int32_t n_cfg) {
    g_prof_on = false;
    if (!out || n_cfg <= 0) return fail(PARROT_E_INVALID, "prof_end: bad output buffer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < n_cfg * 4; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.cfg < n_cfg) {
            out[r.cfg * 4 + 0] += 1.0;
            out[r.cfg * 4 + 1] += ms;
            out[r.cfg * 4 + 2] += r.flops;
            out[r.cfg * 4 + 3] += r.bytes;
        }
    }
    return PARROT_OK;
}

// ---------------------------------------------------------------------------------------------
// conv plan
// ---------------------------------------------------------------------------------------------
struct parrot_conv {
    parrot_conv_desc d{};
    int groups = 1;
    int M = 0, Mg = 0, Cout = 0, Cin = 0;  // Cin per group
    int kk = 1, dil = 1, pad_left = 0, u = 1;
    int cfg = 0;
    int nchunks = 0, n_it = 0;
    float* wfrag = nullptr;
    float* bias = nullptr;
    int prec = 0;              // 0: exact fp32 MFMA, else the split scheme of conv_split.h (PARROT_PREC_*: 16-bit MFMAs, fp32 accumulate)
    uint16_t* wfrag16 = nullptr;  // [m_tile][chunk*tap][piece][lane][8 x 16 bit]
    int n_it16 = 0;
    float wscale = 1.f;        // power-of-two weight scale inside the fp16 pieces (1 for bf16 schemes)
    bool mfma16 = false;       // split plan packed for conv_split16_kernel (16x16x32 MFMA, 32-channel chunks)
    int* err_flag = nullptr;   // device flag of the owning model (set on a non-finite tanh output: conv_post)
    bool late_res = false;     // add the residual in the epilogue instead of folding it into the accumulator init (TTE layers)
    int valu_kind = 0;         // 1: conv1_valu_kernel<7>, 2: convt_valu_kernel<16,4,2,1> (conv_valu.h); weights in their original layout
    float* wraw = nullptr;

    ~parrot_conv() {
        if (wraw) (void)hipFree(wraw);
        if (wfrag16) (void)hipFree(wfrag16);
        if (wfrag) (void)hipFree(wfrag);
        if (bias) (void)hipFree(bias);
    }
    int out_len(int Tin) const {
        if (!d.transposed) return Tin + 2 * d.padding - d.dilation * (d.k - 1);
        return (Tin - 1) * d.stride - 2 * d.padding + d.k;
    }
};

// Process-wide DEFAULTS, read once by every *_create (the handle keeps its own copy and is immutable afterwards, so
// handles stay re-entrant; changing a default never affects a live handle).  Atomics: setters may race with creates.
static std::atomic<int> g_default_prec{-1};
static int parse_prec(const char* e) {
    if (!e) return PARROT_PREC_F16X3;
    if (!strcmp(e, "f32") || !strcmp(e, "0")) return PARROT_PREC_F32;
    if (!strcmp(e, "bf16x6") || !strcmp(e, "1")) return PARROT_PREC_BF16X6;
    if (!strcmp(e, "bf16") || !strcmp(e, "3")) return PARROT_PREC_BF16;
    if (!strcmp(e, "f16") || !strcmp(e, "4")) return PARROT_PREC_F16;
    return PARROT_PREC_F16X3;
}
static int default_prec() {
    int v = g_default_prec.load();
    if (v < 0) {
        v = parse_prec(getenv("PARROT_PRECISION"));
        g_default_prec.store(v);
    }
    return v;
}
extern "C" int parrot_set_default_precision(int32_t prec) {
    if (prec < 0 || prec > PARROT_PREC_F16) return fail(PARROT_E_INVALID, "set_default_precision: PARROT_PREC_* (0..4)");
    g_default_prec.store(prec);
    return PARROT_OK;
}

// Fused whole-ResBlock kernels: 0 off, 1 every eligible stage, 2 (default) all but the exact-fp32 32-channel kernel
// (resblock_fused.h; slower than layer by layer).  PARROT_FUSED / parrot_set_fused_resblocks set the default for
// handles created afterwards.
static std::atomic<int> g_fused{-1};
static int fused_mode() {
    int v = g_fused.load();
    if (v < 0) {
        const char* e = getenv("PARROT_FUSED");
        v = e ? atoi(e) : 2;
        if (v < 0 || v > 2) v = 2;
        g_fused.store(v);
    }
    return v;
}

static inline uint16_t f16_rn_host(float x) {  // round-to-nearest-even, overflow -> inf (what v_cvt_pk_f16_f32 does)
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static inline float f16_to_f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
static inline uint16_t bf16_rn_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Split-scheme weight pieces (conv_split.h).  fp16 schemes: the layer's weights are scaled by the power of two that puts
// max|w| into [2^14, 2^15), so the second piece of every weight that matters is a normal fp16 number.
static float f16_weight_scale(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int e;
    (void)std::frexp(mx, &e);  // mx = m * 2^e, m in [0.5, 1)
    return std::ldexp(1.f, 15 - e);  // mx * scale in [2^14, 2^15)
}
static void split_weight(float v, int scheme, float wscale, uint16_t (&h)[3]) {
    h[0] = h[1] = h[2] = 0;
    if (scheme_is_f16(scheme)) {
        const float vs = v * wscale;
        h[0] = f16_rn_host(vs);
        if (scheme == PARROT_PREC_F16X3) h[1] = f16_rn_host(vs - f16_to_f(h[0]));
        return;
    }
    h[0] = bf16_rn_host(v);
    if (scheme == PARROT_PREC_BF16X6) {
        const float r1 = v - bf16_to_f(h[0]);
        h[1] = bf16_rn_host(r1);
        h[2] = bf16_rn_host(r1 - bf16_to_f(h[1]));
    }
}

static int g_num_cus = 256;  // (MI355X; refreshed from the device at the first *_create)
static void query_device() {
    static bool done = false;
    if (done) return;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) g_num_cus = n;
    done = true;
}

static int choose_cfg(int M, int k) {
    if (M <= 16) return 6;
    if (M <= 32) return 2;
    if (M <= 64) return 1;
    return (k <= 3) ? 3 : 0;
}

// Build a plan.  `groups` > 1: torch grouped-conv weight layout (c_out, c_in/groups, k), d.c_in = TOTAL.
static bool valu_kernels_enabled() {
    static const bool on = [] { const char* e = getenv("PARROT_VALU_KERNELS"); return !e || atoi(e) != 0; }();
    return on;
}
static bool mfma16_enabled() {
    static const bool on = [] { const char* e = getenv("PARROT_MFMA16"); return !e || atoi(e) != 0; }();
    return on;
}
// PARROT_SMALL_TILES: 2 (default) = 64-column tiles for underfilled launches and 64-row tiles for far-underfilled ones, 1 = the
// 64-column tiles only, 0 = neither
static int small_tiles_mode() {
    static const int m = [] { const char* e = getenv("PARROT_SMALL_TILES"); return e ? atoi(e) : 2; }();
    return m;
}
// Per-handle modes (parrot_*_create_ex) reach the plan builders through a THREAD-LOCAL scope, never through the process
// defaults: a create on one thread cannot leak its precision / fusion / merge mode into a parrot_conv_create or another
// *_create running on a second thread, and a concurrent parrot_set_* is neither seen half-way nor reverted afterwards.
static thread_local int tl_prec = -1, tl_fused = -1, tl_merge = -1;
struct CreateScope {
    int p0, f0, m0;
    CreateScope(int prec, int fused, int merge) : p0(tl_prec), f0(tl_fused), m0(tl_merge) {
        if (prec >= 0) tl_prec = prec;
        if (fused >= 0) tl_fused = fused;
        if (merge >= 0) tl_merge = merge ? 1 : 0;
    }
    ~CreateScope() { tl_prec = p0; tl_fused = f0; tl_merge = m0; }
};
static int create_prec() { return tl_prec >= 0 ? tl_prec : default_prec(); }
static int create_fused() { return tl_fused >= 0 ? tl_fused : fused_mode(); }

static int conv_build(parrot_conv** out, const parrot_conv_desc* d, int groups, const float* w, const float* bias, bool allow16 = true) {
    if (!out || !d || !w) return fail(PARROT_E_INVALID, "conv_create: null argument");
    if (d->c_in <= 0 || d->c_out <= 0 || d->k <= 0 || d->dilation <= 0 || groups <= 0 || d->c_in % groups || d->c_out % groups)
        return fail(PARROT_E_INVALID, "conv_create: bad dimensions");
    std::unique_ptr<parrot_conv> c(new parrot_conv());
    c->d = *d;
    c->groups = groups;
    c->Cout = d->c_out;
    c->Cin = d->c_in / groups;
    int dmin = 0;
    if (d->transposed) {
        if (d->stride > 64) return fail(PARROT_E_UNSUPPORTED, "conv_create: transposed stride > 64");
        if (groups != 1 || d->dilation != 1 || d->stride <= 0) return fail(PARROT_E_UNSUPPORTED, "conv_create: transposed conv needs groups=1, dilation=1");
        // polyphase gather form: output tau = t*u + r uses taps kappa = r + p - delta*u, input t + delta
        const int u = d->stride, p = d->padding, k = d->k;
        int dlo = 1 << 30, dhi = -(1 << 30);
        for (int r = 0; r < u; ++r)
            for (int kap = 0; kap < k; ++kap)
                if ((r + p - kap) % u == 0) {
                    int dl = (r + p - kap) / u;
                    dlo = std::min(dlo, dl);
                    dhi = std::max(dhi, dl);
                }
        if (dlo > dhi) return fail(PARROT_E_INVALID, "conv_create: transposed conv has no taps");
        dmin = dlo;
        c->u = u;
        c->kk = dhi - dlo + 1;
        c->dil = 1;
        c->pad_left = -dlo;
        c->M = d->c_out * u;
    } else {
        if (d->stride > 1) return fail(PARROT_E_UNSUPPORTED, "conv_create: strided Conv1d is not on the path");
        c->u = 1;
        c->kk = d->k;
        c->dil = d->dilation;
        c->pad_left = d->padding;
        c->M = d->c_out;
    }
    c->Mg = c->M / groups;
    if ((c->kk - 1) * c->dil > CONV_HALO) return fail(PARROT_E_UNSUPPORTED, "conv_create: (k-1)*dilation exceeds the LDS halo (64)");
    c->cfg = (d->tile_cfg >= 0) ? d->tile_cfg : choose_cfg(c->Mg, c->kk);
    if (c->cfg >= NUM_TILE_CFGS) return fail(PARROT_E_INVALID, "conv_create: tile_cfg out of range");
    if (c->cfg == 6 && (d->transposed || groups != 1 || c->M > 16)) {
        if (d->tile_cfg == 6) return fail(PARROT_E_UNSUPPORTED, "conv_create: the 16-row tile needs a plain conv with <= 16 output channels");
        c->cfg = 2;
    }
    const TileCfg t = tile_cfg(c->cfg);
    if (groups > 1 && c->Mg % t.bm) return fail(PARROT_E_UNSUPPORTED, "conv_create: rows per group must be a multiple of the tile height");
    const int CI = t.ci, QN = CI / 8;
    c->nchunks = (c->Cin + CI - 1) / CI;
    c->n_it = c->nchunks * c->kk * QN;
    const int mtiles = (c->M + t.bm - 1) / t.bm * (t.bm / 32);
    const size_t nfl = ((size_t)mtiles * c->n_it + 1) * 256;  // +1 group: the kernel prefetches one past the end
    std::vector<float> pk(nfl, 0.f);
    const int k = d->k, Cing = c->Cin;
    if (c->cfg == 6) {  // 16x16x4 fragments: [chunk][tap][lane][4]: row = lane&15, channel = 16*chunk + 4*e + (lane>>4)
        c->n_it = c->nchunks * c->kk;
        pk.assign(((size_t)c->n_it + 1) * 256, 0.f);
        for (int ch = 0; ch < c->nchunks; ++ch)
            for (int j = 0; j < c->kk; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int m = lane & 15, i = ch * 16 + 4 * e + (lane >> 4);
                        if (m < c->M && i < Cing) pk[((size_t)ch * c->kk + j) * 256 + lane * 4 + e] = w[((size_t)m * Cing + i) * k + j];
                    }
    } else
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ch = 0; ch < c->nchunks; ++ch)
            for (int j = 0; j < c->kk; ++j)
                for (int q = 0; q < QN; ++q) {
                    float* g = pk.data() + ((size_t)mt * c->n_it + ((size_t)ch * c->kk + j) * QN + q) * 256;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int m = mt * 32 + (lane & 31);
                            const int i = ch * CI + 8 * q + 2 * e + (lane >> 5);
                            float v = 0.f;
                            if (m < c->M && i < Cing) {
                                if (d->transposed) {
                                    const int o = m / c->u, r = m % c->u;
                                    const int kap = r + d->padding - (j + dmin) * c->u;
                                    if (kap >= 0 && kap < k) v = w[((size_t)i * d->c_out + o) * k + kap];
                                } else {
                                    v = w[((size_t)m * Cing + i) * k + j];
                                }
                            }
                            g[lane * 4 + e] = v;
                        }
                }
    // weight accessor shared by both packings: W'(m, i, j) of the GEMM view (0 outside the real extents)
    auto wval = [&](int m, int i, int j) -> float {
        if (m >= c->M || i >= Cing) return 0.f;
        if (d->transposed) {
            const int o = m / c->u, r = m % c->u;
            const int kap = r + d->padding - (j + dmin) * c->u;
            return (kap >= 0 && kap < k) ? w[((size_t)i * d->c_out + o) * k + kap] : 0.f;
        }
        return w[((size_t)m * Cing + i) * k + j];
    };
    const int want_prec = (d->precision >= 0) ? d->precision : create_prec();
    if (want_prec > PARROT_PREC_F16) return fail(PARROT_E_INVALID, "conv_create: unknown precision");
    // split kernels: at 32 rows the exact kernel is as fast (measured); the slab fetch needs whole 16-channel chunks
    // and evaluates the leaky ReLU as max(v, slope * v).  Everything else runs on the exact kernel (same results class).
    const bool slope_ok = d->pre_act != PRE_LRELU || (d->pre_slope >= 0.f && d->pre_slope <= 1.f);
    if (want_prec >= 1 && c->Mg >= 32 && d->tile_cfg < 0 && c->Cin % 16 == 0 && slope_ok) {
        // split plan: 16 channels per chunk, one MFMA k-step per tap; [m_tile][chunk*tap][piece][lane][8]
        c->prec = want_prec;
        const int NP = scheme_pieces(want_prec);
        const bool f16 = scheme_is_f16(want_prec);
        c->cfg = (c->Mg <= 32) ? 2 : (c->Mg <= 64) ? 1 : 0;  // exact-kernel tile ids with the same block shapes
        const TileCfg t16 = tile_cfg(c->cfg);
        if (groups > 1 && c->Mg % t16.bm) return fail(PARROT_E_UNSUPPORTED, "conv_create: rows per group must be a multiple of the tile height");
        if (f16) c->wscale = f16_weight_scale(w, (size_t)d->c_in / groups * d->c_out * d->k);
        // wide plain convs: the 16x16x32 kernel (conv_split16.h): 32-channel chunks, [m16 tile][chunk*tap][piece][lane][8]
        c->mfma16 = allow16 && mfma16_enabled() && split16_has(want_prec, c->kk) && !d->transposed && groups == 1 && c->Cin % 32 == 0 && c->M >= 64;
        size_t n16 = 0;
        std::vector<uint16_t> pk16;
        if (c->mfma16) {
            int bm16, bn16;
            split16_tile(c->M >= 128 ? 0 : 1, bm16, bn16);
            c->nchunks = c->Cin / 32;
            c->n_it16 = c->nchunks * c->kk;
            const int mt = (c->M + bm16 - 1) / bm16 * (bm16 / 16);
            const size_t step_h = (size_t)NP * 512;
            n16 = ((size_t)mt * c->n_it16 + 1) * step_h;
            if (n16 * sizeof(uint16_t) >= ((size_t)1 << 31)) return fail(PARROT_E_UNSUPPORTED, "conv_create: packed weight stream larger than 2 GiB");
            pk16.assign(n16, 0);
            for (int m16 = 0; m16 < mt; ++m16)
                for (int ch = 0; ch < c->nchunks; ++ch)
                    for (int j = 0; j < c->kk; ++j) {
                        uint16_t* g = pk16.data() + ((size_t)m16 * c->n_it16 + (size_t)ch * c->kk + j) * step_h;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                uint16_t h[3];
                                split_weight(wval(m16 * 16 + (lane & 15), ch * 32 + 8 * (lane >> 4) + e, j), want_prec, c->wscale, h);
                                for (int pc = 0; pc < NP; ++pc) g[pc * 512 + lane * 8 + e] = h[pc];
                            }
                    }
        } else {
        c->nchunks = (c->Cin + 15) / 16;
        c->n_it16 = c->nchunks * c->kk;
        const int mt16 = (c->M + t16.bm - 1) / t16.bm * (t16.bm / 32);
        const size_t step_h = (size_t)NP * 512;  // 16-bit words per step: NP pieces x 64 lanes x 8
        n16 = ((size_t)mt16 * c->n_it16 + 1) * step_h;  // (+1 pad step)
        if (n16 * sizeof(uint16_t) >= ((size_t)1 << 31)) return fail(PARROT_E_UNSUPPORTED, "conv_create: packed weight stream larger than 2 GiB");
        pk16.assign(n16, 0);
        for (int mt = 0; mt < mt16; ++mt)
            for (int ch = 0; ch < c->nchunks; ++ch)
                for (int j = 0; j < c->kk; ++j) {
                    uint16_t* g = pk16.data() + ((size_t)mt * c->n_it16 + (size_t)ch * c->kk + j) * step_h;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            uint16_t h[3];
                            split_weight(wval(mt * 32 + (lane & 31), ch * 16 + 8 * (lane >> 5) + e, j), want_prec, c->wscale, h);
                            for (int pc = 0; pc < NP; ++pc) g[pc * 512 + lane * 8 + e] = h[pc];
                        }
                }
        }
        HIP_TRY(hipMalloc((void**)&c->wfrag16, n16 * sizeof(uint16_t)));
        HIP_TRY(hipMemcpy(c->wfrag16, pk16.data(), n16 * sizeof(uint16_t), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMalloc((void**)&c->wfrag, pk.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->wfrag, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // the two narrowest vocoder layers stream through plain fp32 FMA kernels (conv_valu.h) in either precision mode
    const bool slope01 = d->pre_act != PRE_LRELU || (d->pre_slope >= 0.f && d->pre_slope <= 1.f);
    if (d->tile_cfg < 0 && groups == 1 && slope01 && d->dilation == 1 && valu_kernels_enabled()) {
        if (!d->transposed && d->c_out == 1 && ((d->k == 7 && d->padding == 3) || (d->k == 1 && d->padding == 0)) &&
            (d->act == ACT_NONE || d->act == ACT_TANH))
            c->valu_kind = 1;  // conv_post; the duration predictor's Linear(256 -> 1)
        if (d->transposed && d->c_out == 16 && d->k == 4 && d->stride == 2 && d->padding == 1 && d->act == ACT_NONE) c->valu_kind = 2;
        if (c->valu_kind) {
            const size_t n = (size_t)d->c_in * d->c_out * d->k;
            HIP_TRY(hipMalloc((void**)&c->wraw, n * sizeof(float)));
            HIP_TRY(hipMemcpy(c->wraw, w, n * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (bias) {
        HIP_TRY(hipMalloc((void**)&c->bias, (size_t)d->c_out * sizeof(float)));
        HIP_TRY(hipMemcpy(c->bias, bias, (size_t)d->c_out * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = c.release();
    return PARROT_OK;
}

// Operand planes between conv_split16 layers (conv_split16.h): `xplane` replaces x as the input (the values are the same: the
// producer applied this layer's own leaky ReLU / scale / split); `yplane` is written beside y -- or instead of it (plane_only) --
// with the NEXT layer's slope.  Dense batch rows of pieces x 2 C T bytes.
struct PlaneArgs {
    const void* xplane = nullptr;
    void* yplane = nullptr;
    float yslope = 1.f;
    int plane_only = 0;
};
// can layer `c` take its input from / write its output to an operand plane?  (conv_split16 plans of the MRF: k = 7 / 11)
static bool plane_ok(const parrot_conv* c) { return c && c->mfma16 && c->prec >= 1 && (c->kk == 7 || c->kk == 11) && c->M % 16 == 0 && c->Cin % 32 == 0; }
static size_t plane_row_bytes(int prec, int C, int T) { return (size_t)(prec == PARROT_PREC_F16X3 ? 2 : 1) * 2 * C * T; }

// x_bstride / y_bstride / res_bstride in elements; <= 0 means dense.
static int conv_launch(const parrot_conv* c, const float* x, const float* res, float* y, int B, int Tin, int epi, float div,
                       long x_bstride, long y_bstride, long res_bstride, hipStream_t s, const int32_t* row_len = nullptr,
                       int row_len_mul = 1, int row_len_add = 0, const PlaneArgs* pl = nullptr) {
    if (B <= 0 || Tin <= 0) return fail(PARROT_E_INVALID, "conv_run: empty batch or sequence");
    const int Tout = c->out_len(Tin);
    if (Tout <= 0) return fail(PARROT_E_INVALID, "conv_run: sequence shorter than the kernel");
    if (c->valu_kind && !res && epi == EPI_STORE && x_bstride <= 0 && y_bstride <= 0 && (double)c->d.c_in * Tin * 4.0 < 2147483648.0) {
        ConvValuParams q{};
        q.x = x; q.w = c->wraw; q.bias = c->bias; q.y = y;
        q.B = B; q.Cin = c->d.c_in; q.Tin = Tin; q.Tout = Tout;
        q.slope = c->d.pre_act == PRE_LRELU ? c->d.pre_slope : 1.f;
        q.act = c->d.act;
        q.row_len = row_len; q.row_len_mul = row_len_mul; q.row_len_add = row_len_add;
        q.err = c->err_flag;
        ProfRec rec{};
        const double macs = (double)B * c->d.c_out * c->d.c_in * c->d.k * (c->d.transposed ? (double)Tin : (double)Tout);
        if (g_prof_on) TRY(prof_open(rec, NUM_TILE_CFGS + 6 + c->valu_kind, 2.0 * macs, 4.0 * B * ((double)c->d.c_in * Tin + (double)c->d.c_out * Tout), s));
        if (c->valu_kind == 1 && c->d.k == 7 && (Tin & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0)
            hipLaunchKernelGGL(conv1_valu7_vec_kernel, dim3((Tout + 1023) / 1024, B), dim3(256), 0, s, q);
        else if (c->valu_kind == 1 && c->d.k == 7) hipLaunchKernelGGL(conv1_valu_kernel<7>, dim3((Tout + 1023) / 1024, B), dim3(256), 0, s, q);
        else if (c->valu_kind == 1) hipLaunchKernelGGL(linear1_valu_kernel, dim3((Tout + 63) / 64, B), dim3(256), 0, s, q);
        else hipLaunchKernelGGL((convt_valu_kernel<16, 4, 2, 1>), dim3((Tin + 255) / 256, B), dim3(256), 0, s, q);
        HIP_TRY(hipGetLastError());
        if (g_prof_on) TRY(prof_close(rec, s));
        return PARROT_OK;
    }
    ConvParams p{};
    p.x = x; p.wfrag = c->wfrag; p.bias = c->bias; p.res = res; p.y = y;
    p.B = B; p.Cin = c->Cin; p.Tin = Tin; p.M = c->M; p.Cout = c->Cout;
    p.Ncols = (c->u > 1) ? (Tout + c->u - 1) / c->u : Tout;
    p.Tout = Tout;
    p.k = c->kk; p.dil = c->dil; p.pad_left = c->pad_left;
    p.nchunks = c->nchunks; p.n_it = c->n_it;
    p.pre = c->d.pre_act; p.pre_slope = c->d.pre_slope; p.act = c->d.act;
    p.epi = epi; p.div = div; p.u = c->u; p.u_inv16 = (65536 + c->u - 1) / c->u;
    p.groups = c->groups; p.Mg = c->Mg;
    p.row_len = row_len; p.row_len_mul = row_len_mul; p.row_len_add = row_len_add;
    p.acc_scale = p.out_scale = 1.f;
    p.lean = 1;  // conv_split_kernel: the buffer-addressed prologue / epilogue instantiations for plain convs (conv_lean_ok)
    p.n_cus = g_num_cus;
    p.fold_res = c->late_res ? 0 : 1;
    if (pl && (pl->xplane || pl->yplane)) {
        if (!plane_ok(c) || (pl->yplane && epi != EPI_STORE)) return fail(PARROT_E_INVALID, "conv_run: operand planes need a conv_split16 layer (k = 7 / 11) and EPI_STORE");
        p.xplane = pl->xplane; p.yplane = pl->yplane;
        p.xplane_bstride = (long)plane_row_bytes(c->prec, c->Cin, Tin);
        p.yplane_bstride = (long)plane_row_bytes(c->prec, c->M, Tout);
        p.yplane_slope = pl->yslope; p.plane_only = pl->plane_only;
    }
    p.x_bstride = x_bstride > 0 ? x_bstride : (long)c->d.c_in * Tin;
    p.y_bstride = y_bstride > 0 ? y_bstride : (long)c->Cout * Tout;
    p.res_bstride = res_bstride > 0 ? res_bstride : p.y_bstride;
    int cfg = c->cfg;
    if (c->prec == 0 && (cfg == 0 || cfg == 3) && p.Ncols <= 64 && tile_cfg(4).ci == tile_cfg(cfg).ci) cfg = 4;  // same packing, narrower tile
    if (c->prec >= 1) {
        p.wfrag = reinterpret_cast<const float*>(c->wfrag16);
        p.n_it = c->n_it16;
        p.acc_scale = scheme_xs(c->prec) * c->wscale;
        p.out_scale = 1.f / p.acc_scale;
        // 32-bit byte offsets inside one batch row (buffer addressing of the slab fetch)
        if ((double)c->Cin * Tin * 4.0 >= 2147483648.0) return fail(PARROT_E_UNSUPPORTED, "conv_run: batch row larger than 2 GiB");
    }
    TileCfg t = tile_cfg(cfg);
    int variant16 = 0;
    const bool small_tiles = small_tiles_mode() >= 1;
    if (c->mfma16) {
        variant16 = c->M >= 128 ? 0 : 1;
        split16_tile(variant16, t.bm, t.bn, c->kk);
        // small batches: a launch that would not give every CU a workgroup takes the 64-column tiles (2-3x the workgroups,
        // a half / third of the MFMAs per step: the per-launch latency is what counts there, not the operand reuse)
        if (small_tiles && (long)((p.Ncols + t.bn - 1) / t.bn) * B * ((c->M + t.bm - 1) / t.bm) < g_num_cus) {
            variant16 += 2;
            // ... and 64-row workgroups for the 128-row layers when even that leaves more than half of the CUs idle (one to four
            // utterances): four waves per workgroup, one per SIMD, twice the workgroups -- single utterance 2.08 -> 2.00 ms, B = 4
            // 2.63 -> 2.57 ms; 32-row workgroups (2 waves, four slab items per thread) measured slower (2.11 / 2.69 ms)
            split16_tile(variant16, t.bm, t.bn, c->kk);
            if (small_tiles_mode() >= 2 && (long)((p.Ncols + t.bn - 1) / t.bn) * B * ((c->M + t.bm - 1) / t.bm) * 2 <= g_num_cus && c->M >= 128) variant16 = 3;
        } else if (small_tiles && p.Ncols <= 64) variant16 += 2;  // sequences of <= 64 steps (the TTE encoder side) would leave half of a 128-column tile empty
        else if (variant16 == 0 && split16_wide_fits(p.Ncols, B, (c->M + 127) / 128, g_num_cus)) variant16 = 4;  // 128 x 160: no half-empty last round
        split16_tile(variant16, t.bm, t.bn, c->kk);
        // conv_split16_kernel addresses the (M, Tout) output / residual tile of a batch row with 32-bit byte offsets (RowTile)
        if ((double)c->M * Tout * 4.0 >= 2147483648.0) return fail(PARROT_E_UNSUPPORTED, "conv_run: output row tile larger than 2 GiB");
        {   // rows that start on 16-byte boundaries take the 16-byte epilogue (round-4 A/B on one box, profiles/r04a_*: the dominant
            // kernel 231.0 us per launch with it, 231.1 us without -- the C/D-layout stores were not what bounds the epilogue)
            auto al16 = [](const void* q, long stride) { return (reinterpret_cast<size_t>(q) & 15) == 0 && (stride & 3) == 0; };
            p.epi16 = (Tout % 4 == 0) && al16(y, p.y_bstride) && (!res || al16(res, p.res_bstride));
        }
    } else if (c->prec >= 1) {
        // 1x1 convs (Linear layers) have one MFMA step per barrier: the 128x64 / 3-waves-per-SIMD variant hides that
        // better (76 vs 61 TF on the qkv projection); every other layer is faster on the 64x64 wave tile
        // (and so are sequences of <= 64 steps -- the TTE encoder side -- which would leave half of a 128-column tile empty)
        const bool few = small_tiles && cfg == 0 && (c->kk == 3 || c->kk == 9) && (long)((p.Ncols + 127) / 128) * B * ((c->M + 127) / 128) < g_num_cus;
        variant16 = (cfg == 2) ? 3 : (cfg == 0 && (c->kk == 1 || p.Ncols <= 64 || few)) ? 2 : cfg;
        split_tile(variant16, t.bm, t.bn);
    }
    p.tiles_n = (p.Ncols + t.bn - 1) / t.bn;
    ProfRec rec{};
    if (g_prof_on) {
        // algorithmic work of the layer (real taps only; DESIGN.md "roofline accounting")
        const double macs = (double)B * c->d.c_out * c->Cin * c->d.k * (c->d.transposed ? (double)Tin : (double)Tout);
        const double elems = (double)B * ((double)c->d.c_in * Tin + (double)c->Cout * Tout * (1 + (res ? 1 : 0) + (epi != EPI_STORE ? 1 : 0)));
        const int row = c->mfma16 ? (variant16 == 4 ? NUM_TILE_CFGS + 12 : NUM_TILE_CFGS + 9 + (variant16 & 1))
                                  : (c->prec >= 1) ? (variant16 >= 2 ? NUM_TILE_CFGS + 1 + variant16 : NUM_TILE_CFGS + cfg) : cfg;  // split rows follow the exact ones
        TRY(prof_open(rec, row, 2.0 * macs, 4.0 * (elems + (double)c->d.c_out * c->Cin * c->d.k), s));
    }
    HIP_TRY(c->mfma16 ? launch_conv_split16(c->prec, variant16, p, s) : c->prec >= 1 ? launch_conv_split(c->prec, variant16, p, s) : (cfg == 6 ? launch_conv_mfma16(p, s) : launch_conv(cfg, p, s)));
    if (c->d.act == ACT_TANH) {  // dense (B, Cout, Tout) output assumed for the tanh layers (conv_post)
        const size_t n = (size_t)B * c->Cout * Tout;
        hipLaunchKernelGGL(tanh_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, n, c->err_flag);
        HIP_TRY(hipGetLastError());
    }
    if (g_prof_on) TRY(prof_close(rec, s));
    return PARROT_OK;
}

extern "C" int parrot_conv_create(parrot_conv_t** out, const parrot_conv_desc* d, const float* w_host, const float* bias_host) {
    return conv_build(out, d, 1, w_host, bias_host);
}
extern "C" void parrot_conv_destroy(parrot_conv_t* c) { delete c; }
extern "C" int parrot_conv_out_len(const parrot_conv_t* c, int32_t T_in) { return c ? c->out_len(T_in) : PARROT_E_INVALID; }
extern "C" int parrot_conv_num_tile_cfgs(void) { return NUM_TILE_CFGS; }
extern "C" int parrot_conv_run(parrot_conv_t* c, const float* x, const float* res, float* y, int32_t B, int32_t T_in,
                               int32_t epilogue, float div, void* stream) {
    if (!c || !x || !y) return fail(PARROT_E_INVALID, "conv_run: null argument");
    if (epilogue < 0 || epilogue > 2) return fail(PARROT_E_INVALID, "conv_run: bad epilogue");
    if (epilogue == EPI_STORE && y != x && y != res && c->out_len(T_in) > 0)
        TRY(poison(y, (size_t)B * c->Cout * c->out_len(T_in) * sizeof(float), (hipStream_t)stream));
    return conv_launch(c, x, res, y, B, T_in, epilogue, div, 0, 0, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// self test: MFMA fragment layout
// ---------------------------------------------------------------------------------------------
extern "C" int parrot_selftest(void* stream) {
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 64 * 16 * sizeof(float)));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    std::vector<float> h(64 * 16);
    hipError_t e = hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PARROT_E_HIP, hipGetErrorString(e));
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            const float want = (float)(col + 1) * (float)(1001 * row + 100000);
            if (h[lane * 16 + r] != want) {
                char buf[160];
                snprintf(buf, sizeof buf, "mfma 32x32x2 layout probe: lane %d reg %d got %g want %g", lane, r, h[lane * 16 + r], want);
                return fail(PARROT_E_UNSUPPORTED, buf);
            }
        }
    return PARROT_OK;
}

// ---------------------------------------------------------------------------------------------
// vocoder
// ---------------------------------------------------------------------------------------------
struct parrot_voc {
    parrot_voc_cfg cfg{};
    float* dict = nullptr;
    float* spkr = nullptr;
    int* err = nullptr;
    std::unique_ptr<parrot_conv> conv_pre, conv_post;
    std::vector<std::unique_ptr<parrot_conv>> ups, rb;
    std::vector<uint16_t*> rb_stream;  // per (stage, kernel): concatenated split weight stream of the block (or null)
    std::vector<float> rb_wsc;         // per resblock conv: weight scale inside that stream (fp16 schemes; else 1)
    std::vector<size_t> rb_conv_halves;  // 16-bit words per conv in that stream
    // whole-MRF launches (resblock_split.h, MRF instantiation): mrf_ok[stage] = every branch of the stage has a pair-kernel weight
    // stream (rb_stream) and the window survives the widest branch's reach
    std::vector<char> mrf_ok;
    int up_total = 1;
    bool odd_stage = false;            // some stage has odd kernel_size - rate: T_out = T u + 1 there (no constant hop)
    int scheme = 0;                    // PARROT_PREC_* captured at create (immutable afterwards)
    int fused = 2;                     // fused-ResBlock mode captured at create
    bool planes = true;                // operand planes between the layer-by-layer convs of a pair (PARROT_PLANES, read at create)
    // MRF branch concurrency: the n_kernels ResBlocks of a stage are independent chains until the final sum, so each runs
    // on its own HIP stream (the caller's + side streams owned by the handle), forked / joined with events; the final
    // accumulating launches are ordered with events (sum order j = 0, 1, 2 as models.py:100-106).  One branch's
    // launch tails, prologues and epilogues then overlap another branch's main loops.
    int mrf_streams = 1;
    bool mrf_auto = true;
    std::mutex side_mu;  // the side streams / events are the handle's: concurrent callers enqueue their fork-join sequences one at a time
    struct StreamSet {
        hipStream_t side[PARROT_MAX_KERNELS] = {};
        hipEvent_t ev_fork = nullptr, ev_last[PARROT_MAX_KERNELS] = {};
    } ss[1];
    // chunk lanes of the chunk-streamed forward (lane 0 is the caller's stream)
    static constexpr int MAX_LANES = 4;
    hipStream_t lane_stream[MAX_LANES] = {};
    hipEvent_t ev_lane_fork = nullptr, ev_lane_join[MAX_LANES] = {};
    // stage events (parrot_voc_wait_stage): recorded on the caller's stream when stage i of a direct forward begins (after its
    // upsampling conv), so that ANOTHER stream can start work beside a chosen part of the forward -- the two-stage pipeline across
    // batches runs the next batch's TTE beside the LDS-resident stage 3 / 4 kernels (one workgroup per CU, barrier waits to fill)
    // instead of beside the chip-filling stage 0 / 1 layer convs
    hipEvent_t ev_stage[PARROT_MAX_STAGES] = {};
    bool capturing = false;            // (a graph capture is in progress on cap_stream: no event records into it)
    float* dbg_absmax = nullptr;       // parrot_voc_debug_absmax: (n_stages + 2) device floats, max |conv input| per group (caller-owned)
    // HIP-graph replay of small forwards (PARROT_VOC_GRAPH, default on; B x U <= 8192 units): a forward is ~130 dependent launches
    // of 7-40 us on up to three streams -- at one utterance the fork / join events and the launch gaps are a fifth of it.  A SHAPE
    // (B, U, with / without speaker ids and row lengths) that keeps recurring is captured once (GRAPH_AFTER below; the first calls also warm lazy
    // state), on a stream of the handle and into staging buffers of the handle (ids, lengths, waveform, workspace: one allocation per
    // cached shape, made at capture time -- never per call): PyTorch's allocator hands out different addresses from call to call, and
    // a graph bakes its addresses in.  A replay copies the caller's ids / lengths in (a few KB), launches the graph and copies the
    // waveform out: the same kernels with the same arguments in the same order, the MRF branch streams as graph edges.
    struct Graph {
        int B = 0, U = 0;
        bool has_spkr = false, has_lens = false;
        char* mem = nullptr;  // one allocation: code | spkr | lens | wav | ws
        int64_t *code = nullptr, *spkr = nullptr;
        int32_t* lens = nullptr;
        float* wav = nullptr;
        void* ws = nullptr;
        size_t ws_bytes = 0, wav_bytes = 0, mem_bytes = 0;
        hipGraphExec_t exec = nullptr;
        hipEvent_t done = nullptr;     // recorded behind the last replay's copy-out: a replay on ANOTHER stream waits for it first
        hipStream_t last = nullptr;
        bool launched = false, dead = false;
        int seen = 0;
        unsigned long stamp = 0;
        void release() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (done) (void)hipEventDestroy(done);
            if (mem) (void)hipFree(mem);
            exec = nullptr; done = nullptr; mem = nullptr;
        }
    };
    // A shape is captured at its GRAPH_AFTER-th sighting (capture + instantiate + the staging allocation cost milliseconds: a
    // driver that feeds ever-changing lengths, one utterance per call, must not pay them -- it never does: its shapes do not recur
    // often enough within the MAX_SHAPES most recent ones); at most MAX_GRAPHS shapes hold a graph, least recently used first out
    static constexpr int MAX_GRAPHS = 8, MAX_SHAPES = 64, GRAPH_AFTER = 4;
    // (staging memory of all cached shapes of a handle together: PARROT_VOC_GRAPH_MB, default 1024 -- see voc_forward_graphed)
    std::vector<Graph> graphs;
    std::mutex graph_mu;
    unsigned long graph_clock = 0;
    hipStream_t cap_stream = nullptr;
    ~parrot_voc() {
        for (Graph& g : graphs) g.release();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        for (StreamSet& q : ss) {
            for (hipStream_t st : q.side)
                if (st) (void)hipStreamDestroy(st);
            if (q.ev_fork) (void)hipEventDestroy(q.ev_fork);
            for (hipEvent_t e : q.ev_last)
                if (e) (void)hipEventDestroy(e);
        }
        for (hipStream_t st : lane_stream)
            if (st) (void)hipStreamDestroy(st);
        if (ev_lane_fork) (void)hipEventDestroy(ev_lane_fork);
        for (hipEvent_t e : ev_stage)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_lane_join)
            if (e) (void)hipEventDestroy(e);
        for (uint16_t* q : rb_stream)
            if (q) (void)hipFree(q);
        if (dict) (void)hipFree(dict);
        if (spkr) (void)hipFree(spkr);
        if (err) (void)hipFree(err);
    }
    int chan(int stage) const { return cfg.upsample_initial_channel >> (stage + 1); }
};

static int upload(float** dst, const float* src, size_t n) {
    HIP_TRY(hipMalloc((void**)dst, n * sizeof(float)));
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return PARROT_OK;
}

static int make_conv(std::unique_ptr<parrot_conv>& slot, int cin, int cout, int k, int dil, int pad, int transposed, int stride,
                     int pre, float slope, int act, const float* w, const float* b, int groups = 1, bool allow16 = true) {
    parrot_conv_desc d{};
    d.c_in = cin; d.c_out = cout; d.k = k; d.dilation = dil; d.padding = pad; d.transposed = transposed; d.stride = stride;
    d.pre_act = pre; d.pre_slope = slope; d.act = act; d.tile_cfg = -1; d.precision = -1;
    parrot_conv* c = nullptr;
    TRY(conv_build(&c, &d, groups, w, b, allow16));
    slot.reset(c);
    return PARROT_OK;
}

static int voc_create_impl(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w, int prec, int fused);
extern "C" int parrot_voc_create(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w) {
    return voc_create_impl(out, cfg, w, -1, -1);
}
extern "C" int parrot_voc_create_ex(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w, int32_t precision,
                                    int32_t fused_resblocks) {
    if (precision > PARROT_PREC_F16 || fused_resblocks > 2) return fail(PARROT_E_INVALID, "voc_create_ex: precision in -1 .. 4, fused_resblocks in -1 .. 2");
    return voc_create_impl(out, cfg, w, precision, fused_resblocks);
}
static int voc_create_body(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w);
static int voc_create_impl(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w, int prec, int fused) {
    CreateScope scope(prec, fused, -1);  // (thread-local: the process defaults are not touched)
    return voc_create_body(out, cfg, w);
}
static int voc_create_body(parrot_voc_t** out, const parrot_voc_cfg* cfg, const parrot_voc_weights* w) {
    if (!out || !cfg || !w) return fail(PARROT_E_INVALID, "voc_create: null argument");
    if (cfg->n_stages <= 0 || cfg->n_stages > PARROT_MAX_STAGES || cfg->n_kernels <= 0 || cfg->n_kernels > PARROT_MAX_KERNELS ||
        cfg->n_dil <= 0 || cfg->n_dil > PARROT_MAX_DIL || (cfg->resblock_type != 1 && cfg->resblock_type != 2))
        return fail(PARROT_E_INVALID, "voc_create: bad config");
    const int in_dim = cfg->embedding_dim * (cfg->multispkr ? 2 : 1);
    // model_in_dim - in_dim input channels come from the caller's extra conditioning streams (parrot_voc_forward_feats)
    if (cfg->model_in_dim < in_dim) return fail(PARROT_E_INVALID, "voc_create: model_in_dim smaller than embedding_dim * (1 + multispkr)");
    if ((cfg->upsample_initial_channel >> cfg->n_stages) < 1) return fail(PARROT_E_INVALID, "voc_create: too many stages for upsample_initial_channel");
    const int per_rb = (cfg->resblock_type == 1 ? 2 : 1) * cfg->n_dil;
    if (w->n_rb != cfg->n_stages * cfg->n_kernels * per_rb) return fail(PARROT_E_INVALID, "voc_create: wrong number of resblock convs");
    std::unique_ptr<parrot_voc> v(new parrot_voc());
    v->cfg = *cfg;
    query_device();
    v->scheme = create_prec();
    v->fused = create_fused();
    for (int i = 0; i < cfg->n_stages; ++i) HIP_TRY(hipEventCreateWithFlags(&v->ev_stage[i], hipEventDisableTiming));
    {   // operand planes: default ON for the single-piece schemes (bf16 / f16: -0.4 ms of 11.1 per step, the configs[2] data path),
        // OFF for fp16x3 (bit-identical, but +-0: co-resident workgroups already hide the conversion, profiles/r06h_planes_ab.txt);
        // PARROT_PLANES=0 / 1 forces either
        const char* e = getenv("PARROT_PLANES");
        v->planes = e ? atoi(e) != 0 : (v->scheme == PARROT_PREC_BF16 || v->scheme == PARROT_PREC_F16);
    }
    {
        const char* e = getenv("PARROT_MRF_STREAMS");
        // default "auto": concurrent branches for small batches (B = 1 ... 16: -6 ... -11 % per batch: their launches do not
        // fill the chip), one stream at B x U > 8192 units (measured +-0 at B = 64: power-limited, not tail-limited -- and
        // per-kernel timings stay clean there).  PARROT_MRF_STREAMS=1 / 3 forces either.
        const bool on = e ? atoi(e) > 1 : true;
        v->mrf_auto = (e == nullptr);
        v->mrf_streams = (on && cfg->n_kernels > 1) ? cfg->n_kernels : 1;
        if (v->mrf_streams > 1) {
            for (parrot_voc::StreamSet& q : v->ss) {
                for (int j = 1; j < v->mrf_streams; ++j) HIP_TRY(hipStreamCreateWithFlags(&q.side[j], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&q.ev_fork, hipEventDisableTiming));
                for (int j = 0; j < v->mrf_streams; ++j) HIP_TRY(hipEventCreateWithFlags(&q.ev_last[j], hipEventDisableTiming));
            }
        }
        for (int l = 1; l < parrot_voc::MAX_LANES; ++l) {
            HIP_TRY(hipStreamCreateWithFlags(&v->lane_stream[l], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&v->ev_lane_join[l], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&v->ev_lane_fork, hipEventDisableTiming));
    }
    // ConvTranspose1d(k, stride u, padding (k - u) // 2) (models.py:80-83) yields T u samples for even k - u (every shipped
    // config) and T u + 1 for odd k - u: lengths are taken from the convs' own out_len chain (voc_out_len), ragged rows carry
    // the extra samples as `row_len_add`.  k < u would mean a negative padding, which torch rejects too.
    for (int i = 0; i < cfg->n_stages; ++i) {
        const int u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
        if (u <= 0 || k < u) return fail(PARROT_E_UNSUPPORTED, "voc_create: upsample_kernel_size must be >= upsample_rate (negative padding)");
        if ((k - u) & 1) v->odd_stage = true;
    }
    TRY(upload(&v->dict, w->dict, (size_t)cfg->num_embeddings * cfg->embedding_dim));
    if (cfg->multispkr) {
        if (!w->spkr) return fail(PARROT_E_INVALID, "voc_create: multispkr without spkr table");
        TRY(upload(&v->spkr, w->spkr, (size_t)cfg->n_spkr * cfg->embedding_dim));
    }
    HIP_TRY(hipMalloc((void**)&v->err, sizeof(int)));
    HIP_TRY(hipMemset(v->err, 0, sizeof(int)));
    const int C0 = cfg->upsample_initial_channel;
    TRY(make_conv(v->conv_pre, cfg->model_in_dim, C0, 7, 1, 3, 0, 1, PRE_NONE, 0.f, ACT_NONE, w->conv_pre_w, w->conv_pre_b));
    v->ups.resize(cfg->n_stages);
    v->rb.resize(w->n_rb);
    for (int i = 0; i < cfg->n_stages; ++i) {
        const int cin = C0 >> i, cout = C0 >> (i + 1), u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
        v->up_total *= u;
        TRY(make_conv(v->ups[i], cin, cout, k, 1, (k - u) / 2, 1, u, PRE_LRELU, 0.1f, ACT_NONE, w->ups_w[i], w->ups_b[i]));
        for (int j = 0; j < cfg->n_kernels; ++j) {
            const int rk = cfg->resblock_kernel_sizes[j];
            // (blocks that run on the fused pair kernels reuse their plans' 32x32x16 weight streams: keep that packing)
            const bool a16 = !(v->fused != 0 && cfg->resblock_type == 1 && resblock_split_has(cout, rk) && per_rb <= RBS_MAX_CONVS);
            for (int m = 0; m < cfg->n_dil; ++m) {
                const int dl = cfg->resblock_dilation_sizes[j][m];
                const int base = (i * cfg->n_kernels + j) * per_rb;
                if (cfg->resblock_type == 1) {
                    TRY(make_conv(v->rb[base + 2 * m], cout, cout, rk, dl, (rk * dl - dl) / 2, 0, 1, PRE_LRELU, 0.1f, ACT_NONE,
                                  w->rb_w[base + 2 * m], w->rb_b[base + 2 * m], 1, a16));
                    TRY(make_conv(v->rb[base + 2 * m + 1], cout, cout, rk, 1, (rk - 1) / 2, 0, 1, PRE_LRELU, 0.1f, ACT_NONE,
                                  w->rb_w[base + 2 * m + 1], w->rb_b[base + 2 * m + 1], 1, a16));
                } else {
                    TRY(make_conv(v->rb[base + m], cout, cout, rk, dl, (rk * dl - dl) / 2, 0, 1, PRE_LRELU, 0.1f, ACT_NONE,
                                  w->rb_w[base + m], w->rb_b[base + m], 1, a16));
                }
            }
        }
    }
    // ResBlock1 blocks of the 64-, 32- and 16-channel stages under a split scheme: one concatenated weight stream per
    // block for the fused pair kernels (resblock_split.h), [conv][step][piece][lane][8 x 16 bit] + padding for the prefetch past the end.
    //   32 / 64 channels: the conv plans' own streams ([row tile][chunk * k + tap]);
    //   16 channels: packed here for the 16x16x32 MFMA (step = tap pair; lane = row l&15, channels 8(g&1).., tap 2*step + (g>>1)).
    v->rb_stream.assign((size_t)cfg->n_stages * cfg->n_kernels, nullptr);
    v->rb_wsc.assign((size_t)w->n_rb, 1.f);
    v->rb_conv_halves.assign((size_t)cfg->n_stages * cfg->n_kernels, 0);
    const int NP = scheme_pieces(v->scheme);
    for (int i = 0; i < cfg->n_stages && v->scheme >= 1; ++i)
        for (int j = 0; j < cfg->n_kernels; ++j) {
            const int rk = cfg->resblock_kernel_sizes[j], C = v->chan(i);
            if (cfg->resblock_type != 1 || !resblock_split_has(C, rk) || per_rb > RBS_MAX_CONVS) continue;
            const int base = (i * cfg->n_kernels + j) * per_rb;
            const int steps = resblock_split_steps(C, rk);
            const size_t step_b = (size_t)NP * 1024, conv_b = (size_t)steps * step_b;
            v->rb_conv_halves[(size_t)i * cfg->n_kernels + j] = conv_b / 2;
            if (C >= 32) {  // the plans' streams are already [row tile][chunk * k + tap]
                bool ok = true;
                for (int q = 0; q < per_rb; ++q) {
                    const parrot_conv* pc = v->rb[base + q].get();
                    ok = ok && pc->prec == v->scheme && !pc->mfma16 && pc->wfrag16 && pc->nchunks == C / 16 && (C / 32) * pc->n_it16 == steps && pc->M == C &&
                         C % tile_cfg(pc->cfg).bm == 0;  // (whole M-blocks: the stream is [32-row tile][chunk * k + tap] over all C / 32 tiles)
                }
                if (!ok) continue;
                // (the kernel's in-place prefetch after the last tap reads "the next conv's" tap 0 of both chunks: one
                //  whole conv of padding keeps that inside the allocation)
                uint16_t* st = nullptr;
                HIP_TRY(hipMalloc((void**)&st, (per_rb + 1) * conv_b));
                v->rb_stream[(size_t)i * cfg->n_kernels + j] = st;
                for (int q = 0; q <= per_rb; ++q)
                    HIP_TRY(hipMemcpy(reinterpret_cast<char*>(st) + q * conv_b, v->rb[base + (q < per_rb ? q : 0)]->wfrag16, conv_b, hipMemcpyDeviceToDevice));
                for (int q = 0; q < per_rb; ++q) v->rb_wsc[base + q] = v->rb[base + q]->wscale;
            } else {
                const size_t step_h = (size_t)NP * 512;
                std::vector<uint16_t> pk(((size_t)per_rb * steps + 2) * step_h, 0);
                for (int q = 0; q < per_rb; ++q) {
                    const float* wq = w->rb_w[base + q];  // (16, 16, rk)
                    const float wsc = scheme_is_f16(v->scheme) ? f16_weight_scale(wq, (size_t)16 * 16 * rk) : 1.f;
                    v->rb_wsc[base + q] = wsc;
                    for (int st = 0; st < steps; ++st) {
                        uint16_t* g = pk.data() + ((size_t)q * steps + st) * step_h;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int row = lane & 15, ch = 8 * ((lane >> 4) & 1) + e, tap = 2 * st + (lane >> 5);
                                const float val = tap < rk ? wq[((size_t)row * 16 + ch) * rk + tap] : 0.f;
                                uint16_t h[3];
                                split_weight(val, v->scheme, wsc, h);
                                for (int pc = 0; pc < NP; ++pc) g[pc * 512 + lane * 8 + e] = h[pc];
                            }
                    }
                }
                uint16_t* st = nullptr;
                HIP_TRY(hipMalloc((void**)&st, pk.size() * sizeof(uint16_t)));
                v->rb_stream[(size_t)i * cfg->n_kernels + j] = st;
                HIP_TRY(hipMemcpy(st, pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            }
        }
    // whole-MRF launches for the 32-channel stage (PARROT_MRF_FUSED, default on): all branches of a stage in one kernel, on the pair
    // kernels' own weight streams -- bit-identical to the per-branch launches and no slower (2.27 vs 2.31 ms per step at B = 64; 14 -> 2
    // passes over the stage's activations).  (64 channels were built and measured slower: profiles/r05b_mrf_ab.jsonl.)
    v->mrf_ok.assign((size_t)cfg->n_stages, 0);
    {
        static const bool want = [] { const char* e = getenv("PARROT_MRF_FUSED"); return !e || atoi(e) != 0; }();
        for (int i = 0; i < cfg->n_stages && want && v->fused != 0 && v->scheme >= 1 && resblock_mrf_scheme(v->scheme); ++i) {
            const int C = v->chan(i);
            if (cfg->resblock_type != 1 || !resblock_mrf_has(C) || per_rb > RBS_MAX_CONVS || cfg->n_kernels > RBS_MAX_BRANCH) continue;
            bool ok = true;
            int Hmax = 0;
            for (int j = 0; j < cfg->n_kernels; ++j) {
                const int rk = cfg->resblock_kernel_sizes[j];
                const int base = (i * cfg->n_kernels + j) * per_rb;
                ok = ok && (rk & 1) && v->rb_stream[(size_t)i * cfg->n_kernels + j];
                int H = 0;
                for (int q = 0; q < per_rb; ++q) H += (rk - 1) / 2 * v->rb[base + q]->dil;
                Hmax = std::max(Hmax, H);
            }
            v->mrf_ok[i] = ok && rbs_mrf_window(C) - 2 * Hmax >= rbs_mrf_window(C) / 2;
        }
    }
    // final F.leaky_relu(x) uses the DEFAULT slope 0.01 (models.py:107, quirk Q5)
    TRY(make_conv(v->conv_post, C0 >> cfg->n_stages, 1, 7, 1, 3, 0, 1, PRE_LRELU, 0.01f, ACT_TANH, w->conv_post_w, w->conv_post_b));
    v->conv_post->err_flag = v->err;  // a non-finite waveform sample (an activation left the fp16 split range) raises the handle's flag
    *out = v.release();
    return PARROT_OK;
}
extern "C" void parrot_voc_destroy(parrot_voc_t* v) { delete v; }
extern "C" int parrot_voc_precision(const parrot_voc_t* v) { return v ? v->scheme : PARROT_E_INVALID; }
// Debug aid (range headroom of the fp16 split scheme, |x| < 8190): while dst_dev != NULL every conv launched by
// parrot_voc_forward records max |input element| into dst_dev[group] (atomic max; group 0 = conv_pre, 1 + i = the layers of
// stage i (ups_i and its ResBlocks), n_stages + 1 = conv_post).  Fused ResBlock launches only see their block's input: create the
// handle with fused_resblocks = 0 (parrot_voc_create_ex) to cover every layer.  The caller zeroes the n_stages + 2 floats.
extern "C" int parrot_voc_debug_absmax(parrot_voc_t* v, float* dst_dev) {
    if (!v) return fail(PARROT_E_INVALID, "voc_debug_absmax: null handle");
    v->dbg_absmax = dst_dev;
    return PARROT_OK;
}

static size_t voc_max_act(const parrot_voc* v, int B, int U) {
    size_t mx = (size_t)B * v->cfg.upsample_initial_channel * U;
    int T = U;
    for (int i = 0; i < v->cfg.n_stages; ++i) {
        T = v->ups[i]->out_len(T);
        mx = std::max(mx, (size_t)B * v->chan(i) * (size_t)T);
    }
    return mx;
}
// waveform samples of an utterance of U units: the transposed convs' out_len chain (U * hop unless a stage has odd k - u)
static long voc_out_len(const parrot_voc* v, int U) {
    long T = U;
    for (int i = 0; i < v->cfg.n_stages; ++i) T = v->ups[i]->out_len((int)T);
    return T;
}
extern "C" int64_t parrot_voc_out_len(const parrot_voc_t* v, int32_t U) { return (v && U > 0) ? voc_out_len(v, U) : 0; }

// One fused launch for ResBlock (stage i, kernel j) when the stage is narrow enough to live in LDS.
// Fused whole-ResBlock kernels (resblock_fused.h): mode 0 off, 1 every eligible stage (16 and 32 channels), 2 only the
// 16-channel stages.  Default 2: measured at B=64, the 16-channel kernel (16x16x4 MFMA, 1024-column windows) beats
// the layer-by-layer path (5.3 vs 6.6 ms for stage 4) while the 32-channel one does not yet (10.5 vs 8.5 ms for
// stage 3: 512-column windows pay 12-23 % halo recompute).  PARROT_FUSED / parrot_set_fused_resblocks override.
extern "C" int parrot_set_fused_resblocks(int32_t mode) {
    if (mode < 0 || mode > 2) return fail(PARROT_E_INVALID, "set_fused_resblocks: mode must be 0, 1 or 2");
    g_fused.store(mode);
    return PARROT_OK;
}
static bool resblock_fusable(const parrot_voc* v, int stage, int j) {
    const parrot_voc_cfg& c = v->cfg;
    const int C = v->chan(stage), k = c.resblock_kernel_sizes[j];
    const int fm = v->fused;
    if (fm == 0 || !(C == 16 || (C == 32 && fm == 1)) || !(k & 1)) return false;
    const int per_rb = (c.resblock_type == 1 ? 2 : 1) * c.n_dil;
    if (per_rb > RB_MAX_CONVS) return false;
    int H = 0;
    for (int m = 0; m < c.n_dil; ++m) {
        const int reach = (k - 1) / 2 * c.resblock_dilation_sizes[j][m];
        if (reach > RB_PAD) return false;
        H += reach + (c.resblock_type == 1 ? (k - 1) / 2 : 0);
    }
    const int base = (stage * c.n_kernels + j) * per_rb;
    for (int q = 0; q < per_rb; ++q)
        if (v->rb[base + q]->prec != 0 || v->rb[base + q]->cfg != (C == 16 ? 6 : 2) || !v->rb[base + q]->wfrag) return false;
    return resblock_window(C) - 2 * H >= 128;
}
static int resblock_fused_launch(const parrot_voc* v, int stage, int j, const float* x, float* y, int B, int T, int epi, float div,
                                 hipStream_t s, const int32_t* row_len, int row_len_mul, int row_len_add, hipEvent_t before_last = nullptr) {
    const parrot_voc_cfg& c = v->cfg;
    const int per_rb = (c.resblock_type == 1 ? 2 : 1) * c.n_dil;
    const int base = (stage * c.n_kernels + j) * per_rb;
    ResblockParams p{};
    p.x = x; p.y = y;
    p.n_conv = per_rb; p.type = c.resblock_type;
    p.k = c.resblock_kernel_sizes[j]; p.C = v->chan(stage); p.T = T; p.B = B;
    p.epi = epi; p.div = div; p.slope = 0.1f;
    p.row_len = row_len; p.row_len_mul = row_len_mul; p.row_len_add = row_len_add;
    int H = 0;
    double macs = 0;
    for (int q = 0; q < per_rb; ++q) {
        const parrot_conv* pc = v->rb[base + q].get();
        p.wfrag[q] = pc->wfrag; p.bias[q] = pc->bias; p.dil[q] = pc->dil;
        H += (p.k - 1) / 2 * pc->dil;
        macs += (double)B * p.C * p.C * p.k * T;
    }
    p.H = H;
    p.TT = resblock_window(p.C) - 2 * H;
    p.tiles = (T + p.TT - 1) / p.TT;
    ProfRec rec{};
    if (before_last) HIP_TRY(hipStreamWaitEvent(s, before_last, 0));
    if (g_prof_on) TRY(prof_open(rec, NUM_TILE_CFGS + 2, 2.0 * macs, 4.0 * B * (double)p.C * T * (2 + (epi != EPI_STORE ? 1 : 0)), s));
    HIP_TRY(launch_resblock_fused(p, s));
    if (g_prof_on) TRY(prof_close(rec, s));
    return PARROT_OK;
}

// Split-bf16 fused pair kernel for a 32-channel ResBlock1 block: the pairs are grouped into launches whose total reach
// stays <= hmax columns per side (a 384-column window keeps >= 84 % of its columns at hmax = 30); every launch but
// the last stores its running residual to a scratch buffer.
constexpr int RBS_HMAX = 30;
static int resblock_split_launch(const parrot_voc* v, int stage, int j, const float* x, float* y, float* tmp_a, float* tmp_b, int B, int T,
                                 int epi, float div, hipStream_t s, const int32_t* row_len, int row_len_mul, int row_len_add,
                                 hipEvent_t before_last = nullptr) {
    const parrot_voc_cfg& c = v->cfg;
    const int per_rb = 2 * c.n_dil, k = c.resblock_kernel_sizes[j], C = v->chan(stage);
    const int base = (stage * c.n_kernels + j) * per_rb;
    const int W = resblock_split_window(C), steps = resblock_split_steps(C, k);
    const int hmax = RBS_HMAX * W / RBS_W;  // the same fraction of the window
    const uint16_t* stream = v->rb_stream[(size_t)stage * c.n_kernels + j];
    const float* src = x;
    int m0 = 0, n_launch = 0;
    while (m0 < per_rb) {
        int m1 = m0, H = 0;
        while (m1 < per_rb) {
            const int h2 = (k - 1) / 2 * (v->rb[base + m1]->dil + v->rb[base + m1 + 1]->dil);
            if (m1 > m0 && H + h2 > hmax) break;
            H += h2;
            m1 += 2;
        }
        const bool last = (m1 == per_rb);
        ResblockSplitParams p{};
        p.x = src;
        p.y = last ? y : ((n_launch & 1) ? tmp_b : tmp_a);
        p.wstream = stream + (size_t)m0 * v->rb_conv_halves[(size_t)stage * c.n_kernels + j];
        p.n_conv = m1 - m0;
        p.early = 1;
        for (int q = m0; q < m1; ++q) {
            p.bias[q - m0] = v->rb[base + q]->bias;
            p.wsc[q - m0] = v->rb_wsc[base + q];
            p.dil[q - m0] = v->rb[base + q]->dil;
            if ((k - 1) / 2 * v->rb[base + q]->dil > 32) p.early = 0;
        }
        p.T = T; p.B = B; p.H = H; p.k = k;
        p.TT = W - 2 * H;
        if (p.TT < 32) return fail(PARROT_E_UNSUPPORTED, "resblock: receptive field too wide for the fused window");
        p.tiles = (T + p.TT - 1) / p.TT;
        p.epi = last ? epi : EPI_STORE;
        p.div = div; p.slope = 0.1f;
        p.row_len = row_len; p.row_len_mul = row_len_mul; p.row_len_add = row_len_add;
        ProfRec rec{};
        const double macs = (double)B * C * C * k * T * (m1 - m0);
        if (last && before_last) HIP_TRY(hipStreamWaitEvent(s, before_last, 0));  // the MRF sum is accumulated in branch order
        // rows: one per kernel instantiation (C = 32 / 64 / 128 / 256 are resblock_split_kernel<SCH, 2 / 4 / 8 / 16>)
        const int prow = C == 16 ? 6 : C == 32 ? 5 : C == 64 ? 13 : C == 128 ? 14 : 15;
        if (g_prof_on) TRY(prof_open(rec, NUM_TILE_CFGS + prow, 2.0 * macs, 4.0 * B * (double)C * T * (2 + (p.epi != EPI_STORE ? 1 : 0)), s));
        HIP_TRY(launch_resblock_split(v->scheme, C, p, s));
        if (g_prof_on) TRY(prof_close(rec, s));
        src = p.y;
        m0 = m1;
        ++n_launch;
    }
    return PARROT_OK;
}

// output columns per window of a whole-MRF launch of this stage (window minus twice the widest branch's total reach)
static int mrf_tile_cols(const parrot_voc* v, int stage) {
    const parrot_voc_cfg& c = v->cfg;
    const int per_rb = 2 * c.n_dil;
    int Hmax = 0;
    for (int j = 0; j < c.n_kernels; ++j) {
        int H = 0;
        for (int q = 0; q < per_rb; ++q) H += (c.resblock_kernel_sizes[j] - 1) / 2 * v->rb[(stage * c.n_kernels + j) * per_rb + q]->dil;
        Hmax = std::max(Hmax, H);
    }
    return std::max(1, rbs_mrf_window(v->chan(stage)) - 2 * Hmax);
}
// Whole-MRF launch of the 32-channel stage (resblock_split.h, MRF instantiations): y = sum_j ResBlock_j(x) / n_kernels.
static int mrf_split_launch(const parrot_voc* v, int stage, const float* x, float* y, int B, int T, hipStream_t s, const int32_t* row_len,
                            int row_len_mul, int row_len_add) {
    const parrot_voc_cfg& c = v->cfg;
    const int per_rb = 2 * c.n_dil, C = v->chan(stage), nk = c.n_kernels;
    const int W = rbs_mrf_window(C);
    ResblockSplitParams p{};
    p.x = x; p.y = y;
    p.n_conv = per_rb; p.n_branch = nk;
    int Hmax = 0, early = 1;
    double macs = 0;
    for (int j = 0; j < nk; ++j) {
        const int base = (stage * nk + j) * per_rb, k = c.resblock_kernel_sizes[j];
        p.bstream[j] = v->rb_stream[(size_t)stage * nk + j];
        p.bk[j] = k;
        int H = 0;
        for (int q = 0; q < per_rb; ++q) {
            p.bias[j * per_rb + q] = v->rb[base + q]->bias;
            p.wsc[j * per_rb + q] = v->rb_wsc[base + q];
            p.dil[j * per_rb + q] = v->rb[base + q]->dil;
            H += (k - 1) / 2 * v->rb[base + q]->dil;
            if ((k - 1) / 2 * v->rb[base + q]->dil > 32) early = 0;
        }
        Hmax = std::max(Hmax, H);
        macs += (double)B * C * C * k * T * per_rb;
    }
    p.k = p.bk[0];
    p.early = early;
    p.T = T; p.B = B; p.H = Hmax;
    p.TT = W - 2 * Hmax;
    p.tiles = (T + p.TT - 1) / p.TT;
    p.epi = nk > 1 ? EPI_ADD_DIV : EPI_STORE;
    p.div = (float)nk; p.slope = 0.1f;
    p.row_len = row_len; p.row_len_mul = row_len_mul; p.row_len_add = row_len_add;
    ProfRec rec{};
    if (g_prof_on) TRY(prof_open(rec, NUM_TILE_CFGS + 16, 2.0 * macs, 4.0 * B * (double)C * T * 2, s));
    HIP_TRY(launch_mrf_split(v->scheme, C, p, s));
    if (g_prof_on) TRY(prof_close(rec, s));
    return PARROT_OK;
}

// MRF branches that run concurrently for this shape: the handle's stream count, or one above B x U = 8192 units in auto mode
// (one rule for the workspace size and for the forward pass)
static int voc_streams(const parrot_voc* v, int B, int U) { return (v->mrf_auto && (long)B * U > 8192) ? 1 : v->mrf_streams; }

static size_t voc_ws_bytes(const parrot_voc* v, int B, int U, int ns) {
    Arena a(nullptr, 0);
    a.take<float>((size_t)B * v->cfg.model_in_dim * U);
    const size_t mx = voc_max_act(v, B, U);
    for (int i = 0; i < 3 + 3 * ns; ++i) a.take<float>(mx);  // stage in / ups out / MRF sum + (T1, RA, RB) per concurrent branch
    return align_up(a.off, 256);
}
extern "C" size_t parrot_voc_workspace_bytes(const parrot_voc_t* v, int32_t B, int32_t U) {
    if (!v || B <= 0 || U <= 0) return 0;
    return voc_ws_bytes(v, B, U, voc_streams(v, B, U));
}

// Receptive field of the generator in units, either side of an output frame: the interval [tau, tau] of one waveform sample is
// propagated back through conv_post, every MRF stage (the widest ResBlock), every ConvTranspose1d (tau = t u - p + kappa) and
// conv_pre, for every phase tau mod hop; a chunk computed with this much real context equals the whole-utterance forward.
static int voc_receptive_units(const parrot_voc* v) {
    const parrot_voc_cfg& c = v->cfg;
    const long hop = v->up_total, base = 4096;  // far from the origin: integer divisions below see positive numbers only
    long worst = 0;
    for (long ph = 0; ph < hop; ++ph) {
        long lo = base * hop + ph, hi = lo;
        lo -= 3; hi += 3;  // conv_post (k = 7)
        for (int i = c.n_stages - 1; i >= 0; --i) {
            long reach = 0;  // widest ResBlock of the stage: sum over its convs of (k - 1) / 2 * dilation
            for (int j = 0; j < c.n_kernels; ++j) {
                const long hk = (c.resblock_kernel_sizes[j] - 1) / 2;
                long r = 0;
                for (int m = 0; m < c.n_dil; ++m) r += hk * c.resblock_dilation_sizes[j][m] + (c.resblock_type == 1 ? hk : 0);
                reach = std::max(reach, r);
            }
            lo -= reach; hi += reach;
            const long u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i], p = (k - u) / 2;
            // output tau depends on inputs t with tau = t u - p + kappa, kappa in [0, k): t in [ceil((tau + p - k + 1) / u), floor((tau + p) / u)]
            lo = (lo + p - k + 1 + u - 1) / u;
            hi = (hi + p) / u;
        }
        lo -= 3; hi += 3;  // conv_pre (k = 7)
        worst = std::max(worst, std::max(base - lo, hi - base));
    }
    return (int)worst;
}
extern "C" int parrot_voc_receptive_units(const parrot_voc_t* v) { return v ? voc_receptive_units(v) : PARROT_E_INVALID; }

static int voc_forward_impl(parrot_voc_t* v, const int64_t* code, int code_stride, const int64_t* spkr, const float* feats,
                            int32_t n_feat_channels, const int32_t* unit_lens, int32_t B, int32_t U, float* wav_out,
                            float* const* stage_out, void* ws, size_t ws_bytes, void* stream, int ns_sized = 0, int lane = 0);

// Small forwards through the graph cache (see parrot_voc::Graph); everything else -- and every failure of the graph path -- direct.
static int voc_forward_graphed(parrot_voc_t* v, const int64_t* code, int code_stride, const int64_t* spkr, const float* feats,
                               int32_t n_feat, const int32_t* unit_lens, int32_t B, int32_t U, float* wav_out, float* const* stage_out,
                               void* ws, size_t ws_bytes, void* stream) {
    static const bool want = [] { const char* e = getenv("PARROT_VOC_GRAPH"); return !e || atoi(e) != 0; }();
    hipStream_t s = (hipStream_t)stream;
    auto direct = [&]() {
        if (poison_word() && v && B > 0 && U > 0) {
            TRY(poison(ws, ws_bytes, s));
            if (wav_out) TRY(poison(wav_out, (size_t)B * (size_t)voc_out_len(v, U) * sizeof(float), s));
        }
        return voc_forward_impl(v, code, code_stride, spkr, feats, n_feat, unit_lens, B, U, wav_out, stage_out, ws, ws_bytes, stream);
    };
    const bool small = v && B > 0 && U > 0 && (long)B * U <= 8192;
    if (!want || !small || stage_out || v->dbg_absmax || g_prof_on || !code || !wav_out || !ws || n_feat != 0 || feats ||
        (v->cfg.multispkr && !spkr) || v->cfg.model_in_dim != v->cfg.embedding_dim * (v->cfg.multispkr ? 2 : 1))
        return direct();  // (argument errors are reported by the direct path)
    {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return direct();  // the caller is capturing: stay inside ITS graph
    }
    // the caller's workspace must be what a direct run needs (a replay does not touch it, but the contract is the same call)
    if (ws_bytes < voc_ws_bytes(v, B, U, voc_streams(v, B, U))) return direct();
    const bool has_spkr = v->cfg.multispkr && spkr != nullptr, has_lens = unit_lens != nullptr;
    std::unique_lock<std::mutex> lk(v->graph_mu);
    parrot_voc::Graph* g = nullptr;
    for (parrot_voc::Graph& q : v->graphs)
        if (q.B == B && q.U == U && q.has_spkr == has_spkr && q.has_lens == has_lens) {
            g = &q;
            break;
        }
    auto drop = [&](size_t i) {  // (a graph may still be running: its buffers are freed only after the device has drained what uses them)
        if (v->graphs[i].launched && v->graphs[i].done) (void)hipEventSynchronize(v->graphs[i].done);
        v->graphs[i].release();
    };
    if (!g) {  // first sighting: count the shape (the least recently seen shape makes room), run directly
        if ((int)v->graphs.size() >= parrot_voc::MAX_SHAPES) {
            size_t old = 0;
            for (size_t i = 1; i < v->graphs.size(); ++i)
                if (v->graphs[i].stamp < v->graphs[old].stamp) old = i;
            drop(old);
            v->graphs.erase(v->graphs.begin() + old);
        }
        parrot_voc::Graph q{};
        q.B = B; q.U = U; q.has_spkr = has_spkr; q.has_lens = has_lens;
        q.seen = 1;
        q.stamp = ++v->graph_clock;
        v->graphs.push_back(q);
        lk.unlock();
        return direct();
    }
    g->stamp = ++v->graph_clock;
    if (g->dead || (!g->exec && ++g->seen < parrot_voc::GRAPH_AFTER)) {
        lk.unlock();
        return direct();
    }
    if (!g->exec) {  // this shape keeps coming: staging buffers + capture on the handle's own stream (nothing runs during the capture)
        const size_t n_code = (size_t)B * U * sizeof(int64_t), n_spkr = has_spkr ? (size_t)B * sizeof(int64_t) : 0,
                     n_lens = has_lens ? (size_t)B * sizeof(int32_t) : 0;
        g->wav_bytes = (size_t)B * (size_t)voc_out_len(v, U) * sizeof(float);
        g->ws_bytes = voc_ws_bytes(v, B, U, voc_streams(v, B, U));
        size_t off = 0;
        auto take = [&](size_t n) { const size_t o = off; off = align_up(off + n, 256); return o; };
        const size_t o_code = take(n_code), o_spkr = take(n_spkr), o_lens = take(n_lens), o_wav = take(g->wav_bytes), o_ws = take(g->ws_bytes);
        // it takes a graph slot and `off` bytes of staging memory: least recently used graphs give theirs up until both fit
        // (PARROT_VOC_GRAPH_MB caps the staging memory of one handle, default 1024 -- it is hipMalloc'ed, outside torch's allocator)
        static const size_t cap = [] { const char* e = getenv("PARROT_VOC_GRAPH_MB"); return (size_t)(e ? std::max(0, atoi(e)) : 1024) << 20; }();
        for (;;) {
            int n_graphs = 0;
            size_t in_use = 0, lru = v->graphs.size();
            for (size_t i = 0; i < v->graphs.size(); ++i) {
                in_use += v->graphs[i].mem_bytes;
                if (v->graphs[i].exec) {
                    ++n_graphs;
                    if (lru == v->graphs.size() || v->graphs[i].stamp < v->graphs[lru].stamp) lru = i;
                }
            }
            if ((n_graphs < parrot_voc::MAX_GRAPHS && in_use + off <= cap) || lru == v->graphs.size()) break;
            drop(lru);
            v->graphs[lru].mem_bytes = 0;
            v->graphs[lru].seen = 0;
            v->graphs[lru].launched = false;
        }
        size_t in_use = 0;
        for (const parrot_voc::Graph& q : v->graphs) in_use += q.mem_bytes;
        if (in_use + off > cap) {  // larger than the whole cap: direct, and not asked again for a while
            g->seen = -1000;
            lk.unlock();
            return direct();
        }
        bool ok = hipMalloc((void**)&g->mem, off) == hipSuccess;
        if (ok) {
            g->mem_bytes = off;
            g->code = reinterpret_cast<int64_t*>(g->mem + o_code);
            g->spkr = has_spkr ? reinterpret_cast<int64_t*>(g->mem + o_spkr) : nullptr;
            g->lens = has_lens ? reinterpret_cast<int32_t*>(g->mem + o_lens) : nullptr;
            g->wav = reinterpret_cast<float*>(g->mem + o_wav);
            g->ws = g->mem + o_ws;
            ok = hipEventCreateWithFlags(&g->done, hipEventDisableTiming) == hipSuccess;
        }
        if (ok && !v->cap_stream) ok = hipStreamCreateWithFlags(&v->cap_stream, hipStreamNonBlocking) == hipSuccess;
        hipGraph_t graph = nullptr;
        if (ok) ok = hipStreamBeginCapture(v->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            v->capturing = true;
            const int rc = voc_forward_impl(v, g->code, U, g->spkr, nullptr, 0, g->lens, B, U, g->wav, nullptr, g->ws, g->ws_bytes, (void*)v->cap_stream);
            v->capturing = false;
            const hipError_t e_end = hipStreamEndCapture(v->cap_stream, &graph);
            ok = rc == PARROT_OK && e_end == hipSuccess && graph && hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        }
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {  // this shape goes direct from now on
            (void)hipGetLastError();
            g->release();
            g->mem_bytes = 0;
            g->dead = true;
            lk.unlock();
            return direct();
        }
    }
    if (g->launched && g->last != s) HIP_TRY(hipStreamWaitEvent(s, g->done, 0));  // the staging buffers are still the previous replay's
    HIP_TRY(hipMemcpy2DAsync(g->code, (size_t)U * sizeof(int64_t), code, (size_t)code_stride * sizeof(int64_t), (size_t)U * sizeof(int64_t), B,
                             hipMemcpyDeviceToDevice, s));
    if (g->spkr) HIP_TRY(hipMemcpyAsync(g->spkr, spkr, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (g->lens) HIP_TRY(hipMemcpyAsync(g->lens, unit_lens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (poison_word()) {
        TRY(poison(g->ws, g->ws_bytes, s));
        TRY(poison(g->wav, g->wav_bytes, s));
        TRY(poison(wav_out, g->wav_bytes, s));
        TRY(poison(ws, ws_bytes, s));
    }
    HIP_TRY(hipGraphLaunch(g->exec, s));
    HIP_TRY(hipMemcpyAsync(wav_out, g->wav, g->wav_bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipEventRecord(g->done, s));
    g->last = s;
    g->launched = true;
    return PARROT_OK;
}

// Make `stream` wait until the most recently ENQUEUED direct forward of this handle has reached stage `stage` (its upsampling conv
// is done, its MRF begins; the events are created with the handle and recorded by every direct forward).  Before the first forward,
// and for graph replays (small shapes), there is nothing to wait for -- a caller that overlaps work with the forward loses the
// delay, never correctness.
extern "C" int parrot_voc_wait_stage(parrot_voc_t* v, int32_t stage, void* stream) {
    if (!v || stage < 0 || stage >= v->cfg.n_stages) return fail(PARROT_E_INVALID, "voc_wait_stage: stage out of range");
    if (!v->ev_stage[stage]) return PARROT_OK;
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, v->ev_stage[stage], 0));  // (never recorded yet: no wait)
    return PARROT_OK;
}

extern "C" int parrot_voc_forward(parrot_voc_t* v, const int64_t* code, const int64_t* spkr, const int32_t* unit_lens, int32_t B,
                                  int32_t U, float* wav_out, float* const* stage_out, void* ws, size_t ws_bytes, void* stream) {
    return voc_forward_graphed(v, code, U, spkr, nullptr, 0, unit_lens, B, U, wav_out, stage_out, ws, ws_bytes, stream);
}

extern "C" int parrot_voc_forward_feats(parrot_voc_t* v, const int64_t* code, const int64_t* spkr, const float* feats,
                                        int32_t n_feat_channels, const int32_t* unit_lens, int32_t B, int32_t U, float* wav_out,
                                        float* const* stage_out, void* ws, size_t ws_bytes, void* stream) {
    return voc_forward_graphed(v, code, U, spkr, feats, n_feat_channels, unit_lens, B, U, wav_out, stage_out, ws, ws_bytes, stream);
}

// Chunk-streamed synthesis inside the library (`chunk_units`; BASELINE configs[4]): consecutive chunks of
// `chunk_units` units are vocoded with `halo_units` (< 0: the generator's receptive field, computed from the config: 21 units
// for the shipped one) of real context on both sides and only their own samples are copied into wav_out -- equal to the
// whole-utterance forward, with the activation memory of chunk_units + 2 halo_units units.  Chunks that touch a true sequence
// edge contain the edge.
// Chunk lanes: consecutive chunks are independent (each is vocoded with its own halo), so several are kept in flight -- chunk c on
// lane c % n, each lane with its own stream, scratch and workspace, and WITHOUT the MRF branch streams: one 256-unit chunk of a
// B = 8 batch is a ninth of the BASELINE batch, its launches leave CUs idle that the other lanes' launches fill, and whole chunks
// overlap better than the three branches of one (B = 8 x 1500 units in 256-unit chunks: 15.9 ms one chunk at a time with branch
// streams, 15.5 with two lanes of three branch streams, 13.4 with two plain lanes; whole utterance 12.0).  PARROT_CHUNK_LANES=1..4.
static int chunk_lanes() {
    static const int n = [] { const char* e = getenv("PARROT_CHUNK_LANES"); const int q = e ? atoi(e) : 2; return std::min(std::max(q, 1), (int)parrot_voc::MAX_LANES); }();
    return n;
}
extern "C" size_t parrot_voc_chunked_workspace_bytes(const parrot_voc_t* v, int32_t B, int32_t chunk_units, int32_t halo_units) {
    if (!v || B <= 0 || chunk_units <= 0) return 0;
    const int halo = halo_units < 0 ? voc_receptive_units(v) : halo_units;
    const int span = chunk_units + 2 * halo;
    Arena a(nullptr, 0);
    for (int lane = 0; lane < chunk_lanes(); ++lane) {
        a.take<float>((size_t)B * span * v->up_total);
        a.take<int32_t>((size_t)B);
        a.off = align_up(a.off, 256) + voc_ws_bytes(v, B, span, 1);
    }
    // a call that ends up on ONE lane (a single chunk, PARROT_CHUNK_LANES=1) runs its chunks with the shape's MRF branch streams
    Arena one(nullptr, 0);
    one.take<float>((size_t)B * span * v->up_total);
    one.take<int32_t>((size_t)B);
    one.off = align_up(one.off, 256) + voc_ws_bytes(v, B, span, voc_streams(v, B, span));
    return align_up(std::max(a.off, one.off), 256);
}
extern "C" int parrot_voc_forward_chunked(parrot_voc_t* v, const int64_t* code, const int64_t* spkr, const int32_t* unit_lens, int32_t B,
                                          int32_t U, int32_t chunk_units, int32_t halo_units, float* wav_out, void* ws, size_t ws_bytes,
                                          void* stream) {
    if (!v || !code || !wav_out || !ws) return fail(PARROT_E_INVALID, "voc_forward_chunked: null argument");
    if (B <= 0 || U <= 0 || chunk_units <= 0) return fail(PARROT_E_INVALID, "voc_forward_chunked: empty batch or chunk");
    if (v->cfg.model_in_dim != v->cfg.embedding_dim * (v->cfg.multispkr ? 2 : 1))
        return fail(PARROT_E_UNSUPPORTED, "voc_forward_chunked: models with extra conditioning streams go through parrot_voc_forward_feats");
    if (v->odd_stage)
        return fail(PARROT_E_UNSUPPORTED, "voc_forward_chunked: a stage with odd upsample_kernel_size - upsample_rate has no constant samples-per-unit hop to cut chunks on");
    const int halo = halo_units < 0 ? voc_receptive_units(v) : halo_units;
    const int span = chunk_units + 2 * halo, hop = v->up_total;
    hipStream_t s = (hipStream_t)stream;
    const int n_chunks = (U + chunk_units - 1) / chunk_units;
    const int n_lanes = std::min(chunk_lanes(), n_chunks);
    if (poison_word()) {
        TRY(poison(ws, ws_bytes, s));
        TRY(poison(wav_out, (size_t)B * U * hop * sizeof(float), s));
    }
    // lanes run WITHOUT the MRF branch streams (whole chunks overlap better than the branches of one); a call on one lane only -- a
    // single chunk, PARROT_CHUNK_LANES=1 -- keeps the branch streams of its shape, like parrot_voc_forward on that span
    const int ns_chunk = n_lanes > 1 ? 1 : voc_streams(v, B, span);
    Arena a(ws, ws_bytes);
    const size_t inner = voc_ws_bytes(v, B, span, ns_chunk);
    float* tmp[parrot_voc::MAX_LANES] = {};
    int32_t* lens[parrot_voc::MAX_LANES] = {};
    void* inner_ws[parrot_voc::MAX_LANES] = {};
    for (int lane = 0; lane < n_lanes; ++lane) {
        tmp[lane] = a.take<float>((size_t)B * span * hop);
        lens[lane] = a.take<int32_t>((size_t)B);
        a.off = align_up(a.off, 256);
        inner_ws[lane] = (char*)ws + a.off;
        a.off += inner;
    }
    if (!a.ok || a.off > ws_bytes) return fail(PARROT_E_NOMEM, "voc_forward_chunked: workspace too small");
    std::unique_lock<std::mutex> side_lock(v->side_mu, std::defer_lock);  // the lane streams and events are the handle's
    if (n_lanes > 1) side_lock.lock();                                     // (one lane: voc_forward_impl takes it for its branch streams)
    // Whatever happens below, the caller's stream must not run past work that the lanes still have queued on ws / wav_out (the
    // caller may free them right after an error return): the join runs on every exit once the lanes have been forked.
    struct LaneJoin {
        parrot_voc* v; hipStream_t s; int n; bool armed;
        ~LaneJoin() {
            if (!armed) return;
            for (int l = 1; l < n; ++l)
                if (hipEventRecord(v->ev_lane_join[l], v->lane_stream[l]) == hipSuccess) (void)hipStreamWaitEvent(s, v->ev_lane_join[l], 0);
        }
    } join{v, s, n_lanes, false};
    if (n_lanes > 1) {  // the other lanes see what the caller's stream has produced so far (code, spkr, unit_lens)
        HIP_TRY(hipEventRecord(v->ev_lane_fork, s));
        join.armed = true;
        for (int l = 1; l < n_lanes; ++l) HIP_TRY(hipStreamWaitEvent(v->lane_stream[l], v->ev_lane_fork, 0));
    }
    int c = 0;
    for (int start = 0; start < U; start += chunk_units, ++c) {
        const int lane = c % n_lanes;
        hipStream_t sl = lane ? v->lane_stream[lane] : s;
        const int stop = std::min(U, start + chunk_units);
        const int lo = std::max(0, start - halo), hi = std::min(U, stop + halo), n = hi - lo;
        if (unit_lens) {
            hipLaunchKernelGGL(rebase_lens_kernel, dim3((B + 255) / 256), dim3(256), 0, sl, unit_lens, lens[lane], B, lo, n);
            HIP_TRY(hipGetLastError());
        }
        // (the branch-stream count the lane's workspace was sized for)
        TRY(voc_forward_impl(v, code + lo, U, spkr, nullptr, 0, unit_lens ? lens[lane] : nullptr, B, n, tmp[lane], nullptr, inner_ws[lane], inner,
                             (void*)sl, ns_chunk, lane));
        HIP_TRY(hipMemcpy2DAsync(wav_out + (size_t)start * hop, (size_t)U * hop * sizeof(float), tmp[lane] + (size_t)(start - lo) * hop,
                                 (size_t)n * hop * sizeof(float), (size_t)(stop - start) * hop * sizeof(float), B, hipMemcpyDeviceToDevice, sl));
    }
    // (join: LaneJoin's destructor -- the caller's stream continues when every lane has written its chunks)
    return PARROT_OK;
}

static int voc_forward_impl(parrot_voc_t* v, const int64_t* code, int code_stride, const int64_t* spkr, const float* feats,
                            int32_t n_feat_channels, const int32_t* unit_lens, int32_t B, int32_t U, float* wav_out,
                            float* const* stage_out, void* ws, size_t ws_bytes, void* stream, int ns_sized, int lane) {
    if (!v || !code || !wav_out || !ws) return fail(PARROT_E_INVALID, "voc_forward: null argument");
    {
        const int base = v->cfg.embedding_dim * (v->cfg.multispkr ? 2 : 1);
        if (n_feat_channels != v->cfg.model_in_dim - base || (n_feat_channels > 0 && !feats))
            return fail(PARROT_E_INVALID, "voc_forward: extra feature channels must fill model_in_dim - embedding_dim * (1 + multispkr)");
    }
    if (B <= 0 || U <= 0) return fail(PARROT_E_INVALID, "voc_forward: empty batch");
    if (v->cfg.multispkr && !spkr) return fail(PARROT_E_INVALID, "voc_forward: multispkr model needs spkr ids");
    hipStream_t s = (hipStream_t)stream;
    const parrot_voc_cfg& c = v->cfg;
    Arena a(ws, ws_bytes);
    float* x0 = a.take<float>((size_t)B * c.model_in_dim * U);
    const size_t mx = voc_max_act(v, B, U);
    // concurrent MRF branches: the shape's own rule, or -- chunked path -- the count the caller sized the workspace with
    const int ns = ns_sized > 0 ? std::min(ns_sized, v->mrf_streams) : voc_streams(v, B, U);
    const int ns_alloc = ns;  // (parrot_voc_workspace_bytes reserves exactly these)
    // (chunk lanes run without branch streams: `lane` only documents the caller)
    (void)lane;
    const parrot_voc::StreamSet& ss = v->ss[0];
    std::unique_lock<std::mutex> side_lock(v->side_mu, std::defer_lock);
    if (ns > 1) side_lock.lock();
    float* P[3];
    for (int i = 0; i < 3; ++i) P[i] = a.take<float>(mx);
    float* TMP[PARROT_MAX_KERNELS][3];  // (T1, RA, RB) per concurrent branch
    for (int j = 0; j < ns_alloc; ++j)
        for (int q = 0; q < 3; ++q) TMP[j][q] = a.take<float>(mx);
    if (!a.ok) return fail(PARROT_E_NOMEM, "voc_forward: workspace too small");

    {
        const int base = c.embedding_dim * (c.multispkr ? 2 : 1);
        dim3 grid((U + 63) / 64, (base + 63) / 64, B);
        hipLaunchKernelGGL(voc_embed_kernel, grid, dim3(256), 0, s, code, spkr, v->dict, v->spkr, x0, U, c.embedding_dim,
                           base, c.model_in_dim, c.num_embeddings, c.n_spkr, v->err, code_stride);
        HIP_TRY(hipGetLastError());
        if (n_feat_channels > 0)  // extra conditioning streams (already upsampled to U frames) behind the embeddings
            HIP_TRY(hipMemcpy2DAsync(x0 + (size_t)base * U, (size_t)c.model_in_dim * U * sizeof(float), feats,
                                     (size_t)n_feat_channels * U * sizeof(float), (size_t)n_feat_channels * U * sizeof(float), B,
                                     hipMemcpyDeviceToDevice, s));
    }
    auto snap = [&](int idx, const float* src, size_t n) -> int {
        if (stage_out && stage_out[idx]) HIP_TRY(hipMemcpyAsync(stage_out[idx], src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return PARROT_OK;
    };
    auto amax = [&](int group, const float* src, size_t n, hipStream_t q) -> int {  // debug: max |conv input| per layer group
        if (v->dbg_absmax) {
            hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, q, src, n, v->dbg_absmax + group);
            HIP_TRY(hipGetLastError());
        }
        return PARROT_OK;
    };
    int ia = 0;
    // unit_lens (optional): per-row number of real units; every layer then applies ITS zero padding at the row's own
    // end (row_len * samples-per-unit so far), so a padded batch row equals the reference's B=1 run of that utterance
    int mul = 1, add = 0;  // a row of n units holds n * mul + add samples at the current layer
    TRY(amax(0, x0, (size_t)B * c.model_in_dim * U, s));
    TRY(conv_launch(v->conv_pre.get(), x0, nullptr, P[ia], B, U, EPI_STORE, 1.f, 0, 0, 0, s, unit_lens, mul, add));
    TRY(snap(0, P[ia], (size_t)B * c.upsample_initial_channel * U));
    int T = U;
    const int nk = c.n_kernels, nd = c.n_dil;
    const int per_rb = (c.resblock_type == 1 ? 2 : 1) * nd;
    for (int i = 0; i < c.n_stages; ++i) {
        float* A = P[ia];
        float* X = P[(ia + 1) % 3];
        float* XS = P[(ia + 2) % 3];
        TRY(amax(1 + i, A, (size_t)B * (c.upsample_initial_channel >> i) * T, s));
        TRY(conv_launch(v->ups[i].get(), A, nullptr, X, B, T, EPI_STORE, 1.f, 0, 0, 0, s, unit_lens, mul, add));
        T = v->ups[i]->out_len(T);
        mul *= c.upsample_rates[i];
        add = add * c.upsample_rates[i] + ((c.upsample_kernel_sizes[i] - c.upsample_rates[i]) & 1);  // out_len(n mul + add)
        const size_t n_act = (size_t)B * v->chan(i) * T;
        if (v->ev_stage[i] && !v->capturing) HIP_TRY(hipEventRecord(v->ev_stage[i], s));
        TRY(snap(1 + 2 * i, X, n_act));
        // every branch of the stage in ONE launch -- when that launch fills the chip twice over (one 512-thread workgroup per CU
        // walks 18 convs: a few windows are faster as per-branch launches on the branch streams; same bits either way at 32 channels)
        if (v->mrf_ok[i] && (double)v->chan(i) * T * 4.0 < 2147483648.0 &&
            (long)B * ((T + mrf_tile_cols(v, i) - 1) / mrf_tile_cols(v, i)) >= 2L * g_num_cus) {
            TRY(amax(1 + i, X, n_act, s));
            TRY(mrf_split_launch(v, i, X, XS, B, T, s, unit_lens, mul, add));
            TRY(snap(2 + 2 * i, XS, n_act));
            ia = (ia + 2) % 3;
            continue;
        }
        if (ns > 1) {  // fork: the side streams see the upsampled stage input
            HIP_TRY(hipEventRecord(ss.ev_fork, s));
            for (int j = 1; j < ns; ++j) HIP_TRY(hipStreamWaitEvent(ss.side[j], ss.ev_fork, 0));
        }
        for (int j = 0; j < nk; ++j) {
            // the longest branch (largest kernel size = last) stays on the caller's stream
            const int slot = (ns > 1) ? (j + 1) % nk : 0;
            hipStream_t sj = (slot == 0) ? s : ss.side[slot];
            float* T1 = TMP[slot][0];
            float* RA = TMP[slot][1];
            float* RB = TMP[slot][2];
            hipEvent_t order = (ns > 1 && j > 0) ? ss.ev_last[j - 1] : nullptr;  // XS accumulates in branch order (models.py:100-106)
            const float* r = X;
            const int base = (i * nk + j) * per_rb;
            const int epi_last = (nk == 1 || j == 0) ? EPI_STORE : (j == nk - 1 ? EPI_ADD_DIV : EPI_ADD);
            // (the fused kernels address a batch row with 32-bit byte offsets: rows of 2 GiB and more go layer by layer)
            // (which kernel family runs a block never depends on the batch size or on the launch size: row b of a batch must equal
            //  the same utterance run alone BIT FOR BIT -- the batched driver's byte-identical WAVs rest on it.  At B = 1 the
            //  96-column windows of the 128- / 256-channel pair kernels are only 16-61 workgroups: 3.4 instead of 3.2 ms per utterance)
            if (j == 0) TRY(amax(1 + i, X, n_act, s));  // the stage input feeds the first conv of every branch
            if (v->fused != 0 && v->rb_stream[(size_t)i * nk + j] && (double)v->chan(i) * T * 4.0 < 2147483648.0) {
                TRY(resblock_split_launch(v, i, j, X, XS, RA, RB, B, T, epi_last, (float)nk, sj, unit_lens, mul, add, order));
            } else if (resblock_fusable(v, i, j)) {
                TRY(resblock_fused_launch(v, i, j, X, XS, B, T, epi_last, (float)nk, sj, unit_lens, mul, add, order));
            } else {
                // Operand planes (conv_split16.h; PARROT_PLANES=0 switches them off): the tensor between the two convs of a pair
                // exists only as the second conv's ready-made operand -- written once by the first conv's epilogue, the same
                // bytes per element as the fp32 tensor it replaces (two fp16 pieces), bit-identical operands -- and, in the
                // single-piece modes (bf16 / f16: 2 bytes per element), each pair's output is written as a plane beside the
                // fp32 residual, so that the next pair's first conv converts nothing either.
                bool planes = v->planes && !v->dbg_absmax && c.resblock_type == 1;
                for (int q = 0; q < per_rb && planes; ++q) planes = plane_ok(v->rb[base + q].get());
                const bool dual = planes && v->scheme != PARROT_PREC_F16X3;  // (one piece: the two planes share T1's 4 C T bytes per row)
                char* const plane_a = reinterpret_cast<char*>(T1);
                char* const plane_b = plane_a + (size_t)B * plane_row_bytes(v->scheme, v->chan(i), T);
                const void* xin = nullptr;  // plane of `r` for the next first conv (dual mode)
                for (int m = 0; m < nd; ++m) {
                    const bool last = (m == nd - 1);
                    float* dst = last ? XS : ((m & 1) ? RB : RA);
                    if (m > 0) TRY(amax(1 + i, r, n_act, sj));
                    if (c.resblock_type == 1) {
                        PlaneArgs pa;
                        pa.xplane = xin;
                        if (planes) { pa.yplane = plane_a; pa.plane_only = 1; pa.yslope = v->rb[base + 2 * m + 1]->d.pre_act == PRE_LRELU ? v->rb[base + 2 * m + 1]->d.pre_slope : 1.f; }
                        TRY(conv_launch(v->rb[base + 2 * m].get(), r, nullptr, T1, B, T, EPI_STORE, 1.f, 0, 0, 0, sj, unit_lens, mul, add, planes ? &pa : nullptr));
                        TRY(amax(1 + i, T1, n_act, sj));
                    }
                    if (last && order) HIP_TRY(hipStreamWaitEvent(sj, order, 0));
                    if (c.resblock_type == 1) {
                        PlaneArgs pa;
                        if (planes) pa.xplane = plane_a;
                        if (dual && !last) { pa.yplane = plane_b; pa.yslope = v->rb[base + 2 * m + 2]->d.pre_act == PRE_LRELU ? v->rb[base + 2 * m + 2]->d.pre_slope : 1.f; xin = plane_b; }
                        TRY(conv_launch(v->rb[base + 2 * m + 1].get(), T1, r, dst, B, T, last ? epi_last : EPI_STORE, (float)nk, 0, 0, 0, sj, unit_lens, mul, add, planes ? &pa : nullptr));
                    } else
                        TRY(conv_launch(v->rb[base + m].get(), r, r, dst, B, T, last ? epi_last : EPI_STORE, (float)nk, 0, 0, 0, sj, unit_lens, mul, add));
                    r = dst;
                }
            }
            if (ns > 1) HIP_TRY(hipEventRecord(ss.ev_last[j], sj));
        }
        if (ns > 1)  // join: every branch (and with it every reader of X and of the branch temporaries) is done
            for (int j = 0; j < nk; ++j) HIP_TRY(hipStreamWaitEvent(s, ss.ev_last[j], 0));
        TRY(snap(2 + 2 * i, XS, n_act));
        ia = (ia + 2) % 3;
    }
    TRY(amax(1 + c.n_stages, P[ia], (size_t)B * v->chan(c.n_stages - 1) * T, s));
    TRY(conv_launch(v->conv_post.get(), P[ia], nullptr, wav_out, B, T, EPI_STORE, 1.f, 0, 0, 0, s, unit_lens, mul, add));
    return PARROT_OK;
}

extern "C" int parrot_wav_to_int16(const float* wav, int16_t* out, size_t n, void* stream) {
    if (!wav || !out) return fail(PARROT_E_INVALID, "wav_to_int16: null argument");
    if (n == 0) return PARROT_OK;
    hipLaunchKernelGGL(wav_to_int16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, out, n);
    HIP_TRY(hipGetLastError());
    return PARROT_OK;
}

// ---------------------------------------------------------------------------------------------
// TTE
// ---------------------------------------------------------------------------------------------
struct FftLayer {
    std::unique_ptr<parrot_conv> qkv, in_proj, out_proj, wo, conv1, conv2;
    float *an_w = nullptr, *an_b = nullptr, *cn_w = nullptr, *cn_b = nullptr;
    int heads = 1;
    bool merged = false;  // qkv holds in_proj * qkv, wo holds wo * out_proj (in_proj / out_proj unused)
    ~FftLayer() {
        for (float* p : {an_w, an_b, cn_w, cn_b})
            if (p) (void)hipFree(p);
    }
};

struct parrot_tte {
    parrot_tte_cfg cfg{};
    float *pe = nullptr, *tok = nullptr, *spk = nullptr;
    float *ln0_w = nullptr, *ln0_b = nullptr, *ln1_w = nullptr, *ln1_b = nullptr;
    int* err = nullptr;
    std::unique_ptr<parrot_conv> dp0, dp1, dp_proj, head;
    std::vector<std::unique_ptr<FftLayer>> enc, dec;
    std::vector<float*> dbg_enc, dbg_dec;  // parrot_tte_debug_stages (tests only)
    int scheme = 0;                        // PARROT_PREC_* captured at create
    bool flash = false;                    // attention core on attn_flash_kernel (any T, no score tensor)
    // tie guard (argmax_cf_kernel / tie_guard_refine_kernel): fp32 head weights (transposed to (D, V)) for the fp64 re-evaluation, the (b, t) list of
    // the last decode's low-margin positions and its statistics {count, min margin bits, ids changed}
    float *head_w = nullptr, *head_b = nullptr;
    // (one set per decoder lane -- parrot_tte_decode_rows: row groups of one batch may decode concurrently on several streams --
    //  laid out back to back: lane l's list / statistics / refined logits start at l x the per-lane size)
    static constexpr int LANES = 1;
    int *glist = nullptr, *gstat = nullptr;
    int lanes_used = 1;  // bit l: lane l took part in the last decoded batch (host-side bookkeeping of the statistics readers)
    float guard = 1e-4f;
    // ... extended to the last decoder block's FFN output (round 4): for a guarded position the block's conv2 (1x1) + bias +
    // residual are re-evaluated in fp64 from the fp32 activations the block itself produced (relu(conv1) and x + attn), then the
    // head: last_w2t = that conv2's weight transposed to (F, D), last_b2 its bias; gref = the refined logits of the guarded
    // positions of the last decode (TIE_GUARD_MAX x V floats, parrot_tte_guard_logits)
    float *last_w2t = nullptr, *last_b2 = nullptr, *gref = nullptr;
    bool merged = true;
    ~parrot_tte() {
        for (float* p : {pe, tok, spk, ln0_w, ln0_b, ln1_w, ln1_b, head_w, head_b, last_w2t, last_b2, gref})
            if (p) (void)hipFree(p);
        if (err) (void)hipFree(err);
        if (glist) (void)hipFree(glist);
        if (gstat) (void)hipFree(gstat);
    }
};

// Default for handles created afterwards: fold the back-to-back bias-free projections of an FFT block (quirk Q3) into one each.
// PARROT_TTE_MERGE / parrot_set_tte_merge; parrot_tte_create_ex overrides it per handle.
static std::atomic<int> g_tte_merge{-1};
static int tte_merge_default() {
    int v = g_tte_merge.load();
    if (v < 0) {
        const char* e = getenv("PARROT_TTE_MERGE");
        v = (!e || atoi(e) != 0) ? 1 : 0;
        g_tte_merge.store(v);
    }
    return v;
}
extern "C" int parrot_set_tte_merge(int32_t on) {
    g_tte_merge.store(on ? 1 : 0);
    return PARROT_OK;
}
static int build_fft(std::unique_ptr<FftLayer>& slot, const parrot_tte_cfg& c, int heads, const parrot_fft_weights& w) {
    std::unique_ptr<FftLayer> L(new FftLayer());
    const int D = c.d_model, F = c.n_filter_ffn;
    if (D % heads) return fail(PARROT_E_INVALID, "tte_create: d_model % n_head != 0");  // fft.py:44
    L->heads = heads;
    // The reference projects twice on each side of the attention core (quirk Q3: the block's own bias-free qkv / wo
    // Linear around nn.MultiheadAttention's bias-free in_proj / out_proj, fft.py:48-57).  Two linear maps with
    // nothing in between are ONE linear map: the products are formed here in fp64 and rounded once to fp32
    //     W_qkv' = blockdiag(W_in_q, W_in_k, W_in_v) * W_qkv   (3D x D),     W_o' = W_wo * W_out   (D x D)
    // which removes two launches per block (PARROT_TTE_MERGE=0 keeps the four separate projections).
    const bool merge = (tl_merge >= 0 ? tl_merge : tte_merge_default()) != 0;
    L->merged = merge;
    if (merge) {
        std::vector<float> wq((size_t)3 * D * D), wo((size_t)D * D);
        std::vector<double> row(D);
        for (int g = 0; g < 3; ++g)
            for (int i = 0; i < D; ++i) {  // row i of group g: sum_j in_proj[gD+i][j] * qkv[gD+j][:]
                std::fill(row.begin(), row.end(), 0.0);
                for (int j = 0; j < D; ++j) {
                    const double a = w.in_proj[((size_t)g * D + i) * D + j];
                    const float* q = w.qkv + ((size_t)g * D + j) * D;
                    for (int c2 = 0; c2 < D; ++c2) row[c2] += a * (double)q[c2];
                }
                for (int c2 = 0; c2 < D; ++c2) wq[((size_t)g * D + i) * D + c2] = (float)row[c2];
            }
        for (int i = 0; i < D; ++i) {  // W_o'[i][:] = sum_j wo[i][j] * out_proj[j][:]
            std::fill(row.begin(), row.end(), 0.0);
            for (int j = 0; j < D; ++j) {
                const double a = w.wo[(size_t)i * D + j];
                const float* q = w.out_proj + (size_t)j * D;
                for (int c2 = 0; c2 < D; ++c2) row[c2] += a * (double)q[c2];
            }
            for (int c2 = 0; c2 < D; ++c2) wo[(size_t)i * D + c2] = (float)row[c2];
        }
        TRY(make_conv(L->qkv, D, 3 * D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, wq.data(), nullptr));
        TRY(make_conv(L->wo, D, D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, wo.data(), nullptr));
    } else {
        TRY(make_conv(L->qkv, D, 3 * D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w.qkv, nullptr));
        // MHA in_proj: three bias-free (D,D) projections of three different inputs = a grouped 1x1 conv
        TRY(make_conv(L->in_proj, 3 * D, 3 * D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w.in_proj, nullptr, 3));
        TRY(make_conv(L->out_proj, D, D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w.out_proj, nullptr));
        TRY(make_conv(L->wo, D, D, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w.wo, nullptr));
    }
    // the residual stream of an FFT block is ~10x larger than what a sub-layer adds to it: add it AFTER the sum (as the
    // reference does, fft.py:97,99), not as the accumulator's starting value
    L->wo->late_res = true;
    TRY(make_conv(L->conv1, D, F, c.ffn_k1, 1, (c.ffn_k1 - 1) / 2, 0, 1, PRE_NONE, 0.f, ACT_RELU, w.conv1_w, w.conv1_b));
    TRY(make_conv(L->conv2, F, D, c.ffn_k2, 1, (c.ffn_k2 - 1) / 2, 0, 1, PRE_NONE, 0.f, ACT_NONE, w.conv2_w, w.conv2_b));
    L->conv2->late_res = true;
    TRY(upload(&L->an_w, w.attn_norm_w, D));
    TRY(upload(&L->an_b, w.attn_norm_b, D));
    TRY(upload(&L->cn_w, w.conv_norm_w, D));
    TRY(upload(&L->cn_b, w.conv_norm_b, D));
    slot = std::move(L);
    return PARROT_OK;
}

static int tte_create_body(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w);
static int tte_create_impl(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w, int prec, int merge) {
    CreateScope scope(prec, -1, merge);  // (thread-local: the process defaults are not touched)
    return tte_create_body(out, cfg, w);
}
extern "C" int parrot_tte_create(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w) {
    return tte_create_impl(out, cfg, w, -1, -1);
}
extern "C" int parrot_tte_create_ex(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w, int32_t precision,
                                    int32_t merge_projections) {
    if (precision > PARROT_PREC_F16 || merge_projections > 1) return fail(PARROT_E_INVALID, "tte_create_ex: precision in -1 .. 4, merge_projections in -1 .. 1");
    return tte_create_impl(out, cfg, w, precision, merge_projections);
}
extern "C" int parrot_tte_precision(const parrot_tte_t* t) { return t ? t->scheme : PARROT_E_INVALID; }
static int tte_create_body(parrot_tte_t** out, const parrot_tte_cfg* cfg, const parrot_tte_weights* w) {
    if (!out || !cfg || !w) return fail(PARROT_E_INVALID, "tte_create: null argument");
    const parrot_tte_cfg& c = *cfg;
    if (c.d_model <= 0 || c.n_filter_ffn <= 0 || c.max_len <= 0 || c.vocab <= 0 || c.n_codes <= 0 || c.dp_filter <= 0 ||
        c.enc_layers < 0 || c.dec_layers < 0 || c.ffn_k1 <= 0 || c.ffn_k2 <= 0 || c.dp_kernel <= 0)
        return fail(PARROT_E_INVALID, "tte_create: bad config");
    if (!(c.ffn_k1 & 1) || !(c.ffn_k2 & 1)) return fail(PARROT_E_UNSUPPORTED, "tte_create: even FFN kernel sizes change the sequence length");
    if (c.dp_kernel != 3) return fail(PARROT_E_UNSUPPORTED, "tte_create: duration_predictor.kernel_size != 3 changes the sequence length in the reference (padding=1 is hard-coded, duration.py:34)");
    std::unique_ptr<parrot_tte> t(new parrot_tte());
    t->cfg = c;
    t->scheme = create_prec();
    {
        // flash attention runs on the fp16 split pipe: the default scheme and the fp16 reduced-precision mode take it; the exact
        // (f32), bf16x6 and bf16 handles keep the fp32-MFMA cores (fused for T <= 256, three kernels beyond)
        static const bool want = [] { const char* e = getenv("PARROT_FLASH_ATTN"); return !e || atoi(e) != 0; }();
        // (attn_flash_kernel is built on the fp16 pipe: a bf16 handle keeps fp32's exponent range by staying on the fp32-MFMA cores)
        const bool sch_ok = t->scheme == PARROT_PREC_F16X3 || t->scheme == PARROT_PREC_F16;
        auto hd_ok = [&](int layers, int heads) { return layers == 0 || (heads > 0 && c.d_model % heads == 0 && attn_flash_has(c.d_model / heads)); };
        t->flash = want && sch_ok && hd_ok(c.enc_layers, c.enc_heads) && hd_ok(c.dec_layers, c.dec_heads);
    }
    const int D = c.d_model;
    TRY(upload(&t->pe, w->pe, (size_t)c.max_len * D));
    TRY(upload(&t->tok, w->tok_emb, (size_t)c.vocab * D));
    if (c.n_speaker > 1) {
        if (!w->speaker_emb) return fail(PARROT_E_INVALID, "tte_create: n_speaker > 1 without speaker_emb");
        TRY(upload(&t->spk, w->speaker_emb, (size_t)c.n_speaker * D));
    }
    HIP_TRY(hipMalloc((void**)&t->err, sizeof(int)));
    HIP_TRY(hipMemset(t->err, 0, sizeof(int)));
    TRY(make_conv(t->dp0, D, c.dp_filter, c.dp_kernel, 1, (c.dp_kernel - 1) / 2, 0, 1, PRE_NONE, 0.f, ACT_NONE, w->dp_conv0_w, w->dp_conv0_b));
    TRY(make_conv(t->dp1, c.dp_filter, c.dp_filter, c.dp_kernel, 1, 1 /* Q4 */, 0, 1, PRE_NONE, 0.f, ACT_NONE, w->dp_conv1_w, w->dp_conv1_b));
    TRY(make_conv(t->dp_proj, c.dp_filter, 1, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w->dp_proj_w, w->dp_proj_b));
    TRY(upload(&t->ln0_w, w->dp_ln0_w, c.dp_filter));
    TRY(upload(&t->ln0_b, w->dp_ln0_b, c.dp_filter));
    TRY(upload(&t->ln1_w, w->dp_ln1_w, c.dp_filter));
    TRY(upload(&t->ln1_b, w->dp_ln1_b, c.dp_filter));
    t->merged = (tl_merge >= 0 ? tl_merge : tte_merge_default()) != 0;
    t->enc.resize(c.enc_layers);
    t->dec.resize(c.dec_layers);
    for (int i = 0; i < c.enc_layers; ++i) TRY(build_fft(t->enc[i], c, c.enc_heads, w->enc[i]));
    for (int i = 0; i < c.dec_layers; ++i) TRY(build_fft(t->dec[i], c, c.dec_heads, w->dec[i]));
    TRY(make_conv(t->head, D, c.n_codes, 1, 1, 0, 0, 1, PRE_NONE, 0.f, ACT_NONE, w->head_w, w->head_b));
    {   // tie guard: PARROT_TIE_GUARD = margin below which a position's head is re-evaluated in fp64 (0 switches it off)
        const char* e = getenv("PARROT_TIE_GUARD");
        t->guard = e ? (float)atof(e) : 1e-4f;
        {   // (D, V): the refine kernel reads one code per thread, coalesced
            std::vector<float> wt((size_t)c.n_codes * D);
            for (int v = 0; v < c.n_codes; ++v)
                for (int ch = 0; ch < D; ++ch) wt[(size_t)ch * c.n_codes + v] = w->head_w[(size_t)v * D + ch];
            TRY(upload(&t->head_w, wt.data(), wt.size()));
        }
        if (w->head_b) TRY(upload(&t->head_b, w->head_b, (size_t)c.n_codes));
        if (c.dec_layers > 0 && c.ffn_k2 == 1) {  // deep guard: the last decoder block's conv2 as (F, D) + its bias
            const parrot_fft_weights& lw = w->dec[c.dec_layers - 1];
            const int F = c.n_filter_ffn;
            std::vector<float> wt((size_t)F * D);
            for (int o = 0; o < D; ++o)
                for (int j = 0; j < F; ++j) wt[(size_t)j * D + o] = lw.conv2_w[(size_t)o * F + j];
            TRY(upload(&t->last_w2t, wt.data(), wt.size()));
            if (lw.conv2_b) TRY(upload(&t->last_b2, lw.conv2_b, (size_t)D));
        }
        HIP_TRY(hipMalloc((void**)&t->gref, (size_t)parrot_tte::LANES * TIE_GUARD_MAX * c.n_codes * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&t->glist, (size_t)parrot_tte::LANES * 2 * TIE_GUARD_MAX * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&t->gstat, (size_t)parrot_tte::LANES * 4 * sizeof(int)));
        HIP_TRY(hipMemset(t->gstat, 0, (size_t)parrot_tte::LANES * 4 * sizeof(int)));
    }
    *out = t.release();
    return PARROT_OK;
}
extern "C" void parrot_tte_destroy(parrot_tte_t* t) { delete t; }

struct TteState {  // persists between encode and decode (sized by B,S only)
    float* enc_out;
    int32_t* cum;
    int32_t* out_len;
};
static TteState tte_state(const parrot_tte* t, Arena& a, int B, int S) {
    TteState st;
    st.enc_out = a.take<float>((size_t)B * t->cfg.d_model * S);
    st.cum = a.take<int32_t>((size_t)B * S);
    st.out_len = a.take<int32_t>((size_t)B);
    return st;
}
struct TteScratch {
    float *x, *n, *qkv1, *qkv2, *scores, *ctx, *o, *h, *f, *logits;
};
static TteScratch tte_scratch(const parrot_tte* t, Arena& a, int B, int T, bool with_logits) {
    const parrot_tte_cfg& c = t->cfg;
    const int Hmax = std::max(std::max(c.enc_heads, c.dec_heads), 1);
    const size_t DT = (size_t)B * c.d_model * T;
    const int Fmax = std::max(c.n_filter_ffn, c.dp_filter);
    TteScratch s;
    s.x = a.take<float>(DT);
    s.n = a.take<float>(std::max(DT, (size_t)B * c.dp_filter * T));
    s.qkv1 = a.take<float>(3 * DT);
    s.qkv2 = a.take<float>(3 * DT);
    s.scores = t->flash ? nullptr : a.take<float>((size_t)B * Hmax * T * T);  // (only the three-kernel attention path materialises scores)
    s.ctx = a.take<float>(DT);
    s.o = a.take<float>(DT);
    s.h = a.take<float>(DT);
    s.f = a.take<float>((size_t)B * Fmax * T);
    s.logits = with_logits ? a.take<float>((size_t)B * c.n_codes * T) : nullptr;
    return s;
}

extern "C" size_t parrot_tte_state_bytes(const parrot_tte_t* t, int32_t B, int32_t S) {
    if (!t || B <= 0 || S <= 0) return 0;
    Arena a(nullptr, 0);
    (void)tte_state(t, a, B, S);
    return align_up(a.off, 256);
}
extern "C" size_t parrot_tte_workspace_bytes(const parrot_tte_t* t, int32_t B, int32_t S, int32_t L_max) {
    if (!t || B <= 0 || S <= 0) return 0;
    Arena a(nullptr, 0);
    (void)tte_scratch(t, a, B, std::max(S, L_max), L_max > 0);
    return align_up(a.off, 256);
}

static int layernorm(const float* x, const float* g, const float* b, float* y, int B, int C, int T, int relu_in, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_cf_kernel<16>, dim3((T + 63) / 64, B), dim3(16 * 64), 0, s, x, g, b, y, C, T, 1e-5f, relu_in);
    HIP_TRY(hipGetLastError());
    return PARROT_OK;
}

// FFTBlock.forward (fft.py:94-100): x -> out (may alias x).  valid (B,T) u8: 1 = attend to this key.
// row_len (B) i32 device, nullable: ROW-EXACT mode -- row b holds row_len[b] real positions and every conv applies its zero padding
// at the row's own end (the reference run of that utterance alone, fft.py:78-82); NULL: the reference's padded-batch semantics, pad
// frames leak through the k = 9 conv (quirk Q7).
static int fft_block(const parrot_tte* t, const FftLayer* L, TteScratch& w, float* x, const uint8_t* valid, int B, int T, hipStream_t s,
                     const int32_t* row_len = nullptr) {
    const int D = t->cfg.d_model, H = L->heads, hd = D / H;
    TRY(layernorm(x, L->an_w, L->an_b, w.n, B, D, T, 0, s));
    if (L->merged) {
        TRY(conv_launch(L->qkv.get(), w.n, nullptr, w.qkv2, B, T, EPI_STORE, 1.f, 0, 0, 0, s));
    } else {
        TRY(conv_launch(L->qkv.get(), w.n, nullptr, w.qkv1, B, T, EPI_STORE, 1.f, 0, 0, 0, s));
        TRY(conv_launch(L->in_proj.get(), w.qkv1, nullptr, w.qkv2, B, T, EPI_STORE, 1.f, 0, 0, 0, s));
    }
    const long DT = (long)D * T;
    if (t->flash) {  // any T, online softmax, no score tensor (attn.h: attn_flash_kernel)
        AttnParams p{};
        p.qkv = w.qkv2; p.valid = valid; p.ctx = w.ctx;
        p.T = T; p.H = H; p.D = D; p.hd = hd;
        p.alpha = (float)std::sqrt(1.0 / (double)hd);
        HIP_TRY(launch_attn_flash(p, B, s));
    } else if (T <= ATTN_TMAX && hd == 128) {  // scores, softmax and context in one launch (attn.h)
        AttnParams p{};
        p.qkv = w.qkv2; p.valid = valid; p.ctx = w.ctx;
        p.T = T; p.H = H; p.D = D; p.hd = hd;
        p.alpha = (float)std::sqrt(1.0 / (double)hd);
        const size_t lds = (size_t)32 * (((T + 31) / 32) * 32 + 1) * sizeof(float);  // 32-query tiles (33 KiB at T = 256: no opt-in needed)
        hipLaunchKernelGGL((attn_fused_kernel<128, 32>), dim3((T + 31) / 32, B * H), dim3(256), lds, s, p);
        HIP_TRY(hipGetLastError());
    } else {
    {   // scores[b,h][tq][tk] = sum_c (q[c][tq] * sqrt(1/hd)) * k[c][tk]
        BgemmParams p{};
        p.A = w.qkv2; p.B = w.qkv2 + DT; p.C = w.scores;
        p.M = T; p.N = T; p.K = hd;
        p.a_sk = T; p.a_sm = 1; p.b_sk = T; p.b_sn = 1;
        p.a_zb = 3 * DT; p.a_zh = (long)hd * T; p.b_zb = 3 * DT; p.b_zh = (long)hd * T;
        p.c_zb = (long)H * T * T; p.c_zh = (long)T * T; p.ldc = T; p.H = H;
        p.alpha = (float)std::sqrt(1.0 / (double)hd);
        hipLaunchKernelGGL(bgemm_mfma_kernel, dim3((T + 63) / 64, (T + 63) / 64, B * H), dim3(256), 0, s, p);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(softmax_mask_kernel, dim3((B * H * T + 3) / 4), dim3(256), 0, s, w.scores, valid, B * H * T, T, H * T);
    HIP_TRY(hipGetLastError());
    {   // ctx[b][h*hd + c][tq] = sum_tk v[c][tk] * P[tq][tk]
        BgemmParams p{};
        p.A = w.qkv2 + 2 * DT; p.B = w.scores; p.C = w.ctx;
        p.M = hd; p.N = T; p.K = T;
        p.a_sk = 1; p.a_sm = T; p.b_sk = 1; p.b_sn = T;
        p.a_zb = 3 * DT; p.a_zh = (long)hd * T; p.b_zb = (long)H * T * T; p.b_zh = (long)T * T;
        p.c_zb = DT; p.c_zh = (long)hd * T; p.ldc = T; p.H = H;
        p.alpha = 1.0f;
        hipLaunchKernelGGL(bgemm_mfma_kernel, dim3((T + 63) / 64, (hd + 63) / 64, B * H), dim3(256), 0, s, p);
        HIP_TRY(hipGetLastError());
    }
    }
    if (L->merged) {
        TRY(conv_launch(L->wo.get(), w.ctx, x, w.h, B, T, EPI_STORE, 1.f, 0, 0, 0, s));        // h = x + attn
    } else {
        TRY(conv_launch(L->out_proj.get(), w.ctx, nullptr, w.o, B, T, EPI_STORE, 1.f, 0, 0, 0, s));
        TRY(conv_launch(L->wo.get(), w.o, x, w.h, B, T, EPI_STORE, 1.f, 0, 0, 0, s));          // h = x + attn
    }
    TRY(layernorm(w.h, L->cn_w, L->cn_b, w.n, B, D, T, 0, s));
    // (a 1x1 conv has no neighbours to leak from: only the k > 1 convs take the per-row ends)
    TRY(conv_launch(L->conv1.get(), w.n, nullptr, w.f, B, T, EPI_STORE, 1.f, 0, 0, 0, s, t->cfg.ffn_k1 > 1 ? row_len : nullptr));  // relu fused
    TRY(conv_launch(L->conv2.get(), w.f, w.h, x, B, T, EPI_STORE, 1.f, 0, 0, 0, s, t->cfg.ffn_k2 > 1 ? row_len : nullptr));      // out = h + ffn
    return PARROT_OK;
}

// Encode rows [row0, row0 + B) of a batch of Bfull rows: every pointer argument is the GROUP's first row; the group's encoder
// output / duration prefix sums land in rows row0.. of `state` (sized for Bfull rows).  The encoder works row by row and pe[S] is
// indexed by the padded length S alone (fft.py:18), so a row's result does not depend on the grouping.
static int tte_encode_rows(parrot_tte_t* t, const int64_t* phones, const uint8_t* src_mask, const int64_t* speaker, const int32_t* src_len,
                           int32_t Bfull, int32_t S, int32_t row0, int32_t B, float* log_dur, int64_t* dur, int32_t* out_lens, void* state,
                           size_t state_bytes, void* ws, size_t ws_bytes, void* stream) {
    if (!t || !phones || !src_mask || !log_dur || !dur || !out_lens || !state || !ws) return fail(PARROT_E_INVALID, "tte_encode: null argument");
    if (Bfull <= 0 || S <= 0) return fail(PARROT_E_INVALID, "tte_encode: empty batch");
    if (row0 < 0 || B <= 0 || row0 + B > Bfull) return fail(PARROT_E_INVALID, "tte_encode: row group outside the batch");
    const parrot_tte_cfg& c = t->cfg;
    if (S >= c.max_len) return fail(PARROT_E_RANGE, "tte_encode: sequence length >= max_len (pe[T] out of range, fft.py:18)");
    if (t->spk && !speaker) return fail(PARROT_E_INVALID, "tte_encode: multi-speaker model needs speaker ids");
    hipStream_t s = (hipStream_t)stream;
    Arena sa(state, state_bytes);
    TteState st = tte_state(t, sa, Bfull, S);
    Arena a(ws, ws_bytes);
    TteScratch w = tte_scratch(t, a, B, S, false);
    if (!sa.ok || !a.ok) return fail(PARROT_E_NOMEM, "tte_encode: state/workspace too small");
    st.enc_out += (size_t)row0 * c.d_model * S;
    st.cum += (size_t)row0 * S;
    st.out_len += row0;
    const int D = c.d_model;
    // pe[S] of the padded batch (quirk Q1 / Q7), or -- row-exact -- pe[src_len[b]]: what the row's own B = 1 run adds (fft.py:18)
    hipLaunchKernelGGL(tte_embed_kernel, dim3((S + 63) / 64, (D + 63) / 64, B), dim3(256), 0, s, phones, t->tok, t->pe, src_len,
                       w.x, S, D, c.vocab, t->err);
    HIP_TRY(hipGetLastError());
    auto dbg = [&](const std::vector<float*>& v, size_t idx, const float* src, size_t n) -> int {
        if (idx < v.size() && v[idx]) HIP_TRY(hipMemcpyAsync(v[idx] + (size_t)row0 * D * S, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return PARROT_OK;
    };
    TRY(dbg(t->dbg_enc, 0, w.x, (size_t)B * D * S));
    for (size_t n = 0; n < t->enc.size(); ++n) {
        TRY(fft_block(t, t->enc[n].get(), w, w.x, src_mask, B, S, s, src_len));
        TRY(dbg(t->dbg_enc, 1 + n, w.x, (size_t)B * D * S));
    }
    if (t->spk) {
        const size_t total = (size_t)B * D * S;
        hipLaunchKernelGGL(add_channel_vec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w.x, speaker, t->spk, D, S,
                           c.n_speaker, total, t->err);
        HIP_TRY(hipGetLastError());
    }
    TRY(dbg(t->dbg_enc, 1 + t->enc.size(), w.x, (size_t)B * D * S));
    HIP_TRY(hipMemcpyAsync(st.enc_out, w.x, (size_t)B * D * S * sizeof(float), hipMemcpyDeviceToDevice, s));
    // duration predictor (duration.py:29-48): conv -> relu -> LN -> conv(pad 1) -> relu -> LN -> linear
    const int NF = c.dp_filter;
    TRY(conv_launch(t->dp0.get(), w.x, nullptr, w.f, B, S, EPI_STORE, 1.f, 0, 0, 0, s, src_len));
    TRY(layernorm(w.f, t->ln0_w, t->ln0_b, w.n, B, NF, S, 1, s));
    TRY(conv_launch(t->dp1.get(), w.n, nullptr, w.f, B, S, EPI_STORE, 1.f, 0, 0, 0, s, src_len));
    TRY(layernorm(w.f, t->ln1_w, t->ln1_b, w.n, B, NF, S, 1, s));
    TRY(conv_launch(t->dp_proj.get(), w.n, nullptr, w.o, B, S, EPI_STORE, 1.f, 0, 0, 0, s));  // (B,1,S)
    hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(256), 0, s, w.o, src_mask, log_dur, dur, st.cum, st.out_len, S, src_len);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_lens, st.out_len, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return PARROT_OK;
}
extern "C" int parrot_tte_encode(parrot_tte_t* t, const int64_t* phones, const uint8_t* src_mask, const int64_t* speaker,
                                 const int32_t* src_len, int32_t B, int32_t S, float* log_dur, int64_t* dur, int32_t* out_lens, void* state,
                                 size_t state_bytes, void* ws, size_t ws_bytes, void* stream) {
    if (poison_word() && B > 0 && S > 0) {
        hipStream_t s = (hipStream_t)stream;
        TRY(poison(state, state_bytes, s));
        TRY(poison(ws, ws_bytes, s));
        TRY(poison(log_dur, (size_t)B * S * sizeof(float), s));
        TRY(poison(dur, (size_t)B * S * sizeof(int64_t), s));
        TRY(poison(out_lens, (size_t)B * sizeof(int32_t), s));
    }
    return tte_encode_rows(t, phones, src_mask, speaker, src_len, B, S, 0, B, log_dur, dur, out_lens, state, state_bytes, ws, ws_bytes, stream);
}

// Decode rows [row0, row0 + n) of the batch that parrot_tte_encode left in `state` (B rows).  ids / tgt_mask / logits point at the
// group's own first row.  L is the WHOLE batch's expanded length (pe[L], parrot.py:106) whichever rows are decoded, and every
// kernel of the decoder works row by row, so a row decoded in a group equals the same row decoded with the whole batch bit for bit.
static int tte_decode_rows(parrot_tte_t* t, int32_t Bfull, int32_t S, int32_t L, int32_t row0, int32_t B, int64_t* ids, uint8_t* tgt_mask,
                           float* logits, void* state, size_t state_bytes, void* ws, size_t ws_bytes, void* stream, int lane, bool guard_restart,
                           bool new_batch, bool row_exact) {
    if (lane < 0 || lane >= parrot_tte::LANES) return fail(PARROT_E_INVALID, "tte_decode: lane out of range");
    if (!t || !ids || !tgt_mask || !state || !ws) return fail(PARROT_E_INVALID, "tte_decode: null argument");
    if (Bfull <= 0 || S <= 0) return fail(PARROT_E_INVALID, "tte_decode: empty batch");
    if (row0 < 0 || B <= 0 || row0 + B > Bfull) return fail(PARROT_E_INVALID, "tte_decode: row group outside the encoded batch");
    if (L <= 0) return fail(PARROT_E_INVALID, "tte_decode: L must be > 0 (all durations zero: the reference fails in MultiheadAttention too)");
    const parrot_tte_cfg& c = t->cfg;
    if (L >= c.max_len) return fail(PARROT_E_RANGE, "tte_decode: expanded length >= max_len (pe[T] out of range, fft.py:18)");
    hipStream_t s = (hipStream_t)stream;
    Arena sa(state, state_bytes);
    TteState st = tte_state(t, sa, Bfull, S);
    Arena a(ws, ws_bytes);
    TteScratch w = tte_scratch(t, a, B, std::max(S, L), true);
    if (!sa.ok || !a.ok) return fail(PARROT_E_NOMEM, "tte_decode: state/workspace too small");
    const int D = c.d_model, V = c.n_codes;
    // tie-guard state of this lane: a lane's first group of a batch restarts its statistics, later groups of the lane append
    int* const gstat = t->gstat ? t->gstat + 4 * lane : nullptr;
    int* const glist = t->glist ? t->glist + (size_t)lane * 2 * TIE_GUARD_MAX : nullptr;
    float* const gref = t->gref ? t->gref + (size_t)lane * TIE_GUARD_MAX * V : nullptr;
    t->lanes_used = (new_batch ? 0 : t->lanes_used) | (1 << lane);
    hipLaunchKernelGGL(length_regulate_kernel, dim3((L + 63) / 64, B), dim3(256), 0, s, st.enc_out + (size_t)row0 * D * S, st.cum + (size_t)row0 * S,
                       st.out_len + row0, t->pe, w.x, tgt_mask, S, L, D, t->guard > 0.f ? gstat : nullptr, guard_restart ? 1 : 0, row_exact ? 1 : 0);
    HIP_TRY(hipGetLastError());
    auto dbg = [&](size_t idx, const float* src, size_t n) -> int {
        if (idx < t->dbg_dec.size() && t->dbg_dec[idx])
            HIP_TRY(hipMemcpyAsync(t->dbg_dec[idx] + (size_t)row0 * D * L, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return PARROT_OK;
    };
    TRY(dbg(0, w.x, (size_t)B * D * L));
    for (size_t n = 0; n < t->dec.size(); ++n) {
        TRY(fft_block(t, t->dec[n].get(), w, w.x, tgt_mask, B, L, s, row_exact ? st.out_len + row0 : nullptr));
        TRY(dbg(1 + n, w.x, (size_t)B * D * L));
    }
    TRY(conv_launch(t->head.get(), w.x, nullptr, w.logits, B, L, EPI_STORE, 1.f, 0, 0, 0, s));
    {   // argmax + tie guard: gstat = {count, ids changed, min margin (float bits)} of this decode
        const bool on = t->guard > 0.f;  // (length_regulate_kernel, the first kernel of this decode, has reset gstat)
        hipLaunchKernelGGL(argmax_cf_kernel, dim3((L + 63) / 64, B), dim3(64 * ARGMAX_WAVES), 0, s, w.logits, ids, V, L, t->err, t->guard,
                           on ? glist : nullptr, on ? gstat : nullptr, row0);
        HIP_TRY(hipGetLastError());
        if (on) {  // re-evaluate the head of the low-margin positions in fp64 (workgroups beyond the count exit at once)
            // w.f / w.h still hold the last decoder block's relu(conv1) and x + attn: with them the refinement starts one layer
            // earlier (conv2 + bias + residual in fp64, then the head); without a decoder block it starts at w.x
            const bool deep = t->last_w2t != nullptr && !t->dec.empty();
            const int F = c.n_filter_ffn;
            hipLaunchKernelGGL(tie_guard_refine_kernel, dim3(TIE_GUARD_MAX), dim3(256), (size_t)(D + (deep ? F : 0)) * sizeof(double), s, w.x,
                               t->head_w, t->head_b, ids, D, V, L, glist, gstat, deep ? w.f : nullptr, deep ? w.h : nullptr, t->last_w2t,
                               t->last_b2, F, gref, row0);
            HIP_TRY(hipGetLastError());
        }
    }
    if (logits) {
        hipLaunchKernelGGL(transpose_cf_to_cl_kernel, dim3((L + 63) / 64, (V + 63) / 64, B), dim3(256), 0, s, w.logits, logits, V, L);
        HIP_TRY(hipGetLastError());
    }
    return PARROT_OK;
}
extern "C" int parrot_tte_decode(parrot_tte_t* t, int32_t B, int32_t S, int32_t L, int32_t row_exact, int64_t* ids, uint8_t* tgt_mask,
                                 float* logits, void* state, size_t state_bytes, void* ws, size_t ws_bytes, void* stream) {
    if (poison_word() && t && B > 0 && L > 0) {
        hipStream_t s = (hipStream_t)stream;
        TRY(poison(ws, ws_bytes, s));
        TRY(poison(ids, (size_t)B * L * sizeof(int64_t), s));
        TRY(poison(tgt_mask, (size_t)B * L, s));
        TRY(poison(logits, (size_t)B * L * t->cfg.n_codes * sizeof(float), s));
    }
    return tte_decode_rows(t, B, S, L, 0, B, ids, tgt_mask, logits, state, state_bytes, ws, ws_bytes, stream, 0, true, true, row_exact != 0);
}

extern "C" int parrot_tte_debug_stages(parrot_tte_t* t, float* const* enc_ptrs, float* const* dec_ptrs) {
    if (!t) return fail(PARROT_E_INVALID, "tte_debug_stages: null handle");
    t->dbg_enc.clear();
    t->dbg_dec.clear();
    if (enc_ptrs) t->dbg_enc.assign(enc_ptrs, enc_ptrs + t->enc.size() + 2);
    if (dec_ptrs) t->dbg_dec.assign(dec_ptrs, dec_ptrs + t->dec.size() + 1);
    return PARROT_OK;
}

// length_regulator on its own (duration.py:6-24): channel-last in / out around the decoder's kernel
extern "C" size_t parrot_length_regulator_workspace_bytes(int32_t B, int32_t S, int32_t D, int32_t L) {
    if (B <= 0 || S <= 0 || D <= 0 || L < 0) return 0;
    Arena a(nullptr, 0);
    a.take<float>((size_t)B * D * S);
    a.take<float>((size_t)B * D * std::max(L, 1));
    a.take<float>((size_t)D);
    a.take<int32_t>((size_t)B * S);
    a.take<int32_t>((size_t)B);
    return align_up(a.off, 256);
}
extern "C" int parrot_length_regulator(const float* seq, const int64_t* dur, int32_t B, int32_t S, int32_t D, int32_t L, float* out,
                                       uint8_t* mask, int32_t* out_lens, void* ws, size_t ws_bytes, void* stream) {
    if (!seq || !dur || !out || !mask || !out_lens || !ws) return fail(PARROT_E_INVALID, "length_regulator: null argument");
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0) return fail(PARROT_E_INVALID, "length_regulator: empty batch or L = 0");
    hipStream_t s = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    float* seq_cf = a.take<float>((size_t)B * D * S);
    float* out_cf = a.take<float>((size_t)B * D * L);
    float* zero = a.take<float>((size_t)D);
    int32_t* cum = a.take<int32_t>((size_t)B * S);
    int32_t* lens = a.take<int32_t>((size_t)B);
    if (!a.ok) return fail(PARROT_E_NOMEM, "length_regulator: workspace too small");
    if (poison_word()) {
        TRY(poison(ws, ws_bytes, s));
        TRY(poison(out, (size_t)B * L * D * sizeof(float), s));
        TRY(poison(mask, (size_t)B * L, s));
        TRY(poison(out_lens, (size_t)B * sizeof(int32_t), s));
    }
    HIP_TRY(hipMemsetAsync(zero, 0, (size_t)D * sizeof(float), s));
    // (B,S,D) -> (B,D,S): the transpose kernel with the roles of C and T swapped
    hipLaunchKernelGGL(transpose_cf_to_cl_kernel, dim3((D + 63) / 64, (S + 63) / 64, B), dim3(256), 0, s, seq, seq_cf, S, D);
    hipLaunchKernelGGL(dur_prefix_kernel, dim3(B), dim3(256), 0, s, dur, cum, lens, S);
    hipLaunchKernelGGL(length_regulate_kernel, dim3((L + 63) / 64, B), dim3(256), 0, s, seq_cf, cum, lens, zero, out_cf, mask, S, L, D, nullptr, 1, 0, 0);
    hipLaunchKernelGGL(transpose_cf_to_cl_kernel, dim3((L + 63) / 64, (D + 63) / 64, B), dim3(256), 0, s, out_cf, out, D, L);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_lens, lens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return PARROT_OK;
}

// plain dword copy (dst[i] = src[i]): the known-byte-count kernel used to calibrate the rocprofv3
// FETCH_SIZE / WRITE_SIZE counters for this library's 4-byte-per-lane access pattern.
extern "C" int parrot_debug_copy(const float* src, float* dst, size_t n, void* stream) {
    if (!src || !dst) return fail(PARROT_E_INVALID, "debug_copy: null argument");
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    HIP_TRY(hipGetLastError());
    return PARROT_OK;
}

// Sustained rate of a bare 16-bit MFMA stream on THIS device under its power limit (tools/probes/mfma_power.hip as a library
// call, so bench.py can put the ceiling next to the kernels' rates): 2 waves per SIMD, 24 / 48 MFMAs per loop iteration on four /
// eight independent accumulators, random fp16 operands with exponents near 1 (or one constant), ~20-40 ms of work.
template <int SHAPE>
static __global__ __launch_bounds__(256) void mfma_ceiling_kernel(const s16x8* in, float* out, int iters) {
    s16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = in[threadIdx.x % 64 + 64 * i]; b[i] = in[threadIdx.x % 64 + 64 * (i + 6)]; }
    float sum = 0.f;
    if (SHAPE == 0) {  // v_mfma_f32_32x32x16_f16
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int g = 0; g < 24; ++g)
                acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(g >> 2) % 6]), __builtin_bit_cast(f16x8, b[g % 6]), acc[g & 3], 0, 0, 0);
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    } else {  // v_mfma_f32_16x16x32_f16
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int g = 0; g < 48; ++g)
                acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(g >> 3) % 6]), __builtin_bit_cast(f16x8, b[g % 6]), acc[g & 7], 0, 0, 0);
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) sum += acc[t][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
extern "C" int parrot_debug_mfma_ceiling(int32_t shape, int32_t constant_data, double* tflops_out) {
    if (!tflops_out || shape < 0 || shape > 1) return fail(PARROT_E_INVALID, "mfma_ceiling: shape 0 (32x32x16) or 1 (16x16x32)");
    query_device();
    s16x8* in = nullptr;
    float* out = nullptr;
    const int grid = g_num_cus * 2, iters = 20000;
    std::vector<uint16_t> h(64 * 12 * 8);
    unsigned st = 12345u;
    for (auto& v : h) {
        st = st * 1664525u + 1013904223u;
        v = constant_data ? 0x3c00 : (uint16_t)(((st >> 16) & 0x83ff) | (0x3800 + ((st >> 9) & 0x400)));
    }
    HIP_TRY(hipMalloc((void**)&in, h.size() * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void**)&out, (size_t)grid * 256 * sizeof(float)));
    HIP_TRY(hipMemcpy(in, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto run = [&](int n) {
        if (shape == 0) hipLaunchKernelGGL(mfma_ceiling_kernel<0>, dim3(grid), dim3(256), 0, nullptr, in, out, n);
        else hipLaunchKernelGGL(mfma_ceiling_kernel<1>, dim3(grid), dim3(256), 0, nullptr, in, out, n);
    };
    run(2000);  // warm-up: lets the clock settle under load
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(e0, nullptr));
    run(iters);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *tflops_out = (double)grid * 4 * iters * 24 * 32768.0 / ms / 1e9;  // (48 x 16384 flops per iteration for the 16x16x32 shape: the same)
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(in);
    (void)hipFree(out);
    return PARROT_OK;
}

// device-side input-range flag (bad unit / speaker / phone ids <-> the reference's Embedding IndexError).
// Synchronises the stream; returns 0 or PARROT_E_RANGE and clears the flag.
static int read_flag(int* err, hipStream_t s, const char* who) {
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, err, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h) {
        HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s));
        if (h == 5)
            return fail(PARROT_E_NONFINITE, std::string(who) + ": non-finite output (waveform sample / logits) -- an activation left the range of the fp16 split "
                                                                "scheme (|x| < 8190); create the handle with PARROT_PREC_BF16X6 or PARROT_PREC_F32");
        return fail(PARROT_E_RANGE, std::string(who) + ": embedding index out of range (code " + std::to_string(h) + ")");
    }
    return PARROT_OK;
}
// The flag without a synchronisation: copy it to dst_dev[0] (device memory) on `stream` and clear it, so the caller can read it
// with a device-to-host transfer it performs anyway (the shims fetch it together with the TTE's expanded lengths).
static int status_async(int* err, int32_t* dst_dev, hipStream_t s) {
    if (!dst_dev) return fail(PARROT_E_INVALID, "status_async: null destination");
    HIP_TRY(hipMemcpyAsync(dst_dev, err, sizeof(int), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s));
    return PARROT_OK;
}
extern "C" int parrot_voc_status_async(parrot_voc_t* v, int32_t* dst_dev, void* stream) { return v ? status_async(v->err, dst_dev, (hipStream_t)stream) : PARROT_E_INVALID; }
// ... and without clearing it (the shims' first-forward range probe: a bad-id flag stays for the regular reporting path)
extern "C" int parrot_voc_status_peek_async(parrot_voc_t* v, int32_t* dst_dev, void* stream) {
    if (!v || !dst_dev) return fail(PARROT_E_INVALID, "voc_status_peek: null argument");
    HIP_TRY(hipMemcpyAsync(dst_dev, v->err, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PARROT_OK;
}
extern "C" int parrot_tte_status_peek_async(parrot_tte_t* t, int32_t* dst_dev, void* stream) {
    if (!t || !dst_dev) return fail(PARROT_E_INVALID, "tte_status_peek: null argument");
    HIP_TRY(hipMemcpyAsync(dst_dev, t->err, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PARROT_OK;
}
extern "C" int parrot_tte_status_async(parrot_tte_t* t, int32_t* dst_dev, void* stream) { return t ? status_async(t->err, dst_dev, (hipStream_t)stream) : PARROT_E_INVALID; }
// Tie-guard statistics of the last decode, copied to dst_dev[0..2] (device memory) on `stream` without synchronising:
// {positions whose top-2 logit margin was below the guard, ids changed by the fp64 re-evaluation of the head, the smallest
// margin of the call as float bits}
static __global__ void guard_stats_sum_kernel(const int* __restrict__ gstat, int mask, int* __restrict__ dst) {
    int n = 0, ch = 0, mn = 0x7f800000;
    for (int l = 0; l < parrot_tte::LANES; ++l)
        if ((mask >> l) & 1) {
            n += gstat[4 * l];
            ch += gstat[4 * l + 1];
            mn = min(mn, gstat[4 * l + 2]);
        }
    dst[0] = n; dst[1] = ch; dst[2] = mn;
}
// out row i = the i-th guarded position of the batch, lanes in order (each lane holds at most TIE_GUARD_MAX)
static __global__ void guard_gather_kernel(const float* __restrict__ gref, const int* __restrict__ glist, const int* __restrict__ gstat, int mask,
                                           int V, int max_n, float* __restrict__ logits, int* __restrict__ list) {
    const int i = blockIdx.x;
    int base = 0, lane = -1, j = 0;
    for (int l = 0; l < parrot_tte::LANES && lane < 0; ++l)
        if ((mask >> l) & 1) {
            const int nl = min(gstat[4 * l], TIE_GUARD_MAX);
            if (i < base + nl) { lane = l; j = i - base; }
            base += nl;
        }
    if (lane < 0 || i >= max_n) return;
    const float* src = gref + ((size_t)lane * TIE_GUARD_MAX + j) * V;
    for (int v = threadIdx.x; v < V; v += blockDim.x) logits[(size_t)i * V + v] = src[v];
    if (threadIdx.x < 2) list[2 * i + threadIdx.x] = glist[((size_t)lane * TIE_GUARD_MAX + j) * 2 + threadIdx.x];
}
extern "C" int parrot_tte_guard_stats_async(parrot_tte_t* t, int32_t* dst_dev, void* stream) {
    if (!t || !dst_dev) return fail(PARROT_E_INVALID, "tte_guard_stats: null argument");
    if (!t->gstat) {
        HIP_TRY(hipMemsetAsync(dst_dev, 0, 3 * sizeof(int), (hipStream_t)stream));
        return PARROT_OK;
    }
    hipLaunchKernelGGL(guard_stats_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t->gstat, t->lanes_used, dst_dev);
    HIP_TRY(hipGetLastError());
    return PARROT_OK;
}
// Refined (fp64-evaluated, rounded to fp32) logits of the guarded positions of the last decode: logits_dev (max_n x V floats) and
// their (b, t) pairs list_dev (2 max_n ints), device memory, no synchronisation; the count is guard_stats[0] (at most 256 per decoder lane).
extern "C" int parrot_tte_guard_logits(parrot_tte_t* t, float* logits_dev, int32_t* list_dev, int32_t max_n, void* stream) {
    if (!t || !logits_dev || !list_dev || max_n <= 0) return fail(PARROT_E_INVALID, "tte_guard_logits: null argument");
    if (!t->gref) return fail(PARROT_E_UNSUPPORTED, "tte_guard_logits: the tie guard of this handle is off");
    hipLaunchKernelGGL(guard_gather_kernel, dim3(max_n), dim3(256), 0, (hipStream_t)stream, t->gref, t->glist, t->gstat, t->lanes_used,
                       t->cfg.n_codes, max_n, logits_dev, list_dev);
    HIP_TRY(hipGetLastError());
    return PARROT_OK;
}
extern "C" int parrot_voc_check(parrot_voc_t* v, void* stream) { return v ? read_flag(v->err, (hipStream_t)stream, "vocoder") : PARROT_E_INVALID; }
extern "C" int parrot_tte_check(parrot_tte_t* t, void* stream) { return t ? read_flag(t->err, (hipStream_t)stream, "tte") : PARROT_E_INVALID; }
