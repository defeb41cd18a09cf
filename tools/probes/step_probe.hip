// Rebuilds one main-loop step of conv_bf16x6_kernel in isolation (gfx950): 24 v_mfma_f32_32x32x16_bf16 with the real
// operand pattern (2 weight tiles x 3 pieces, 2 activation tiles x 3 pieces, 4 accumulators), optionally with the
// step's 6 global_load_dwordx4 (weights, L2 resident) and 6 ds_read_b128 (activations), ping-pong register sets.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/step_probe.hip -o tools/probes/step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool GL, bool DS, bool INTERLEAVE>
__global__ __launch_bounds__(256, 2) void probe(const char* __restrict__ w, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36864 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(w)[i];
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    s16x8 a0[2][3], b0[2][3], a1[2][3], b1[2][3];
    const char* ap = w + (size_t)(wave >> 1) * 2 * 3072 * 96 + lane * 16;   // per wave-row weight stream (96 steps, wraps)
    const char* bp = lds + ((wave & 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16;
    auto ld = [&](s16x8 (&a)[2][3], s16x8 (&b)[2][3], int st) {
        for (int m = 0; m < 2; ++m) for (int p = 0; p < 3; ++p) {
            if (GL) a[m][p] = *reinterpret_cast<const s16x8*>(ap + ((size_t)m * 96 + (st % 96)) * 3072 + p * 1024);
        }
        for (int n = 0; n < 2; ++n) for (int p = 0; p < 3; ++p) {
            if (DS) b[n][p] = *reinterpret_cast<const s16x8*>(bp + p * 6144 + n * 1024 + (st % 11) * 5 * 32);
        }
    };
    auto mm = [&](const s16x8 (&a)[2][3], const s16x8 (&b)[2][3]) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m][PA[t]]), __builtin_bit_cast(bf16x8, b[n][PB[t]]), acc[m][n], 0, 0, 0);
    };
    for (int m = 0; m < 2; ++m) for (int p = 0; p < 3; ++p) {
        a0[m][p] = *reinterpret_cast<const s16x8*>(ap + (size_t)m * 96 * 3072 + p * 1024);
        a1[m][p] = a0[m][p];
        b0[m][p] = *reinterpret_cast<const s16x8*>(bp + p * 6144 + m * 1024);
        b1[m][p] = b0[m][p];
    }
    auto sched = [&]() {
        if (INTERLEAVE) {
#pragma unroll
            for (int g = 0; g < 6; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); if (GL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
            for (int g = 0; g < 6; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); if (DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int st = 0; st < steps; st += 2) {
        ld(a1, b1, st + 1);
        if (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0);
        sched();
        ld(a0, b0, st + 2);
        if (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
        sched();
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool GL, bool DS, bool IL>
void run(const char* name, int bpc, const char* w, float* out) {
    const int steps = 4000, grid = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<GL, DS, IL>), dim3(grid), dim3(256), 36864, 0, w, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<GL, DS, IL>), dim3(grid), dim3(256), 36864, 0, w, out, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * 4 * steps * 24;
    printf("%-44s wg/CU=%d  %8.3f ms  %7.1f TFLOP/s bf16 = %6.1f TF fp32-equivalent (/6)   %5.1f ns/MFMA/SIMD\n", name, bpc, ms,
           n_mfma * 32768 / ms / 1e9, n_mfma * 32768 / ms / 1e9 / 6, ms * 1e6 / (n_mfma / 1024));
}


// Weights through LDS: each step's 12 KB (4 m-tiles x 3 pieces x 1 KiB) are fetched ONCE per workgroup (3 x 16 B per
// thread), written to a 2-slot LDS ring, published by the per-step barrier; every wave then reads its fragments with
// ds_read_b128.  Halves (128x128) / quarters (64x256) the vector-memory traffic of the per-wave global stream.
template <bool PREFETCH_B>
__global__ __launch_bounds__(256, 2) void probe_lds(const char* __restrict__ w, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [0,36864): activation slab, then 2 x 12288 weight ring
    char* ring = lds + 36864;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 36864 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(w)[i];
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    s16x8 a[2][3], b0[2][3], b1[2][3];
    const char* bp = lds + ((wave & 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16;
    // cooperative fetch: thread -> 3 chunks of 16 B of the step's 12 KB; chunk q = tid + 256 i -> m-tile q/192
    uint4 stg[3];
    auto gfetch = [&](int st) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + 256 * i, mt = q / 192, within = q - mt * 192;
            stg[i] = *reinterpret_cast<const uint4*>(w + ((size_t)mt * 96 + (st % 96)) * 3072 + within * 16);
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4*>(ring + slot * 12288 + (tid + 256 * i) * 16) = stg[i];
    };
    auto aread = [&](int slot) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[m][p] = *reinterpret_cast<const s16x8*>(ring + slot * 12288 + ((wave >> 1) * 2 + m) * 3072 + p * 1024 + lane * 16);
    };
    auto bread = [&](s16x8 (&b)[2][3], int st) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[n][p] = *reinterpret_cast<const s16x8*>(bp + p * 6144 + n * 1024 + (st % 11) * 5 * 32);
    };
    auto mm = [&](const s16x8 (&b)[2][3]) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m][PA[t]]), __builtin_bit_cast(bf16x8, b[n][PB[t]]), acc[m][n], 0, 0, 0);
    };
    gfetch(0);
    lstore(0);
    bread(b0, 0);
    auto step = [&](s16x8 (&bc)[2][3], s16x8 (&bn)[2][3], int st) {
        __syncthreads();              // slot st&1 (and a new slab, at chunk starts) is visible; slot (st+1)&1 is free
        aread(st & 1);
        gfetch(st + 1);
        if (PREFETCH_B) bread(bn, st + 1); else bread(bc, st);
        __builtin_amdgcn_sched_barrier(0);
        mm(bc);
        __builtin_amdgcn_sched_barrier(0);
        lstore((st + 1) & 1);
    };
    for (int st = 0; st < steps; st += 2) {
        if (PREFETCH_B) { step(b0, b1, st); step(b1, b0, st + 1); }
        else { step(b0, b0, st); step(b0, b0, st + 1); }
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <bool PB>
void run_lds(const char* name, int bpc, const char* w, float* out) {
    const int steps = 4000, grid = 256 * bpc;
    const int lds = 36864 + 2 * 12288;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_lds<PB>), dim3(grid), dim3(256), lds, 0, w, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe_lds<PB>), dim3(grid), dim3(256), lds, 0, w, out, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * 4 * steps * 24;
    printf("%-44s wg/CU=%d  %8.3f ms  %7.1f TFLOP/s bf16 = %6.1f TF fp32-equivalent (/6)   %5.1f ns/MFMA/SIMD\n", name, bpc, ms,
           n_mfma * 32768 / ms / 1e9, n_mfma * 32768 / ms / 1e9 / 6, ms * 1e6 / (n_mfma / 1024));
}

int main() {
    const size_t wbytes = (size_t)2 * 2 * 96 * 3072 + 65536;
    std::vector<unsigned short> h(wbytes / 2);
    unsigned st = 777u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (unsigned short)(((st >> 16) & 0x807f) | (0x3f00 + ((st >> 9) & 0x80))); }
    char* w; float* out;
    (void)hipMalloc(&w, wbytes); (void)hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<false, false, false>("MFMA only (kernel operand pattern)", bpc, w, out);
        run<true, false, false>("+ 6 global_load_dwordx4 / step, up front", bpc, w, out);
        run<false, true, false>("+ 6 ds_read_b128 / step, up front", bpc, w, out);
        run<true, true, false>("+ both, up front", bpc, w, out);
        run<true, true, true>("+ both, interleaved with MFMA pairs", bpc, w, out);
        run_lds<false>("weights via LDS ring, barrier per step", bpc, w, out);
        run_lds<true>("weights via LDS ring + B prefetch", bpc, w, out);
    }
    return 0;
}
