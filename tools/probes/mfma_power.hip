// Sustained MFMA throughput under the power limit (gfx950): fp16 / bf16, 32x32x16 vs 16x16x32, RANDOM and CONSTANT operands
// in the same run, long launches (DVFS settles), and a duty-cycle sweep (idle gaps between MFMA bursts): is throughput set
// by the energy per MFMA rather than by the issue rate?  One JSON line per measurement (committed under profiles/):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o build_exp/mfma_power && build_exp/mfma_power
// `eff_clock_ghz` = shader cycles (s_memtime) a wave spent inside the kernel / the kernel's HIP-event duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: f16 32x32x16   1: bf16 32x32x16   2: f16 16x16x32   3: bf16 16x16x32;  SLEEP: s_sleep units after each 24-MFMA burst
template <int MODE>
__global__ __launch_bounds__(256) void probe(const s16x8* in, float* out, unsigned long long* cyc, int iters, int sleep) {
    s16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = in[threadIdx.x % 64 + 64 * i]; b[i] = in[threadIdx.x % 64 + 64 * (i + 6)]; }
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (MODE < 2) {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                if (MODE == 0) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(g >> 2) % 6]), __builtin_bit_cast(f16x8, b[g % 6]), acc[g & 3], 0, 0, 0);
                else acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(g >> 2) % 6]), __builtin_bit_cast(bf16x8, b[g % 6]), acc[g & 3], 0, 0, 0);
            }
            if (sleep) __builtin_amdgcn_s_sleep(1), __builtin_amdgcn_sched_barrier(0);
            for (int q = 1; q < sleep; ++q) __builtin_amdgcn_s_sleep(1);
        }
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                if (MODE == 2) acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(g >> 3) % 6]), __builtin_bit_cast(f16x8, b[g % 6]), acc[g & 7], 0, 0, 0);
                else acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(g >> 3) % 6]), __builtin_bit_cast(bf16x8, b[g % 6]), acc[g & 7], 0, 0, 0);
            }
            for (int q = 0; q < sleep; ++q) __builtin_amdgcn_s_sleep(1);
        }
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = c1 - c0;
}

static const char* g_data = "random";

template <int MODE>
void run(const char* name, int blocks_per_cu, s16x8* in, float* out, unsigned long long* cyc, int sleep) {
    const int iters = 40000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<grid, 256>>>(in, out, cyc, 2000, sleep);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<MODE><<<grid, 256>>>(in, out, cyc, iters, sleep);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc = 0;
    (void)hipMemcpy(&hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 24 * 32768.0;  // (48 x 16384 for the 16x16x32 shapes: the same)
    printf("{\"probe\": \"mfma_power\", \"data\": \"%s\", \"mfma\": \"%s\", \"waves_per_simd\": %d, \"idle_s_sleep_per_burst\": %d, \"ms\": %.2f, "
           "\"tflops\": %.1f, \"eff_clock_ghz\": %.3f, \"mfma_per_burst\": %d}\n",
           g_data, name, blocks_per_cu, sleep, ms, flop / ms / 1e9, (double)hc / (ms * 1e6), MODE < 2 ? 24 : 48);
    fflush(stdout);
}

int main() {
    s16x8* in; float* out; unsigned long long* cyc;
    (void)hipMalloc(&in, 64 * 12 * sizeof(s16x8));
    (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    (void)hipMalloc(&cyc, sizeof(unsigned long long));
    for (int constant = 0; constant < 2; ++constant) {
        g_data = constant ? "constant" : "random";
        // random half / bf16 bit patterns with exponents near 1.0 (realistic toggling), or one constant (1.0)
        static unsigned short h[64 * 12 * 8];
        unsigned st = 12345u;
        for (auto& v : h) { st = st * 1664525u + 1013904223u; v = constant ? 0x3c00 : (unsigned short)(((st >> 16) & 0x83ff) | (0x3800 + ((st >> 9) & 0x400))); }
        (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
        for (int w = 1; w <= 2; ++w) {
            run<0>("f16 32x32x16", w, in, out, cyc, 0);
            run<1>("bf16 32x32x16", w, in, out, cyc, 0);
            run<2>("f16 16x16x32", w, in, out, cyc, 0);
            run<3>("bf16 16x16x32", w, in, out, cyc, 0);
        }
        for (int sl : {2, 4, 8, 16}) {
            run<0>("f16 32x32x16", 2, in, out, cyc, sl);
            run<2>("f16 16x16x32", 2, in, out, cyc, sl);
        }
    }
    return 0;
}
