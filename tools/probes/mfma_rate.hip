// MFMA issue-rate probe (gfx950): how fast does one SIMD retire v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32
// with 4 independent accumulators, 1..3 waves per SIMD, with and without register copies between groups?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const s16x8* in, float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    s16x8 a[6], b[6], a2[6], b2[6];
    for (int i = 0; i < 6; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (i + 6)]; a2[i] = a[i]; b2[i] = b[i]; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {  // 48 register copies per group, like a_cur = a_nxt / b_cur = b_nxt
            for (int i = 0; i < 6; ++i) { a[i] = a2[i] + (short)1; b[i] = b2[i] + (short)1; }  // 48 packed VALU ops
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 96; ++g) {
                float av = __builtin_bit_cast(float, (int)a[g % 6][0]), bv = __builtin_bit_cast(float, (int)b[g % 6][1]);
                acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g & 3], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 24; ++g)
                acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(g >> 2) % 6]), __builtin_bit_cast(bf16x8, b[(g >> 2) % 3]), acc[g & 3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1) for (int i = 0; i < 6; ++i) { a2[i] = a[i]; b2[i] = b[i]; }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, s16x8* in, float* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<grid, 256>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<grid, 256>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * 4 * iters * (MODE == 2 ? 96 : 24);
    const double flop = n_mfma * (MODE == 2 ? 4096.0 : 32768.0);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", name, blocks_per_cu, ms, flop / ms / 1e9, ms * 1e6 / (n_mfma / 1024));
}

int main() {
    s16x8* in; float* out;
    hipMalloc(&in, 64 * 12 * sizeof(s16x8));
    {   // random bf16 values in (-2, 2): realistic toggling -> realistic power / clock (zero or constant data overstates it)
        unsigned short h[64 * 12 * 8];
        unsigned st = 12345u;
        for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (unsigned short)(((st >> 16) & 0x807f) | (0x3f00 + ((st >> 9) & 0x80))); }
        hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    for (int w = 1; w <= 3; ++w) {
        run<0>("bf16 32x32x16, 4 acc", w, in, out);
        run<1>("bf16 32x32x16, 4 acc + movs", w, in, out);
        run<2>("f32 32x32x2, 4 acc", w, in, out);
    }
    return 0;
}
