// Two questions about one SIMD of gfx950, answered with s_memtime inside the kernel (cycles are shader clocks):
//  (1) how many independent accumulators does a back-to-back v_mfma_f32_32x32x16_f16 stream of ONE wave need to run at the
//      pipe's rate (dependency distance 1 .. 4)?
//  (2) what does a VALU-only wave cost / get when it shares the SIMD with an MFMA-only wave (the anti-phase experiment of
//      resblock_dual.h): MFMA wave alone, VALU wave alone, both together -- per-instruction cycles of each.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coissue.hip -o build_exp/coissue && build_exp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_chain(const f16x8* in, float* out, unsigned long long* cyc, int iters) {
    f16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 12; ++g) acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = c1 - c0;
}

// waves 0-3: role A, waves 4-7: role B (wave i and i + 4 share a SIMD).  role bits: 1 = MFMA stream, 2 = VALU stream (fma chain x8),
// 4 = VALU stream made of the write_p mix (pk_mul, max, cvt_pk, fma_mix)
__global__ __launch_bounds__(512) void coissue(const f16x8* in, float* out, unsigned long long* cyc, int iters, int roleA, int roleB) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    f16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = out[threadIdx.x + i];
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 12; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 12; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);  // 96 VALU per iteration
        }
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 12; ++g) {  // per pair: 2 pk_mul, 2 max, cvt_pk, 2 fma_mix, cvt_pk = 8 VALU; 4 pairs -> 32 per g... x3 = 96 per iteration of 4 g's
                if (g % 4 != 0) continue;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
#pragma unroll
                    for (int rep = 0; rep < 3; ++rep) {
                        f32x2 vv = {v[i], v[i + 1]};
                        f32x2 x = vv * 1.25f, y = vv * 0.125f;
                        float v0, v1;
                        asm volatile("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(x[0]), "v"(y[0]));
                        asm volatile("v_max_f32 %0, %1, %2" : "=v"(v1) : "v"(x[1]), "v"(y[1]));
                        unsigned hi, lo;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
                        float r0, r1;
                        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
                        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
                        v[i] = __builtin_bit_cast(float, hi ^ 0x3c003c00u) * 0.5f + r0;
                        v[i + 1] = __builtin_bit_cast(float, lo) + r1 + 1.0f;
                    }
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = c1 - c0;
}

int main() {
    f16x8* in; float* out; unsigned long long* cyc;
    (void)hipMalloc(&in, 128 * sizeof(f16x8));
    (void)hipMalloc(&out, 256 * 4 * 1024 * sizeof(float));
    (void)hipMalloc(&cyc, 2 * sizeof(unsigned long long));
    static unsigned short h[128 * 8];
    unsigned st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (unsigned short)(((st >> 16) & 0x83ff) | (0x3800 + ((st >> 9) & 0x400))); }
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMemset(out, 0, 256 * 4 * 1024 * sizeof(float));
    const int iters = 4000;
    unsigned long long hc[2];
    auto chain = [&](auto kern, int nacc) {
        kern<<<256, 256>>>(in, out, cyc, 200);
        kern<<<256, 256>>>(in, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hc, cyc, sizeof(unsigned long long), hipMemcpyDeviceToHost);
        printf("{\"probe\": \"mfma_chain\", \"independent_accumulators\": %d, \"cycles_per_mfma_32x32x16_f16\": %.1f}\n", nacc, (double)hc[0] / (iters * 12.0));
    };
    chain(mfma_chain<1>, 1); chain(mfma_chain<2>, 2); chain(mfma_chain<3>, 3); chain(mfma_chain<4>, 4);
    const char* names[5] = {"idle", "mfma", "valu_fma", "", "valu_split_mix"};
    for (int ra : {1, 0}) for (int rb : {0, 2, 4}) {
        if (ra == 0 && rb == 0) continue;
        coissue<<<256, 512>>>(in, out, cyc, 200, ra, rb);
        coissue<<<256, 512>>>(in, out, cyc, iters, ra, rb);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
        printf("{\"probe\": \"coissue\", \"wave_a\": \"%s\", \"wave_b\": \"%s\", \"a_cycles_per_mfma\": %.1f, \"b_cycles_per_valu\": %.2f}\n", names[ra], names[rb],
               ra ? (double)hc[0] / (iters * 12.0) : 0.0, rb ? (double)hc[1] / (iters * 96.0) : 0.0);
    }
    return 0;
}
