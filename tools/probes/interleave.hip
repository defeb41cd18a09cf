// Can ONE wave overlap its own VALU work with its own MFMA stream on gfx950?  (coissue.hip asked the question for two waves
// of one SIMD; the fused MRF kernels need it inside a wave: the conversion of one branch's accumulators interleaved with
// the MFMAs of another.)  Every instruction is volatile inline asm, so the issue order is the source order:
//     per MFMA (v_mfma_f32_16x16x32_f16: 16 clocks of matrix pipe; or 32x32x16: 32)  +  NV instructions of the write_p mix
//     (v_pk_mul_f32 x2, v_max_f32 x2, v_cvt_pk_f16_f32, v_fma_mix_f32 x2, v_cvt_pk_f16_f32, cyclic) on registers the MFMAs
//     do not touch, optionally one ds_read_b128 / ds_write_b128 every RD / WR MFMAs.
// Prints shader clocks per MFMA for 1 and 2 waves per SIMD; "serial" would be 16 + NV x ~5, "overlapped" max(16, NV x ~5).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/interleave.hip -o build_exp/interleave && build_exp/interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ValuRegs {  // two independent register groups so that consecutive instructions of one kind do not chain
    f32x2 a[2], b[2], k;
    float x0[2], x1[2], m0[2], m1[2], r0[2], r1[2];
    unsigned h[2], l[2];
};
template <int I>
__device__ __forceinline__ void valu_one(ValuRegs& R) {
    constexpr int s = I % 8, q = (I / 8) % 2;
    if constexpr (s == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.b[q]) : "v"(R.a[q]), "v"(R.k));
    else if constexpr (s == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(R.a[q]) : "v"(R.b[q]), "v"(R.k));
    else if constexpr (s == 2) asm volatile("v_max_f32 %0, %1, %2" : "=v"(R.m0[q]) : "v"(R.x0[q]), "v"(R.x1[q]));
    else if constexpr (s == 3) asm volatile("v_max_f32 %0, %1, %2" : "=v"(R.m1[q]) : "v"(R.x1[q]), "v"(R.x0[q]));
    else if constexpr (s == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(R.h[q]) : "v"(R.m0[q]), "v"(R.m1[q]));
    else if constexpr (s == 5) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(R.r0[q]) : "v"(R.h[q]), "v"(R.m0[q]));
    else if constexpr (s == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(R.r1[q]) : "v"(R.h[q]), "v"(R.m1[q]));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(R.l[q]) : "v"(R.r0[q]), "v"(R.r1[q]));
}
template <int BASE, int N>
__device__ __forceinline__ void valu_n(ValuRegs& R) {
    if constexpr (N > 0) {
        valu_one<BASE>(R);
        valu_n<BASE + 1, N - 1>(R);
    }
}

// SHAPE 0: 16x16x32 (4 accumulators of f32x4), 1: 32x32x16 (2 accumulators of f32x16).  NV VALU per MFMA.  RD / WR: one LDS read /
// write per that many MFMAs (0 = none).  16 MFMAs per loop iteration.
template <int SHAPE, int NV, int RD, int WR>
__global__ __launch_bounds__(512) void probe(const f16x8* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    f16x8 a = in[lane], b = in[64 + lane];
    f32x4 c4[4];
    f32x16 c16[2];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) c4[t][r] = 0.f;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) c16[t][r] = 0.f;
    ValuRegs R;
    for (int q = 0; q < 2; ++q) {
        const float f = out[threadIdx.x + q] * 1e-3f + 1.f;
        R.a[q] = f32x2{f, f + 1.f}; R.b[q] = R.a[q];
        R.x0[q] = f; R.x1[q] = -f; R.m0[q] = f; R.m1[q] = f; R.r0[q] = 0.f; R.r1[q] = 0.f; R.h[q] = 0; R.l[q] = 0;
    }
    R.k = f32x2{1.f, 1.f};
    u32x4 ld = {0, 0, 0, 0};
    const unsigned laddr = (threadIdx.x * 16) & 65535;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[g & 3]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[g & 1]) : "v"(a), "v"(b));
            if (RD > 0 && g % (RD > 0 ? RD : 1) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(laddr));
            if (WR > 0 && g % (WR > 0 ? WR : 1) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(laddr), "v"(ld));
            if (g % 4 == 0) valu_n<0, NV>(R);
            else if (g % 4 == 1) valu_n<NV, NV>(R);
            else if (g % 4 == 2) valu_n<2 * NV, NV>(R);
            else valu_n<3 * NV, NV>(R);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += c4[t][r];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += c16[t][r];
    for (int q = 0; q < 2; ++q) s += R.a[q][0] + R.b[q][1] + R.m0[q] + R.m1[q] + R.r0[q] + R.r1[q] + (float)(R.h[q] + R.l[q]);
    s += (float)ld.x;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = c1 - c0;
}

template <int SHAPE, int NV, int RD, int WR>
static void run(const f16x8* in, float* out, unsigned long long* cyc, int blocks, int threads, const char* note) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<SHAPE, NV, RD, WR>), dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);  // warm
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SHAPE, NV, RD, WR>), dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / (iters * 16.0);
    // s_memtime ticks at a constant 100 MHz on gfx950: convert with the wall time of the launch when it looks like that
    printf("{\"probe\": \"interleave\", \"mfma\": \"%s\", \"valu_per_mfma\": %d, \"ds_read_every\": %d, \"ds_write_every\": %d, \"blocks\": %d, \"waves_per_simd\": %d, "
           "\"memtime_ticks_per_mfma\": %.3f, \"ns_per_mfma\": %.3f, \"note\": \"%s\"}\n",
           SHAPE == 0 ? "16x16x32" : "32x32x16", NV, RD, WR, blocks, threads / 256, per, ms * 1e6 / (iters * 16.0), note);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    f16x8* in;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&in, 128 * sizeof(f16x8));
    hipMalloc(&out, 1024 * 512 * sizeof(float) + 64);
    hipMalloc(&cyc, 8);
    {
        _Float16 h[1024];
        for (int i = 0; i < 1024; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.f);
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipMemset(out, 0, 1024 * 512 * sizeof(float) + 64);
    }
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512}) {
            const char* n = blocks == 1 ? "one CU" : "every CU (power-limited clock)";
            run<0, 0, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 1, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 2, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 3, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 4, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 6, 0, 0>(in, out, cyc, blocks, threads, n);
            run<0, 0, 1, 0>(in, out, cyc, blocks, threads, n);
            run<0, 2, 1, 0>(in, out, cyc, blocks, threads, n);
            run<0, 2, 2, 8>(in, out, cyc, blocks, threads, n);
            run<0, 3, 2, 8>(in, out, cyc, blocks, threads, n);
            run<1, 0, 0, 0>(in, out, cyc, blocks, threads, n);
            run<1, 2, 0, 0>(in, out, cyc, blocks, threads, n);
            run<1, 4, 0, 0>(in, out, cyc, blocks, threads, n);
            run<1, 6, 0, 0>(in, out, cyc, blocks, threads, n);
            run<1, 8, 0, 0>(in, out, cyc, blocks, threads, n);
            run<1, 4, 1, 8>(in, out, cyc, blocks, threads, n);
        }
    }
    return 0;
}
