// Issue cost (shader clocks per wave-instruction, one wave per SIMD, 8 independent chains) of the instructions the operand
// conversion (write_p) is made of -- alone and beside a saturated MFMA stream of a second wave on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o build_exp/valu_rates && build_exp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Regs {
    float v[8], c[8];
    f32x2 p[8], q[8];
    unsigned u[8];
    u32x4 w[8];
};

template <int OP>
__device__ __forceinline__ void body(Regs& r, int lane) {
#define ONE(i)                                                                                                                    \
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r.v[i]) : "v"(r.c[i]));                                          \
    if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.q[i]));                                            \
    if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r.v[i]) : "v"(r.c[i]));                                              \
    if (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.u[i]) : "v"(r.v[i]), "v"(r.c[i]));                          \
    if (OP == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r.v[i]) : "v"(r.u[i]), "v"(r.c[i])); \
    if (OP == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r.u[i]), "+v"(r.u[(i + 1) & 7]));                           \
    if (OP == 6) asm volatile("ds_write_b128 %0, %1" ::"v"(lane * 16 + i * 1024), "v"(r.w[i]) : "memory");                       \
    if (OP == 7) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(4)" : "=v"(r.w[i]) : "v"(lane * 16 + i * 1024) : "memory"); \
    if (OP == 8) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r.v[i]) : "v"(r.u[i]));                                              \
    if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r.v[i]) : "v"(r.c[i]) : "vcc");                             \
    if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r.p[i]) : "v"(r.q[i]));                                      \
    if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r.v[i]) : "v"(r.c[i]));
    REP8(ONE)
#undef ONE
}

template <int OP>
__global__ __launch_bounds__(512) void rate(const f16x8* in, float* out, unsigned long long* cyc, int iters, int with_mfma) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f16x8 a = in[lane], b = in[64 + lane];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    Regs r;
    for (int i = 0; i < 8; ++i) {
        r.v[i] = out[threadIdx.x + i] + 1.0f + i;
        r.c[i] = 1.0001f + 0.001f * i;
        r.p[i] = f32x2{r.v[i], r.c[i]};
        r.q[i] = f32x2{1.0001f, 0.9999f};
        r.u[i] = 0x3c003c00u + i;
        r.w[i] = u32x4{r.u[i], r.u[i], r.u[i], r.u[i]};
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (wave >= 4) {
        for (int it = 0; it < iters; ++it) {
            body<OP>(r, lane); body<OP>(r, lane); body<OP>(r, lane); body<OP>(r, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (with_mfma == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
        }
    } else if (with_mfma == 2) {  // the same flops as 16x16x32 tiles: 16 instructions of 16 clocks
        f32x4v a4[8];
        for (int t = 0; t < 8; ++t) for (int q = 0; q < 4; ++q) a4[t][q] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g) a4[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, a4[g & 7], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) for (int q = 0; q < 4; ++q) acc[0][q] += a4[t][q];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += r.v[i] + r.c[i] + r.p[i][0] + r.p[i][1] + (float)r.u[i] + (float)r.w[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = c1 - c0;
}

template <int OP>
void run(const char* name, f16x8* in, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    unsigned long long hc[2];
    double r[3], m[3];
    for (int wm = 0; wm < 3; ++wm) {
        rate<OP><<<256, 512, 65536>>>(in, out, cyc, 100, wm);
        rate<OP><<<256, 512, 65536>>>(in, out, cyc, iters, wm);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
        r[wm] = (double)hc[1] / (iters * 32.0);
        m[wm] = (double)hc[0] / (iters * 8.0);
    }
    printf("{\"probe\": \"valu_rates\", \"instr\": \"%s\", \"cycles_alone\": %.2f, \"cycles_beside_32x32x16_wave\": %.2f, \"cycles_beside_16x16x32_wave\": %.2f, "
           "\"mfma_wave_cycles_per_32x32x16\": %.1f, \"mfma_wave_cycles_per_two_16x16x32\": %.1f}\n", name, r[0], r[1], r[2], m[1], m[2]);
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
    (void)hipFuncSetAttribute((const void*)rate<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)rate<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("v_fma_f32", in, out, cyc);
    run<11>("v_mul_f32", in, out, cyc);
    run<1>("v_pk_mul_f32", in, out, cyc);
    run<10>("v_pk_fma_f32", in, out, cyc);
    run<2>("v_max_f32", in, out, cyc);
    run<3>("v_cvt_pk_f16_f32", in, out, cyc);
    run<8>("v_cvt_f32_f16", in, out, cyc);
    run<4>("v_fma_mix_f32", in, out, cyc);
    run<5>("v_permlane32_swap_b32", in, out, cyc);
    run<9>("v_cndmask_b32", in, out, cyc);
    run<6>("ds_write_b128 (4 waves per CU)", in, out, cyc);
    run<7>("ds_read_b128 (4 waves per CU)", in, out, cyc);
    return 0;
}
