// Issue rate of MFMA forms on one SIMD: one wave per SIMD (256 workgroups x 256 threads), independent accumulators
// pinned in AGPRs with empty asm so that the compiler cannot chain them, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    h4 a4 = {(_Float16)seed, 1, 2, 3}, b4 = {1, 2, 3, (_Float16)seed};
    h8 a8 = {1, 2, 3, 4, 5, 6, 7, (_Float16)seed}, b8 = a8;
    float af = seed, bf = 2.0f, s = 0;
    if (MODE == 3) {
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i]));
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f4){seed, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
                else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(acc[i]));
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* o; (void)hipMalloc(&o, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    const char* names[4] = {"v_mfma_f32_16x16x16_f16 (8 Kflop)", "v_mfma_f32_16x16x32_f16 (16 Kflop)", "v_mfma_f32_16x16x4_f32 (2 Kflop)",
                            "v_mfma_f32_32x32x16_f16 (32 Kflop)"};
    for (int m = 0; m < 4; ++m) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, o, iters, 1.0f);
            else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, o, iters, 1.0f);
            else if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, o, iters, 1.0f);
            else hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, o, iters, 1.0f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)iters * (m == 3 ? 4 : 16);
        printf("%s: %.2f ns per MFMA per SIMD\n", names[m], ms * 1e6 / n);
    }
    return 0;
}
