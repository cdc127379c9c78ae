// Issue rate of the two fp16 MFMA forms on one SIMD (one wave per SIMD, 4 independent accumulators, back to back):
//   v_mfma_f32_16x16x16_f16 (CDNA1-3 form, K = 16) vs v_mfma_f32_16x16x32_f16 (gfx950, K = 32).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const _Float16 v = (_Float16)(threadIdx.x * 0.001f);
    h4 a4 = {v, v, v, v}, b4 = {v, (_Float16)1, v, (_Float16)1};
    h8 a8 = {v, v, v, v, v, v, v, v}, b8 = {v, (_Float16)1, v, (_Float16)1, v, (_Float16)1, v, (_Float16)1};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (MODE == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[k], 0, 0, 0);
                else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[k], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// fp16 SUBNORMAL inputs: A = 2^-16 everywhere (subnormal: the smallest normal fp16 is 2^-14), B = 1: D = K 2^-16 if honoured, 0 if flushed
__global__ void denorm_probe(float* out) {
    const _Float16 sub = (_Float16)1.52587890625e-05f, one = (_Float16)1.0f;
    h4 a4 = {sub, sub, sub, sub}, b4 = {one, one, one, one};
    h8 a8 = {sub, sub, sub, sub, sub, sub, sub, sub}, b8 = {one, one, one, one, one, one, one, one};
    f4 z = {0, 0, 0, 0};
    const f4 d16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, z, 0, 0, 0);
    const f4 d32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, z, 0, 0, 0);
    const f4 e32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, z, 0, 0, 0);        // subnormal on the B side
    if (threadIdx.x == 0) { out[0] = d16[0]; out[1] = d32[0]; out[2] = e32[0]; }
}

int main() {
    {
        float* o; hipMalloc(&o, 16);
        hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, o);
        float h[3]; hipMemcpy(h, o, 12, hipMemcpyDeviceToHost);
        printf("subnormal fp16 inputs (2^-16 x 1 summed over K): 16x16x16_f16 -> %g (expected %g), 16x16x32_f16 -> %g / %g (expected %g)\n",
               h[0], 16 * 1.52587890625e-05, h[1], h[2], 32 * 1.52587890625e-05);
    }

    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * 32;
            const double flop = n * (mode == 0 ? 8192.0 : 16384.0) * 4 * 256;      // 4 waves per CU (one per SIMD), 256 CUs
            printf("%s: %.1f counter ticks per MFMA, %.3f ms, %.0f TFLOP/s chip-wide\n", mode == 0 ? "16x16x16_f16" : "16x16x32_f16",
                   (double)c / n, ms, flop / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
