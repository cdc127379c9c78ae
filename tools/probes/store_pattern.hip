// Write-only bandwidth of two store shapes at the pin-GEMM size (512 planes of 6 x 512 x 512 floats):
//  A: GEMM epilogue shape - per wave-instruction 16 planes x 64 contiguous bytes (lane (g, r): plane r, 16 B at 4g)
//  B: one plane x 1 KiB contiguous per wave-instruction
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void store_a(float* y, long plane, int ctiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const long n0 = (long)blockIdx.x * 256 + wave * 64;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int c = 0; c < ctiles; ++c)
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(y + (long)(c * 16 + r) * plane + n0 + p * 16 + g * 4) = v;
}
__global__ __launch_bounds__(256) void store_b(float* y, long plane, int ctiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n0 = (long)blockIdx.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int c = 0; c < ctiles; ++c)
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(y + (long)(c * 16 + wave * 4 + q) * plane + n0 + lane * 4) = v;
}
int main() {
    const long plane = 6L * 512 * 512; const int ctiles = 32;      // 512 planes = 3.2 GB
    float* y; hipMalloc(&y, plane * 16 * ctiles * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 2; ++it) {
            if (mode == 0) hipLaunchKernelGGL(store_a, dim3(plane / 256), dim3(256), 0, 0, y, plane, ctiles);
            else hipLaunchKernelGGL(store_b, dim3(plane / 256), dim3(256), 0, 0, y, plane, ctiles);
        }
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) {
            if (mode == 0) hipLaunchKernelGGL(store_a, dim3(plane / 256), dim3(256), 0, 0, y, plane, ctiles);
            else hipLaunchKernelGGL(store_b, dim3(plane / 256), dim3(256), 0, 0, y, plane, ctiles);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per 3.2 GB -> %.2f TB/s\n", mode == 0 ? "A 16 planes x 64 B per instruction" : "B 1 KiB contiguous per instruction",
               ms / 5 * 1e3, plane * 16.0 * ctiles * 4 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
