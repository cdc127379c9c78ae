// LDS bank-conflict micro-probe: one kernel per access pattern of the product kernels, 64 lanes,
// the pattern's ds instruction issued ITER times; run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -- ./lds_conflict
// conflict share of a pattern = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of its kernel (0 = conflict free).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/lds_conflict.hip -o tools/probes/lds_conflict
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITER 4096

template <int PAT>
__device__ unsigned addr_of(int lane) {
    const int i = lane & 15, g = lane >> 4;
    switch (PAT) {
        case 0: return lane * 16;                                             // lane-linear ds_read_b128 (weight fragments)
        case 1: return i * 128 + 16 * ((2 * g + (i >> 1)) & 7);               // Gram T = 6, round 2 rotation (row >> 1)
        case 2: { const int t[8] = {0, 2, 1, 3, 5, 7, 4, 6};                  // Gram T = 6, round 3 rotation
                  return i * 128 + 16 * ((2 * g + t[(i >> 1) & 7]) & 7); }
        case 3: return i * 256 + 16 * ((2 * g + i) & 15);                     // Gram T = 3, round 2 rotation (row)
        case 4: { const int t[16] = {0, 2, 4, 6, 1, 3, 5, 7, 9, 11, 13, 15, 8, 10, 12, 14};
                  return i * 256 + 16 * ((2 * g + t[i]) & 15); }              // Gram T = 3, round 3 rotation
        case 5: return i * 160 + 16 * g;                                      // fused branch kernels: stencil reads of the h image
        case 6: return 16 * g;                                                // fused branch kernels: tap reads (4 addresses per wave)
        case 7: return i * 160 + 16 * g;                                      // fused branch kernels: ds_write_b128 of the h image
        case 8: return (g * 128 + i) * 4;                                     // gemm_ring: ds_read_b32 of X rows 128 floats apart
        case 9: return (g * 128 + ((i + 16 * (g & 1)) & 127)) * 4;            // gemm_ring: odd rows shifted by 16 pixels (not built)
        case 10: return (i * 264 + 4 * g) * 4;                                // residual transpose read (row stride 264 floats)
    }
    return 0;
}

template <int PAT>
__global__ __launch_bounds__(64) void probe(float* out) {
    __shared__ __attribute__((aligned(16))) float sm[16 * 1024];
    for (int e = threadIdx.x; e < 16 * 1024; e += 64) sm[e] = (float)e;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)sm + addr_of<PAT>(threadIdx.x);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ITER; ++it) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (PAT == 7) asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(acc) : "memory");
        else if (PAT == 8 || PAT == 9) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v[0]) : "v"(a) : "memory");
        else asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        acc += v;
    }
    out[PAT * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    float* out;
    hipMalloc(&out, 16 * 64 * sizeof(float));
#define RUN(P) hipLaunchKernelGGL(probe<P>, dim3(1), dim3(64), 0, 0, out);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
