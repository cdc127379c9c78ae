// Micro-probe: do the matrix, vector and LDS pipes of a CU overlap better when the two waves of a SIMD run DIFFERENT
// roles (waves 0-3: MFMA stream, waves 4-7: LDS reads + FMAs) than when all 8 waves run the same mixed stream?
// One 512-thread workgroup per CU, ITER barrier-separated iterations; per iteration and CU the same total work in both
// variants: 8 x 72 MFMA (16x16x32 f16), 8 x 312 v_fma_f32, 8 x 64 ds_read_b128.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/role_split.hip -o tools/probes/role_split
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define ITER 400

template <int NM>
__device__ __forceinline__ void mfma_block(f4 (&acc)[6], const h8& a, const h8& b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i % 6], 0, 0, 0);
}
// NR ds_read_b128, each feeding NF / NR fmas
template <int NR, int NF>
__device__ __forceinline__ void stencil_block(const float* sm, unsigned base, f4 (&o)[4], f4 k) {
    constexpr int PER = NF / NR / 4;        // packed groups of 4 fmas per read
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const f4 p = *reinterpret_cast<const f4*>(sm + ((base + i * 160) & 8191));
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            f4& t = o[(i + e) & 3];
            t[0] = fmaf(p[0], k[e & 3], t[0]); t[1] = fmaf(p[1], k[e & 3], t[1]);
            t[2] = fmaf(p[2], k[e & 3], t[2]); t[3] = fmaf(p[3], k[e & 3], t[3]);
        }
    }
}

template <int MODE>   // 0: all waves mixed (interleaved by the compiler), 1: roles by wave >= 4, 2: roles by parity
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float sm[8192 + 64];
    for (int e = threadIdx.x; e < 8192 + 64; e += 512) sm[e] = (float)(e & 7) * 0.125f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f4 acc[6], o[4];
    for (int i = 0; i < 6; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) o[i] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * lane); b[i] = (_Float16)(0.002f * i); }
    const f4 k = {0.5f, 0.25f, 0.125f, 0.0625f};
    const unsigned base = (lane & 15) * 40 + (lane >> 4) * 4 + wave * 16;
    const bool role_m = MODE == 1 ? wave < 4 : (wave & 1) == 0;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
            // the mixed stream of the product kernel: 8 chunks of 9 MFMAs + 8 reads + 39 fmas
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                stencil_block<8, 32>(sm, base + c * 8, o, k);
                mfma_block<9>(acc, a, b);
#pragma unroll
                for (int q = 0; q < 9; ++q) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0); }
            }
        } else if (MODE == 3) {                  // matrix work only (every wave)
            mfma_block<72>(acc, a, b);
        } else if (MODE == 4) {                  // LDS reads + FMAs only (every wave)
#pragma unroll
            for (int c = 0; c < 8; ++c) stencil_block<8, 32>(sm, base + c * 8, o, k);
        } else if (MODE == 5) {                  // both, not interleaved by sched_group_barrier
#pragma unroll
            for (int c = 0; c < 8; ++c) { stencil_block<8, 32>(sm, base + c * 8, o, k); mfma_block<9>(acc, a, b); }
        } else if (MODE == 6) {                  // FMAs on registers only (no LDS) + MFMAs, interleaved
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f4& t = o[e & 3];
                    t[0] = fmaf(t[1], k[e & 3], t[0]); t[1] = fmaf(t[2], k[e & 3], t[1]);
                    t[2] = fmaf(t[3], k[e & 3], t[2]); t[3] = fmaf(t[0], k[e & 3], t[3]);
                }
                mfma_block<9>(acc, a, b);
#pragma unroll
                for (int q = 0; q < 9; ++q) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0); }
            }
        } else if (MODE == 7) {                  // FMAs on registers only, no MFMAs
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f4& t = o[e & 3];
                    t[0] = fmaf(t[1], k[e & 3], t[0]); t[1] = fmaf(t[2], k[e & 3], t[1]);
                    t[2] = fmaf(t[3], k[e & 3], t[2]); t[3] = fmaf(t[0], k[e & 3], t[3]);
                }
            }
        } else if (MODE == 8 || MODE == 9) {     // roles without LDS: waves 0-3 (8) / even waves (9): 144 MFMA, the others: 512 register FMAs
            if (MODE == 8 ? wave < 4 : (wave & 1) == 0) {
                mfma_block<144>(acc, a, b);
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f4& t = o[e & 3];
                        t[0] = fmaf(t[1], k[e & 3], t[0]); t[1] = fmaf(t[2], k[e & 3], t[1]);
                        t[2] = fmaf(t[3], k[e & 3], t[2]); t[3] = fmaf(t[0], k[e & 3], t[3]);
                    }
                }
            }
        } else if (role_m) {
            mfma_block<144>(acc, a, b);
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) stencil_block<8, 32>(sm, base + c * 8, o, k);
        }
        __syncthreads();
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += o[i][0] + o[i][1] + o[i][2] + o[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    long long h[256];
    const char* names[10] = {"all waves mixed (sched_group_barrier 1 MFMA : 3 VALU)", "roles: waves 0-3 MFMA, 4-7 LDS+FMA", "roles by wave parity",
                            "72 MFMA only", "64 ds_read_b128 + 256 FMA only", "mixed, compiler order", "72 MFMA + 256 register FMAs interleaved", "256 register FMAs only", "roles, no LDS: waves 0-3 144 MFMA, waves 4-7 512 register FMAs", "roles, no LDS, by parity"};
    for (int mode = 0; mode < 10; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 7) hipLaunchKernelGGL(probe<7>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(256), dim3(512), 0, 0, out, cyc);
            if (mode == 9) hipLaunchKernelGGL(probe<9>, dim3(256), dim3(512), 0, 0, out, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("mode %d (%s): %.0f clock64 ticks per iteration\n", mode, names[mode], s / 256 / ITER);
    }
    return 0;
}
