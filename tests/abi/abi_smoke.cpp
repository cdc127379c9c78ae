// Torch-free check of the C ABI (include/irm_hip.h): a compiled caller - what the reference side would be if
// it were not Python - allocates device memory with the HIP runtime, packs a 1x1-conv weight on the host
// exactly as the header describes, calls irm_ln_stats_f32 + irm_gemm1x1_f32 (LayerNorm prologue, bias,
// residual) on a HIP stream of its own and compares with a double-precision host evaluation of
// restormer.py:25-70,82-107,146-150; rejected arguments must come back as IRM_EINVAL.  Exit code 0 = ok.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "irm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(float lo, float hi) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return lo + (hi - lo) * (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.0f;
}

// wp[mtile][kstep][lane] = W[16 mtile + (lane & 15)][4 kstep + (lane >> 4)], zero padded (irm_hip.h)
static std::vector<float> pack_gemm(const std::vector<float>& w, int M, int K) {
    const int mt = (M + 15) / 16, ks = 4 * ((K + 15) / 16);
    std::vector<float> p((size_t)mt * ks * 64, 0.0f);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k)
            p[((size_t)(m / 16) * ks + k / 4) * 64 + (k % 4) * 16 + (m % 16)] = w[(size_t)m * K + k];
    return p;
}

template <class T>
static float* to_dev(const std::vector<T>& v) {
    float* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    if (irm_version() < 1) return 3;
    const int B = 2, C = 48, M = 144, H = 12, W = 20, N = H * W;
    std::vector<float> x((size_t)B * C * N), w((size_t)M * C), lnw(C), lnb(C), bias(M), res((size_t)B * M * N);
    for (auto& v : x) v = rnd(-2.f, 3.f);
    for (auto& v : w) v = rnd(-0.2f, 0.2f);
    for (auto& v : lnw) v = rnd(0.9f, 1.1f);
    for (auto& v : lnb) v = rnd(-0.1f, 0.1f);
    for (auto& v : bias) v = rnd(-0.3f, 0.3f);
    for (auto& v : res) v = rnd(-1.f, 1.f);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *dx = to_dev(x), *dwp = to_dev(pack_gemm(w, M, C)), *dlw = to_dev(lnw), *dlb = to_dev(lnb), *db = to_dev(bias),
          *dr = to_dev(res), *dy = nullptr, *dstats = nullptr;
    CK(hipMalloc(&dy, (size_t)B * M * N * 4));
    CK(hipMalloc(&dstats, (size_t)B * 2 * N * 4));
    if (!dx || !dwp || !dlw || !dlb || !db || !dr) return 2;
    int rc = irm_ln_stats_f32(dx, (long)C * N, dstats, B, C, N, 1e-5f, st);
    if (rc) { std::printf("irm_ln_stats_f32 -> %d\n", rc); return 4; }
    rc = irm_gemm1x1_f32(dwp, 0, dx, (long)C * N, dy, (long)M * N, dr, (long)M * N, db, dstats, dlw, dlb,
                         /*ln_mode WithBias*/ 1, /*act*/ 0, B, M, C, N, /*ct*/ 9, /*ygroups*/ 1, nullptr, 1e-5f, nullptr, st);
    if (rc) { std::printf("irm_gemm1x1_f32 -> %d\n", rc); return 4; }
    std::vector<float> y((size_t)B * M * N);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            double mu = 0, var = 0;
            for (int c = 0; c < C; ++c) mu += x[((size_t)b * C + c) * N + n];
            mu /= C;
            for (int c = 0; c < C; ++c) { const double d = x[((size_t)b * C + c) * N + n] - mu; var += d * d; }
            const double rstd = 1.0 / std::sqrt(var / C + 1e-5);
            for (int m = 0; m < M; ++m) {
                double acc = bias[m] + res[((size_t)b * M + m) * N + n];
                for (int c = 0; c < C; ++c)
                    acc += (double)w[(size_t)m * C + c] * ((x[((size_t)b * C + c) * N + n] - mu) * rstd * lnw[c] + lnb[c]);
                worst = std::fmax(worst, std::fabs(acc - y[((size_t)b * M + m) * N + n]));
            }
        }
    std::printf("LN + 1x1 conv + bias + residual through the C ABI: max-abs %.3e\n", worst);
    if (!(worst < 2e-4)) return 1;

    // rejected arguments come back as IRM_EINVAL (-1), nothing is launched
    if (irm_gemm1x1_f32(nullptr, 0, dx, (long)C * N, dy, (long)M * N, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
                        B, M, C, N, 9, 1, nullptr, 1e-5f, nullptr, st) != -1) return 5;
    if (irm_dwconv3x3_f32(dx, (long)C * N, nullptr, nullptr, dy, (long)C * N, B, C, H, W, 0, st) != -1) return 5;
    for (float* p : {dx, dwp, dlw, dlb, db, dr, dy, dstats}) (void)hipFree(p);
    (void)hipStreamDestroy(st);
    std::printf("abi_smoke ok (version %d)\n", irm_version());
    return 0;
}
