// Pointwise (1x1 conv) GEMM on the exact-f32 MFMA, planar NCHW activations.
//
//   Y[b][co][n] = epilogue( sum_k W[co][k] * prologue(X[b][k][n]) )
//
// Replaces the reference's bias-free 1x1 nn.Conv2d projections
// (src/restormer/restormer.py:82,86,105,107,223,228,240) together with the
// LayerNorm that feeds them (restormer.py:25-70, as a prologue on the X tile)
// and the residual add that follows them (restormer.py:146-150, epilogue).
//
// MI355X mapping
//  * MFMA v_mfma_f32_16x16x4_f32, pixels on the MFMA row index so every lane
//    ends up with 4 consecutive pixels of one output channel -> 16-byte stores.
//  * one 256-thread workgroup = 64*PT pixels; each of the 4 waves owns 16*PT
//    pixels and CT output-channel tiles (16 channels each) per pass; X rows are
//    streamed through a double-buffered LDS tile 16 input channels at a time,
//    so X is read from HBM once when all output channels fit one pass and from
//    L2 on further passes.
//  * weights are pre-packed on the host in MFMA B-operand order
//    Wp[mtile][kstep][lane] = W[16*mtile + (lane&15)][4*kstep + (lane>>4)]
//    (zero padded; ksteps padded to a multiple of 4) and staged through LDS
//    with plain 16-byte copies.
#include "irm_common.h"

struct GemmArgs {
    const float* Wp; long w_bs;   // packed weights, per-batch stride (0 = shared)
    const float* X;  long x_bs;   // [B][K][N]
    float* Y;        long y_bs;   // [B][M][N]
    const float* R;  long r_bs;   // residual [B][M][N] or null
    const float* bias;            // [M] or null
    const float* stats;           // [B][2][N] mean, rstd (LN prologue) or null
    const float* lnw;             // [K]
    const float* lnb;             // [K] (WithBias only)
    int M, K, N;
    int mtiles, ksteps;           // ceil(M/16), 4*ceil(K/16)
    int ln_mode, act;
};

template <int PT, int CT, bool VEC>
__global__ __launch_bounds__(256) void gemm_pw_kernel(GemmArgs a) {
    constexpr int BN = 64 * PT;      // pixels per workgroup
    constexpr int BNP = BN + 16;     // row stride: rows k and k+1 land on disjoint bank halves
    constexpr int BK = 16;           // input channels per stage
    constexpr int XV = BN / 4;       // float4 per X row
    __shared__ __attribute__((aligned(16))) float xs[2][BK * BNP];
    __shared__ __attribute__((aligned(16))) float ws[2][CT * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;

    const float* X = a.X + (long)b * a.x_bs;
    const float* Wp = a.Wp + (long)b * a.w_bs;
    float* Y = a.Y + (long)b * a.y_bs;
    const float* R = a.R ? a.R + (long)b * a.r_bs : nullptr;

    // loader geometry: thread owns float4 column c4 of rows r0, r0 + 256/XV, ...
    const int c4 = tid % XV;
    const int r0 = tid / XV;
    constexpr int RSTEP = 256 / XV;
    const int ncol = n0 + c4 * 4;
    const bool col_ok = ncol < a.N;

    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.ln_mode != IRM_LN_NONE && col_ok) {
        const float* st = a.stats + (long)b * 2 * a.N;
        mu = irm_ld4<VEC>(st, ncol, a.N);
        rs = irm_ld4<VEC>(st + a.N, ncol, a.N);
    }

    const int nstages = a.ksteps / 4;
    const int nchunks = (a.mtiles + CT - 1) / CT;

    float4 xr[PT];
    float4 wr[(CT * 64 + 255) / 256];

    auto load_stage = [&](int s, int mt0) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int k = s * BK + r0 + j * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && k < a.K) {
                v = irm_ld4<VEC>(X + (long)k * a.N, ncol, a.N);
                if (a.ln_mode == IRM_LN_WITHBIAS) {
                    const float w = a.lnw[k], bb = a.lnb[k];
                    v.x = (v.x - mu.x) * rs.x * w + bb;
                    v.y = (v.y - mu.y) * rs.y * w + bb;
                    v.z = (v.z - mu.z) * rs.z * w + bb;
                    v.w = (v.w - mu.w) * rs.w * w + bb;
                } else if (a.ln_mode == IRM_LN_BIASFREE) {
                    const float w = a.lnw[k];
                    v.x = v.x * rs.x * w;
                    v.y = v.y * rs.y * w;
                    v.z = v.z * rs.z * w;
                    v.w = v.w * rs.w * w;
                }
            }
            xr[j] = v;
        }
#pragma unroll
        for (int j = 0; j < (CT * 64 + 255) / 256; ++j) {
            const int f = tid + j * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < CT * 64) {
                const int ct = f >> 6, q = f & 63;
                const int mt = mt0 + ct;
                if (mt < a.mtiles)
                    v = *reinterpret_cast<const float4*>(Wp + ((long)mt * a.ksteps + s * 4) * 64 + q * 4);
            }
            wr[j] = v;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int r = r0 + j * RSTEP;
            *reinterpret_cast<float4*>(&xs[buf][r * BNP + c4 * 4]) = xr[j];
        }
#pragma unroll
        for (int j = 0; j < (CT * 64 + 255) / 256; ++j) {
            const int f = tid + j * 256;
            if (f < CT * 64) *reinterpret_cast<float4*>(&ws[buf][f * 4]) = wr[j];
        }
    };

    const int arow = lane >> 4;                       // k within a k-step
    const int acol = wave * 16 * PT + (lane & 15);    // pixel within the tile

    for (int chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        const int mt0 = chunk * CT;
        const int nct = min(CT, a.mtiles - mt0);
        f32x4 acc[PT][CT];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        load_stage(0, mt0);
        store_stage(0);
        __syncthreads();
        for (int s = 0; s < nstages; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstages) load_stage(s + 1, mt0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float af[PT], bf[CT];
#pragma unroll
                for (int p = 0; p < PT; ++p) af[p] = xs[buf][(kk * 4 + arow) * BNP + acol + p * 16];
#pragma unroll
                for (int c = 0; c < CT; ++c) bf[c] = ws[buf][(c * 4 + kk) * 64 + lane];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (c < nct) {
#pragma unroll
                        for (int p = 0; p < PT; ++p) acc[p][c] = irm_mfma16(af[p], bf[c], acc[p][c]);
                    }
                }
            }
            if (s + 1 < nstages) store_stage(buf ^ 1);
            __syncthreads();
        }

        // epilogue: lane holds pixels pix..pix+3 of channel co for every (p, c)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int co = (mt0 + c) * 16 + (lane & 15);
            if (c < nct && co < a.M) {
                const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const int pix = n0 + wave * 16 * PT + p * 16 + (lane >> 4) * 4;
                    if (pix < a.N) {
                        float4 v = make_float4(acc[p][c][0] + bv, acc[p][c][1] + bv, acc[p][c][2] + bv,
                                               acc[p][c][3] + bv);
                        if (a.act != IRM_ACT_NONE) {
                            v.x = irm_act(v.x, a.act); v.y = irm_act(v.y, a.act);
                            v.z = irm_act(v.z, a.act); v.w = irm_act(v.w, a.act);
                        }
                        const long off = (long)co * a.N;
                        if (R) {
                            const float4 r = irm_ld4<VEC>(R + off, pix, a.N);
                            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                        }
                        irm_st4<VEC>(Y + off, pix, a.N, v);
                    }
                }
            }
        }
    }
}

template <int PT, int CT>
static int launch_gemm(const GemmArgs& a, int B, int ygroups, bool vec, hipStream_t stream) {
    constexpr int BN = 64 * PT;
    dim3 grid((a.N + BN - 1) / BN, ygroups, B);
    if (vec) hipLaunchKernelGGL((gemm_pw_kernel<PT, CT, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gemm_pw_kernel<PT, CT, false>), grid, dim3(256), 0, stream, a);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
// C ABI (declared in include/irm_hip.h)
extern "C" int irm_gemm1x1_f32(const float* wp, long w_bs, const float* x, long x_bs, float* y, long y_bs,
                               const float* res, long r_bs, const float* bias, const float* stats,
                               const float* lnw, const float* lnb, int ln_mode, int act, int B, int M, int K,
                               int N, int ct, int ygroups, hipStream_t stream) {
    if (!wp || !x || !y || B <= 0 || M <= 0 || K <= 0 || N <= 0) return IRM_EINVAL;
    if (ln_mode != IRM_LN_NONE && (!stats || !lnw || (ln_mode == IRM_LN_WITHBIAS && !lnb))) return IRM_EINVAL;
    if (ln_mode < 0 || ln_mode > 2 || act < 0 || act > 3) return IRM_EINVAL;
    if ((w_bs & 3) || !irm_aligned16(wp)) return IRM_EINVAL;
    // 16-byte fast path needs every row of every operand 16-byte aligned
    const bool vec = !(N & 3) && !(x_bs & 3) && !(y_bs & 3) && !(r_bs & 3) && irm_aligned16(x) &&
                     irm_aligned16(y) && irm_aligned16(res) && irm_aligned16(stats);
    GemmArgs a;
    a.Wp = wp; a.w_bs = w_bs; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.R = res; a.r_bs = r_bs;
    a.bias = bias; a.stats = stats; a.lnw = lnw; a.lnb = lnb;
    a.M = M; a.K = K; a.N = N; a.mtiles = (M + 15) / 16; a.ksteps = 4 * ((K + 15) / 16);
    a.ln_mode = ln_mode; a.act = act;
    const int nchunks = (a.mtiles + (ct > 0 ? ct : 1) - 1) / (ct > 0 ? ct : 1);
    if (ygroups <= 0) ygroups = 1;
    if (ygroups > nchunks) ygroups = nchunks;
    if (B > 65535 || ygroups > 65535) return IRM_EINVAL;
    switch (ct) {
        case 3: return launch_gemm<2, 3>(a, B, ygroups, vec, stream);
        case 4: return launch_gemm<2, 4>(a, B, ygroups, vec, stream);
        case 6: return launch_gemm<2, 6>(a, B, ygroups, vec, stream);
        case 8: return launch_gemm<2, 8>(a, B, ygroups, vec, stream);
        case 9: return launch_gemm<2, 9>(a, B, ygroups, vec, stream);
        default: return IRM_EINVAL;
    }
}
