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
#include <stdlib.h>

struct GemmArgs {
    const float* Wp; long w_bs;   // packed weights, per-batch stride (0 = shared)
    const float* X;  long x_bs;   // [B][K][N]
    float* Y;        long y_bs;   // [B][M][N]
    const float* R;  long r_bs;   // residual [B][M][N] or null
    const float* rscale;          // optional [M]: y = acc + res * rscale[co] (MaIR skip_scale)
    const float* bias;            // [M] or null
    const float* stats;           // [B][2][N] mean, rstd (LN prologue) or null
    const float* lnw;             // [K]
    const float* lnb;             // [K] (WithBias only)
    int M, K, N;
    int mtiles, ksteps;           // ceil(M/16), 4*ceil(K/16)
    int ln_mode, act;
    float* stats_out;             // optional [B][2][N]: LayerNorm statistics of Y over its M channels
    float eps;                    // (single-pass launches only: every output channel lives in one workgroup)
    int dbg;                      // timing experiments only (IRM_GEMM_DBG): 1 = no DMA, 2 = no stores
};


// LayerNorm statistics of the finished output tile straight from the accumulators: lane (r, g)
// holds pixels pix..pix+3 of channels 16 c + r, so a channel reduction is CT in-lane adds plus
// a 16-lane butterfly.  Two passes (mean, then centred squares) in registers.
template <int PT, int CT>
__device__ __forceinline__ void irm_stats_from_acc(const f32x4 (&acc)[PT][CT], int mt0, int M, int N, int r,
                                                   const int (&pixs)[PT], float* st, float eps) {
    float sum[PT][4], sq[PT][4];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sum[p][e] = 0.f; sq[p][e] = 0.f; }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const bool ok = (mt0 + c) * 16 + r < M;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[p][e] += ok ? acc[p][c][e] : 0.f;
    }
    const float inv = 1.0f / (float)M;
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum[p][e] += __shfl_xor(sum[p][e], o);
            sum[p][e] *= inv;                                   // mean
        }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const bool ok = (mt0 + c) * 16 + r < M;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = acc[p][c][e] - sum[p][e];
                sq[p][e] += ok ? d * d : 0.f;
            }
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sq[p][e] += __shfl_xor(sq[p][e], o);
            sq[p][e] = 1.0f / sqrtf(sq[p][e] * inv + eps);      // rstd
        }
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (pixs[p] + e < N) {
                    st[pixs[p] + e] = sum[p][e];
                    st[N + pixs[p] + e] = sq[p][e];
                }
        }
    }
}

template <int PT, int CT, bool VEC>
__global__ __launch_bounds__(256) void gemm_pw_kernel(GemmArgs a) {
    IRM_KERNEL_ENTRY();
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
        int pixs[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) pixs[p] = n0 + wave * 16 * PT + p * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int co = (mt0 + c) * 16 + (lane & 15);
            const bool row_ok = c < nct && co < a.M;
            const float bv = (a.bias && row_ok) ? a.bias[co] : 0.0f;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                float4 v = make_float4(acc[p][c][0] + bv, acc[p][c][1] + bv, acc[p][c][2] + bv, acc[p][c][3] + bv);
                if (a.act != IRM_ACT_NONE) {
                    v.x = irm_act(v.x, a.act); v.y = irm_act(v.y, a.act);
                    v.z = irm_act(v.z, a.act); v.w = irm_act(v.w, a.act);
                }
                if (R && row_ok && pixs[p] < a.N) {
                    const float4 q = irm_ld4<VEC>(R + (long)co * a.N, pixs[p], a.N);
                    const float sc = a.rscale ? a.rscale[co] : 1.0f;
                    v.x = fmaf(q.x, sc, v.x); v.y = fmaf(q.y, sc, v.y); v.z = fmaf(q.z, sc, v.z); v.w = fmaf(q.w, sc, v.w);
                }
                acc[p][c] = (f32x4){v.x, v.y, v.z, v.w};
            }
        }
        if (a.stats_out)
            irm_stats_from_acc<PT, CT>(acc, mt0, a.M, a.N, lane & 15, pixs, a.stats_out + (long)b * 2 * a.N, a.eps);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int co = (mt0 + c) * 16 + (lane & 15);
            if (c < nct && co < a.M) {
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    if (pixs[p] < a.N)
                        irm_st4<VEC>(Y + (long)co * a.N, pixs[p], a.N,
                                     make_float4(acc[p][c][0], acc[p][c][1], acc[p][c][2], acc[p][c][3]));
            }
        }
    }
}

__device__ __attribute__((noinline)) float irm_act_slow(float v, int act) { return irm_act(v, act); }

// Masked-off lanes of the epilogue stores write here instead of being skipped, so that every wave
// issues exactly PT*CT store instructions per pass and the counted vmcnt waits stay exact.
__device__ float4 irm_dump[256 * 64];

template <int N>
__device__ __forceinline__ void irm_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// F16: the GEMM proper runs on the fp16 matrix cores as an fp32 emulation - both operands are split into
// fp16 hi + lo parts (x = hi + lo exactly up to 2^-22 |x|; the weights are split on the host), three
// 16x16x16 MFMAs (lo*hi, hi*lo, hi*hi) accumulate in fp32: 2^-21 relative per product instead of 2^-24, at
// 24 instead of 128 matrix cycles per tile and 16 channels, and off the fp32 datapath the VALU needs.
typedef _Float16 irm_h4 __attribute__((ext_vector_type(4)));

template <int PT, int CT, int NS, int LN, bool RES, bool F16 = false>
__global__ __launch_bounds__(256, (PT == 2 && NS == 3) ? 3 : 2) void gemm_ring_kernel(GemmArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int BN = 64 * PT, BK = 16;            // pixels per workgroup, channels per stage
    constexpr int RPU = 256 / BN;                   // X rows per 1 KiB DMA instruction (PT 2: 2, PT 4: 1)
    constexpr int XS = BK * BN;                 // floats of X per stage
    constexpr int STG = XS + CT * 256;          // floats per stage
    constexpr int WL = (CT + 3) / 4;            // weight DMA instructions per wave per stage
    constexpr int LPS = PT + WL;                // DMA instructions per wave per stage
    constexpr int NST = PT * CT;                // store instructions per wave per pass (always issued)
    static_assert((NS - 2) * LPS + NST <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: address math on the SALU
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const float* X = a.X + (long)b * a.x_bs;
    const float* Wp = a.Wp + (long)b * a.w_bs;
    float* Y = a.Y + (long)b * a.y_bs;
    const float* R = (RES && a.R) ? a.R + (long)b * a.r_bs : nullptr;
    const int KP = a.ksteps * 4;
    float* lnp = smem + NS * STG;               // [2][KP]: LN weight, LN bias (zero padded)

    float rs[PT], nmr[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) { rs[p] = 1.f; nmr[p] = 0.f; }
    if (LN != IRM_LN_NONE) {
        const float* st = a.stats + (long)b * 2 * a.N;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int pix = min(n0 + wave * 16 * PT + p * 16 + r, a.N - 1);
            const float m = st[pix], q = st[a.N + pix];
            rs[p] = q;
            nmr[p] = -m * q;
        }
        for (int i = tid; i < KP; i += 256) {
            lnp[i] = i < a.K ? a.lnw[i] : 0.0f;
            lnp[KP + i] = (LN == IRM_LN_WITHBIAS && i < a.K) ? a.lnb[i] : 0.0f;
        }
        __syncthreads();                         // before any DMA is in flight
    }

    const int S = a.ksteps / 4;
    const int nchunks = (a.mtiles + CT - 1) / CT;
    const int my_chunks = (nchunks - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int TOT = my_chunks * S;

    const int xrow = wave * PT * RPU + (lane * 4) / BN;        // + j * RPU
    // ODD rows of a stage are stored rotated by 16 pixels (the per-lane DMA source is free): the operand reads below
    // (ds_read_b32, lanes (r, g): row 4 kk + g, pixel r) then put rows g and g + 1 of a 32-lane half on different banks -
    // unrotated, rows are BN floats = a multiple of 32 banks apart and every read is a 2-way conflict
    // (tools/probes/lds_conflict.hip patterns 8 / 9: 4.0 -> 2.06 LDS cycles per instruction).  Row parity: PT 2 = two
    // rows per DMA instruction, lanes 32-63 the odd one; PT 4 = one row per instruction, odd j.
    const int pc = (lane * 4) % BN;
    const int xcol = min(n0 + ((pc - (PT == 2 ? 16 * ((lane >> 5) & 1) : 0)) & (BN - 1)), a.N - 4);
    const int xcol_odd = PT == 2 ? xcol : min(n0 + ((pc - 16) & (BN - 1)), a.N - 4);
    // per-lane source of row xrow; stages and the PT rows of a wave are uniform offsets from it (the VALU
    // shares its datapath with the f32 MFMA: two adds per DMA instead of a clamp and a 64-bit multiply-add)
    const float* xlane = X + (long)xrow * a.N + xcol;
    const float* xlane_odd = X + (long)xrow * a.N + xcol_odd;
    const int rdsh = 16 * (g & 1);                             // the same rotation on the read side
    const bool ragged = (a.K % BK) != 0;                       // only then can a row index pass K - 1

    auto issue = [&](int it) {
        if (IRM_DBG(a.dbg, 1)) return;
        const int ci = it / S, s = it - ci * S;
        const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * CT;
        float* xb = smem + (it % NS) * STG;
        const bool tail = ragged && s == S - 1;
        const float* sbase = xlane + (long)s * BK * a.N;
        const float* sbase_odd = xlane_odd + (long)s * BK * a.N;
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const bool oddj = PT == 4 && (j & 1);
            const float* src = (oddj ? sbase_odd : sbase) + (long)(j * RPU) * a.N;
            if (tail) src = X + (long)min(s * BK + xrow + j * RPU, a.K - 1) * a.N + (oddj ? xcol_odd : xcol);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xb + (wave * PT + j) * 256),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int ct = min(wave + 4 * i, CT - 1);
            const int mt = min(mt0 + ct, a.mtiles - 1);
            const float* src = Wp + ((long)mt * a.ksteps + s * 4) * 64 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xb + XS + ct * 256), 16, 0, 0);
        }
    };

    f32x4 acc[PT][CT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pixs[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) pixs[p] = n0 + wave * 16 * PT + p * 16 + g * 4;
    // residual / bias of the running pass: loaded (clamped addresses, back to back) during the
    // pass's first stage and first used in its epilogue, so their latency hides under the MFMAs
    float4 rv[RES ? PT : 1][RES ? CT : 1];
    float bvs[CT];

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < TOT) issue(j);

    int s = 0, ci = 0, since_epi = 1 << 20;
    for (int it = 0; it < TOT; ++it) {
        // stage `it` has landed once at most `rem` younger stages are still in flight
        // vmcnt retires in issue order and counts stores too: the NST stores of the last epilogue are
        // younger than every DMA issued before it, so for the next NS-1 stages they are added to the
        // number of operations allowed to stay in flight (otherwise each pass boundary would drain
        // two prefetched stages and wait for write acknowledgements)
        const int rem = min(NS - 2, TOT - 1 - it);
        const bool st = since_epi <= NS - 1 && S >= NS - 1;
        if (S < NS - 1 && since_epi <= NS - 1) irm_wait_vmcnt<0>();
        else if (rem >= 2 && NS >= 4) { if (st) irm_wait_vmcnt<(NS >= 4 ? 2 : 0) * LPS + NST>(); else irm_wait_vmcnt<(NS >= 4 ? 2 : 0) * LPS>(); }
        else if (rem == 1 && NS >= 3) { if (st) irm_wait_vmcnt<LPS + NST>(); else irm_wait_vmcnt<LPS>(); }
        else { if (st) irm_wait_vmcnt<NST>(); else irm_wait_vmcnt<0>(); }
        ++since_epi;
        asm volatile("s_barrier" ::: "memory");
        // operands of the first k-step before anything else: their LDS round trip runs under the address
        // math of the DMA issue instead of in front of the first MFMA
        const float* xb = smem + (it % NS) * STG;
        const float* wb = xb + XS;
        float x0[PT], b0[CT];
#pragma unroll
        for (int p = 0; p < PT; ++p) x0[p] = xb[g * BN + ((wave * 16 * PT + p * 16 + rdsh) & (BN - 1)) + r];
#pragma unroll
        for (int c = 0; c < CT; ++c) b0[c] = F16 ? 0.0f : wb[c * 4 * 64 + lane];
        if (it + NS - 1 < TOT) issue(it + NS - 1);
        if (s == 0) {
            const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * CT;
            if (RES && R) {
#pragma unroll
                for (int c = 0; c < (RES ? CT : 0); ++c) {
                    const long row = (long)min((mt0 + c) * 16 + r, a.M - 1) * a.N;
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        rv[p][c] = *reinterpret_cast<const float4*>(R + row + min(pixs[p], a.N - 4));
                }
            }
            if (a.bias) {
#pragma unroll
                for (int c = 0; c < CT; ++c) bvs[c] = a.bias[min((mt0 + c) * 16 + r, a.M - 1)];
            }
        }

        if constexpr (F16) {
            // stage layout of the weights: per tile 64 lanes x (4 hi halves | 4 lo halves); lane (g, m) holds
            // W[m][16 s + 4 j + g], j = 0..3 - the same channel order in which lane (g, i) reads x below
            irm_h4 ah[PT], al[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                float xv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float x = kk == 0 ? x0[p] : xb[(kk * 4 + g) * BN + ((wave * 16 * PT + p * 16 + rdsh) & (BN - 1)) + r];
                    if (LN == IRM_LN_WITHBIAS)
                        xv[kk] = fmaf(fmaf(x, rs[p], nmr[p]), lnp[s * BK + kk * 4 + g], lnp[KP + s * BK + kk * 4 + g]);
                    else if (LN == IRM_LN_BIASFREE) xv[kk] = x * rs[p] * lnp[s * BK + kk * 4 + g];
                    else xv[kk] = irm_sat_h(x * 0.0625f);            // not normalised: 2^-4 keeps |x| up to 1e6 inside fp16
                }
                irm_split4(xv, ah[p], al[p]);
            }
            irm_h4 bh[CT], bl[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                bh[c] = *reinterpret_cast<const irm_h4*>(wb + c * 256 + lane * 2);
                bl[c] = *reinterpret_cast<const irm_h4*>(wb + c * 256 + 128 + lane * 2);
            }
            // three sweeps over the PT x CT independent accumulators (no MFMA waits on the one before it)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[p], bh[c], acc[p][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[p], bl[c], acc[p][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[p], bh[c], acc[p][c], 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float af[PT], bf[CT];
            float wk = 1.f, bk = 0.f;
            if (LN != IRM_LN_NONE) {
                wk = lnp[s * BK + kk * 4 + g];
                if (LN == IRM_LN_WITHBIAS) bk = lnp[KP + s * BK + kk * 4 + g];
            }
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const float x = kk == 0 ? x0[p] : xb[(kk * 4 + g) * BN + ((wave * 16 * PT + p * 16 + rdsh) & (BN - 1)) + r];
                if (LN == IRM_LN_WITHBIAS) af[p] = fmaf(fmaf(x, rs[p], nmr[p]), wk, bk);
                else if (LN == IRM_LN_BIASFREE) af[p] = x * rs[p] * wk;
                else af[p] = x;
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) bf[c] = kk == 0 ? b0[c] : wb[(c * 4 + kk) * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][c] = irm_mfma16(af[p], bf[c], acc[p][c]);
        }

        }
        if (++s == S) {
            // pass finished: (+ bias, activation, + residual) and store tiles mt0 .. mt0 + CT - 1
            const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * CT;
            if constexpr (F16 && LN == IRM_LN_NONE) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[p][c] *= 16.0f;      // undo the operand scale
            }
            if (a.bias) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[p][c] += bvs[c];
            }
            if (a.act != IRM_ACT_NONE) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[p][c][e] = irm_act_slow(acc[p][c][e], a.act);
            }
            if (RES && R) {
#pragma unroll
                for (int c = 0; c < (RES ? CT : 0); ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        const float sc = a.rscale ? a.rscale[min((mt0 + c) * 16 + r, a.M - 1)] : 1.0f;
                        acc[p][c][0] = fmaf(rv[p][c].x, sc, acc[p][c][0]); acc[p][c][1] = fmaf(rv[p][c].y, sc, acc[p][c][1]);
                        acc[p][c][2] = fmaf(rv[p][c].z, sc, acc[p][c][2]); acc[p][c][3] = fmaf(rv[p][c].w, sc, acc[p][c][3]);
                    }
            }
            if (a.stats_out) irm_stats_from_acc<PT, CT>(acc, mt0, a.M, a.N, r, pixs, a.stats_out + (long)b * 2 * a.N, a.eps);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = (mt0 + c) * 16 + r;
                const bool row_ok = mt0 + c < a.mtiles && co < a.M;
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    // masked lanes write a dump slot so that exactly NST store instructions are issued
                    float4* dst = (row_ok && pixs[p] < a.N && !IRM_DBG(a.dbg, 2))
                        ? reinterpret_cast<float4*>(Y + (long)co * a.N + pixs[p])
                        : irm_dump + ((blockIdx.x & 255) * 64 + lane);
                    *dst = make_float4(acc[p][c][0], acc[p][c][1], acc[p][c][2], acc[p][c][3]);
                    acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            s = 0;
            ++ci;
            since_epi = 1;
        }
    }
}

template <int PT, int CT, int NS, int LN, bool RES, bool F16>
static int launch_ring_ns(const GemmArgs& a, int B, int ygroups, hipStream_t stream) {
    constexpr int BN = 64 * PT;
    const size_t lds = ((size_t)NS * (16 * BN + CT * 256) + 2 * (size_t)a.ksteps * 4) * sizeof(float);
    IRM_ALLOW_BIG_LDS((&gemm_ring_kernel<PT, CT, NS, LN, RES, F16>));
    dim3 grid((a.N + BN - 1) / BN, ygroups, B);
    hipLaunchKernelGGL((gemm_ring_kernel<PT, CT, NS, LN, RES, F16>), grid, dim3(256), lds, stream, a);
    return irm_launch_status();
}

template <int PT, int CT, int LN, bool RES, bool F16 = false>
static int launch_ring(const GemmArgs& a, int B, int ygroups, hipStream_t stream) {
    // Residual GEMMs with up to 6 output tiles per pass and K <= 512 (attention apply, project_out at C <= 192): a 3-deep
    // ring of 14 KiB stages and <= 168 registers let THREE workgroups share a CU (launch bounds of the kernel), so one
    // workgroup's store epilogue and the next one's ring prologue overlap a third workgroup's steady state: M = K = 96 at
    // 6 x 512^2 409 -> 381 us, M = K = 192 at 6 x 128^2 65 -> 57 us, M 192 K 510 129 -> 125 us (round 3,
    // tools/bench_apply.py; a 5-deep ring at two workgroups had bought nothing in round 2).  Long reductions (K 1021:
    // 136 -> 144 us) want the deeper ring; 8 / 9 tiles per pass need more registers: both keep two workgroups.
#ifndef IRM_NO_RING3                              // (variant builds for A/B: tools/build_variant.sh NAME -DIRM_NO_RING3 gemm_pw.hip)
    if constexpr (PT == 2 && RES && CT <= 6) {
        if (a.K <= 512) return launch_ring_ns<PT, CT, 3, LN, RES, F16>(a, B, ygroups, stream);
    }
#endif
    return launch_ring_ns<PT, CT, PT == 2 ? 4 : 3, LN, RES, F16>(a, B, ygroups, stream);
}

// instantiated combinations: 64-pixel waves (PT 4) only without a residual (the prefetched residual
// doubles the accumulator-sized register block); LayerNorm prologue only without a residual (the
// path never combines them)
template <int CT>
static int launch_ring_any(const GemmArgs& a, int B, int ygroups, int pt, hipStream_t stream) {
    if (a.R) {
        if (a.ln_mode != IRM_LN_NONE) return IRM_EINVAL;
        return launch_ring<2, CT, IRM_LN_NONE, true>(a, B, ygroups, stream);
    }
    if (pt == 4) {
        if (a.ln_mode == IRM_LN_WITHBIAS) return launch_ring<4, CT, IRM_LN_WITHBIAS, false>(a, B, ygroups, stream);
        if (a.ln_mode == IRM_LN_BIASFREE) return launch_ring<4, CT, IRM_LN_BIASFREE, false>(a, B, ygroups, stream);
        return launch_ring<4, CT, IRM_LN_NONE, false>(a, B, ygroups, stream);
    }
    if (a.ln_mode == IRM_LN_WITHBIAS) return launch_ring<2, CT, IRM_LN_WITHBIAS, false>(a, B, ygroups, stream);
    if (a.ln_mode == IRM_LN_BIASFREE) return launch_ring<2, CT, IRM_LN_BIASFREE, false>(a, B, ygroups, stream);
    return launch_ring<2, CT, IRM_LN_NONE, false>(a, B, ygroups, stream);
}

// split-fp16 variant: 64-pixel waves, no residual
template <int CT>
static int launch_ring_split(const GemmArgs& a, int B, int ygroups, hipStream_t stream) {
    if (a.R) {
        if (a.ln_mode != IRM_LN_NONE) return IRM_EINVAL;
        return launch_ring<2, CT, IRM_LN_NONE, true, true>(a, B, ygroups, stream);
    }
    static const int pt = irm_probe_int("IRM_GEMM_SPLIT_PT", 4);
    if (pt == 2) {
        if (a.ln_mode == IRM_LN_WITHBIAS) return launch_ring<2, CT, IRM_LN_WITHBIAS, false, true>(a, B, ygroups, stream);
        if (a.ln_mode == IRM_LN_BIASFREE) return launch_ring<2, CT, IRM_LN_BIASFREE, false, true>(a, B, ygroups, stream);
        return launch_ring<2, CT, IRM_LN_NONE, false, true>(a, B, ygroups, stream);
    }
    if (a.ln_mode == IRM_LN_WITHBIAS) return launch_ring<4, CT, IRM_LN_WITHBIAS, false, true>(a, B, ygroups, stream);
    if (a.ln_mode == IRM_LN_BIASFREE) return launch_ring<4, CT, IRM_LN_BIASFREE, false, true>(a, B, ygroups, stream);
    return launch_ring<4, CT, IRM_LN_NONE, false, true>(a, B, ygroups, stream);
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
static bool irm_force_generic() {
    static const bool v = irm_probe_set("IRM_GEMM_GENERIC");   // A/B switch for benchmarking only
    return v;
}
// gemm_xres.hip: input-resident variant of the emulated GEMM for K <= 96
int irm_gemm_xres_dispatch(const float* wp, const float* x, long x_bs, float* y, long y_bs, const float* bias,
                           const float* stats, const float* lnw, const float* lnb, int ln_mode, int act, int B, int M,
                           int K, int N, hipStream_t stream);

static int gemm_entry(const float* wp, long w_bs, const float* x, long x_bs, float* y, long y_bs,
                      const float* res, long r_bs, const float* bias, const float* stats,
                      const float* lnw, const float* lnb, int ln_mode, int act, int B, int M, int K,
                      int N, int ct, int ygroups, float* stats_out, float eps, const float* res_scale,
                      bool split, hipStream_t stream) {
    if (!wp || !x || !y || B <= 0 || M <= 0 || K <= 0 || N <= 0) return IRM_EINVAL;
    if (ln_mode != IRM_LN_NONE && (!stats || !lnw || (ln_mode == IRM_LN_WITHBIAS && !lnb))) return IRM_EINVAL;
    if (ln_mode < 0 || ln_mode > 2 || act < 0 || act > 3) return IRM_EINVAL;
    if ((w_bs & 3) || !irm_aligned16(wp)) return IRM_EINVAL;
    // 16-byte fast path needs every row of every operand 16-byte aligned
    const bool vec = !(N & 3) && !(x_bs & 3) && !(y_bs & 3) && !(r_bs & 3) && irm_aligned16(x) &&
                     irm_aligned16(y) && irm_aligned16(res) && irm_aligned16(stats);
    GemmArgs a;
    a.Wp = wp; a.w_bs = w_bs; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.R = res; a.r_bs = r_bs;
    a.bias = bias; a.stats = stats; a.lnw = lnw; a.lnb = lnb; a.rscale = res ? res_scale : nullptr;
    a.M = M; a.K = K; a.N = N; a.mtiles = (M + 15) / 16; a.ksteps = 4 * ((K + 15) / 16);
    a.ln_mode = ln_mode; a.act = act; a.stats_out = stats_out; a.eps = eps;
    a.dbg = irm_probe_int("IRM_GEMM_DBG", 0);
    // fused output statistics need every output channel in one workgroup pass
    if (stats_out && (ct <= 0 || a.mtiles > ct)) return IRM_EINVAL;
    const int nchunks = (a.mtiles + (ct > 0 ? ct : 1) - 1) / (ct > 0 ? ct : 1);
    if (ygroups <= 0) ygroups = 1;
    if (ygroups > nchunks) ygroups = nchunks;
    if (B > 65535 || ygroups > 65535) return IRM_EINVAL;
    if (split) {
        // weights packed by the caller as fp16 hi/lo pairs: only the ring kernel understands them
        if (!vec || N < 4 || (res && ln_mode != IRM_LN_NONE)) return IRM_EINVAL;
        static const bool no_xres = irm_probe_set("IRM_GEMM_NO_XRES");
        if (K <= 192 && ln_mode != IRM_LN_NONE && !stats_out && !res && !w_bs && !no_xres && B <= 65535 && (long)(N + 127) / 128 <= 2147483647L) {
            const int rc = irm_gemm_xres_dispatch(wp, x, x_bs, y, y_bs, bias, stats, lnw, lnb, ln_mode, act, B, M, K, N, stream);
            if (rc != IRM_EINVAL) return rc;
        }
        switch (ct) {
            case 3: return launch_ring_split<3>(a, B, ygroups, stream);
            case 4: return launch_ring_split<4>(a, B, ygroups, stream);
            case 6: return launch_ring_split<6>(a, B, ygroups, stream);
            case 8: return launch_ring_split<8>(a, B, ygroups, stream);
            case 9: return launch_ring_split<9>(a, B, ygroups, stream);
            default: return IRM_EINVAL;
        }
    }
    if (vec && N >= 4 && !irm_force_generic() && !(res && ln_mode != IRM_LN_NONE)) {
        // 64 pixels per wave (PT 4) doubles the MFMAs per barrier; it is used when the accumulators
        // (+ the prefetched residual) still fit 2 waves per SIMD and the grid stays large
        int pt = (!res && (long)B * ((N + 255) / 256) * ygroups >= 512) ? 4 : 2;
        pt = irm_probe_int("IRM_GEMM_PT", pt);
        switch (ct) {
            case 3: return launch_ring_any<3>(a, B, ygroups, pt, stream);
            case 4: return launch_ring_any<4>(a, B, ygroups, pt, stream);
            case 6: return launch_ring_any<6>(a, B, ygroups, pt, stream);
            case 8: return launch_ring_any<8>(a, B, ygroups, pt, stream);
            case 9: return launch_ring_any<9>(a, B, ygroups, pt, stream);
            default: return IRM_EINVAL;
        }
    }
    switch (ct) {
        case 3: return launch_gemm<2, 3>(a, B, ygroups, vec, stream);
        case 4: return launch_gemm<2, 4>(a, B, ygroups, vec, stream);
        case 6: return launch_gemm<2, 6>(a, B, ygroups, vec, stream);
        case 8: return launch_gemm<2, 8>(a, B, ygroups, vec, stream);
        case 9: return launch_gemm<2, 9>(a, B, ygroups, vec, stream);
        default: return IRM_EINVAL;
    }
}

extern "C" int irm_gemm1x1_f32(const float* wp, long w_bs, const float* x, long x_bs, float* y, long y_bs,
                               const float* res, long r_bs, const float* bias, const float* stats,
                               const float* lnw, const float* lnb, int ln_mode, int act, int B, int M, int K,
                               int N, int ct, int ygroups, float* stats_out, float eps, const float* res_scale,
                               hipStream_t stream) {
    return gemm_entry(wp, w_bs, x, x_bs, y, y_bs, res, r_bs, bias, stats, lnw, lnb, ln_mode, act, B, M, K, N, ct,
                      ygroups, stats_out, eps, res_scale, false, stream);
}

extern "C" int irm_gemm1x1_f16x3_f32(const float* wp_split, long w_bs, const float* x, long x_bs, float* y, long y_bs,
                                     const float* res, long r_bs, const float* bias, const float* stats,
                                     const float* lnw, const float* lnb, int ln_mode, int act, int B, int M, int K,
                                     int N, int ct, int ygroups, float* stats_out, float eps,
                                     const float* res_scale, hipStream_t stream) {
    return gemm_entry(wp_split, w_bs, x, x_bs, y, y_bs, res, r_bs, bias, stats, lnw, lnb, ln_mode, act, B, M, K, N, ct,
                      ygroups, stats_out, eps, res_scale, true, stream);
}
