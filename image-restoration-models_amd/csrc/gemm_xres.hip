// LayerNorm + 1x1 conv for K <= 192 input channels as an fp32 emulation on the fp16 matrix cores, with the
// INPUT tile resident: the workgroup's 256 (K <= 96) or 128 (K <= 192) pixels x K channels are read once,
// normalised, split into fp16 hi + lo (x = hi + lo up to 2^-22 |x|) and parked in LDS in MFMA A-operand order
// (96 KiB); after that every output-channel pass is pure matrix work - the split weights (packed by the host,
// L2 resident) stream through a small LDS-DMA ring, three v_mfma_f32_16x16x32_f16 (lo*hi, hi*lo, hi*hi) per
// tile and 32 input channels accumulate in fp32.  The streaming kernel (gemm_pw.hip, F16) re-reads and
// re-splits the input for every pass; here the passes cost no VALU and no input traffic.
//   replaces: LayerNorm (restormer.py:25-70) + Attention.qkv / FeedForward.project_in (restormer.py:82,105)
#include "irm_common.h"
#include <stdlib.h>

typedef _Float16 xr_h4 __attribute__((ext_vector_type(4)));

struct XresArgs {
    const float* Wp;              // pack_gemm_weight_split: [mtile][stage][hi 64x4 | lo 64x4] halves
    const float* X; long x_bs;    // [B][K][N]
    float* Y; long y_bs;          // [B][M][N]
    const float* bias;            // [M] or null
    const float* stats;           // [B][2][N] mean, rstd or null (ln_mode 0)
    const float* lnw; const float* lnb;
    int M, K, N, mtiles, stages;  // stages = ceil(K/16)
    int ln_mode, act;
    int dbg;                      // IRM_XRES_DBG (timing only): 1 = no input conversion, 2 = no passes, 4 = no stores
};

template <int N>
__device__ __forceinline__ void xr_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __attribute__((noinline)) float xr_act(float v, int act) { return irm_act(v, act); }

typedef _Float16 xr_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ xr_h8 xr_cat(xr_h4 a, xr_h4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);      // register-tuple concatenation, no moves
}

// KT = 16-channel stages of the input (even: two of them feed one 16x16x32 MFMA), CT = output tiles per pass,
// NS = ring depth of the weight stages (one ring stage = the weights of CT tiles for 32 input channels).
// One workgroup per CU: 16 / WP compute waves (256 pixels, WP pixel tiles each) + 1 loader wave.  The loader alone
// issues the weight DMAs and waits for them: vmcnt is per wave and retires in order, stores included, so a wave
// that both stores results and waits for operands sits out the write acknowledgements at every pass boundary;
// split this way the compute waves never wait on memory, the stage barrier is the only hand-over.
template <int KT, int CT, int NS, int WP, int NPT>
__global__ __launch_bounds__((NPT / WP + 1) * 64, 1) void gemm_xres_kernel(XresArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int NW = NPT / WP;                   // compute waves, WP pixel tiles each
    constexpr int BN = NPT * 16;                   // pixels per workgroup (256 for K <= 96, 128 / 64 for K <= 192 / 384:
                                                   // the resident input is 96 KiB in every case)
    constexpr int NPAR = NW * 64 / BN;             // threads per pixel in the conversion
    constexpr int NQ = 4 * KT / NPAR;              // channel-quad groups per thread
    static_assert(NW * 64 % BN == 0 && (4 * KT) % NPAR == 0, "conversion mapping");
    constexpr int AH = KT * NPT * 128;             // floats of the hi (or lo) half of the resident input
    constexpr int WST = CT * 512;                  // floats per weight stage (CT tiles x 2 KiB)
    constexpr int KP = KT / 2;                     // stage pairs
    static_assert(KT % 2 == 0, "two 16-channel stages per MFMA");
    static_assert((NS - 2) * CT * 2 <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ah = smem;                              // [stage pair][ptile][hi even, hi odd, lo even, lo odd][lane][4 halves]
    float* wring = smem + 2 * AH;
    float* lnp = wring + NS * WST;                 // [2][16 KT] LayerNorm weight, bias (zero padded)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const float* X = a.X + (long)b * a.x_bs;
    float* Y = a.Y + (long)b * a.y_bs;

    const int nchunks = (a.mtiles + CT - 1) / CT;
    const int TOT = IRM_DBG(a.dbg, 2) ? 0 : nchunks * KP;

    if (a.ln_mode != IRM_LN_NONE) {
        for (int k = tid; k < 16 * KT; k += (NW + 1) * 64) {
            lnp[k] = k < a.K ? a.lnw[k] : 0.0f;
            lnp[16 * KT + k] = (a.ln_mode == IRM_LN_WITHBIAS && k < a.K) ? a.lnb[k] : 0.0f;
        }
    }
    __syncthreads();
    if (wave == NW) {
        // ------------------------------------------------------------------ loader wave
        auto issue = [&](int it) {
            const int ci = it / KP, t = it - ci * KP;
            float* dst = wring + (it % NS) * WST;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int mt = min(ci * CT + ct, a.mtiles - 1);
                // LDS image of a tile: [hi stage 2t | hi stage 2t+1 | lo 2t | lo 2t+1] (512 B each), so that the
                // two halves of an 8-half operand sit one ds_read2st64_b64 apart and land in adjacent registers
                const float* src = a.Wp + ((long)mt * a.stages + 2 * t + (lane >> 5)) * 256 + (lane & 31) * 4;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + h * 128),
                                                     (__attribute__((address_space(3))) void*)(dst + ct * 512 + h * 256),
                                                     16, 0, 0);
            }
        };
#pragma unroll
        for (int j = 0; j < NS - 1; ++j)
            if (j < TOT) issue(j);
        __syncthreads();                           // pairs with the barrier after the input conversion
        for (int it = 0; it < TOT; ++it) {
            if (TOT - 1 - it >= NS - 2 && NS >= 3) xr_wait_vmcnt<(NS - 2) * CT * 2>();
            else xr_wait_vmcnt<0>();
            asm volatile("s_barrier" ::: "memory");   // stage it is in LDS; everybody is done with stage it-1
            if (it + NS - 1 < TOT) issue(it + NS - 1);
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    // resident input: thread = (pixel, half of the channel-quad groups); channel of k-slot (g, j) of stage s
    // is 16 s + 4 j + g (the order the weights are packed in)
    if (!IRM_DBG(a.dbg, 1)) {
        const int px = tid % BN, par = tid / BN;
        const int n = min(n0 + px, a.N - 1);
        float mean = 0.f, rstd = 1.f;
        if (a.ln_mode != IRM_LN_NONE) {
            const float* st = a.stats + (long)b * 2 * a.N;
            mean = st[n]; rstd = st[a.N + n];
        }
        const int pt = px >> 4, i = px & 15;
        // all 8 KT loads of this thread in flight before the first conversion (one memory round trip)
        float xv[NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kg = q * NPAR + par, s = kg >> 2, gg = kg & 3;   // (stage, g) pair: kg = 4 s + g
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = min(16 * s + 4 * j + gg, a.K - 1);
                xv[q][j] = X[(long)k * a.N + n];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kg = q * NPAR + par, s = kg >> 2, gg = kg & 3;
            xr_h4 h, l;
            float xn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * s + 4 * j + gg;
                float x = xv[q][j];
                if (a.ln_mode == IRM_LN_WITHBIAS) x = fmaf((x - mean) * rstd, lnp[k], lnp[16 * KT + k]);
                else if (a.ln_mode == IRM_LN_BIASFREE) x = x * rstd * lnp[k];
                if (k >= a.K) x = 0.0f;
                xn[j] = x;
            }
            irm_split4(xn, h, l);
            // image of a (stage pair, pixel tile): [hi even stage | hi odd stage | lo even | lo odd], 512 B each
            const int off = (((s >> 1) * NPT + pt) * 4 + (s & 1)) * 128 + (gg * 16 + i) * 2;   // floats
            *reinterpret_cast<xr_h4*>(ah + off) = h;
            *reinterpret_cast<xr_h4*>(ah + off + 256) = l;
        }
    }
    __syncthreads();                               // the resident tile is complete and visible

    f32x4 acc[WP][CT];
#pragma unroll
    for (int p = 0; p < WP; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pixs[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) pixs[p] = n0 + (wave * WP + p) * 16 + g * 4;

    int t = 0, ci = 0;
    for (int it = 0; it < TOT; ++it) {
        asm volatile("s_barrier" ::: "memory");
        const float* wb = wring + (it % NS) * WST;
        xr_h8 xh[WP], xl[WP];
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const float* xa = ah + ((t * NPT + wave * WP + p) * 4) * 128 + lane * 2;
            xh[p] = xr_cat(*reinterpret_cast<const xr_h4*>(xa), *reinterpret_cast<const xr_h4*>(xa + 128));
            xl[p] = xr_cat(*reinterpret_cast<const xr_h4*>(xa + 256), *reinterpret_cast<const xr_h4*>(xa + 384));
        }
        // groups of CG tiles: their weights in registers, then the three partial products as three sweeps over the
        // 2 x CG independent accumulators (small terms first)
        constexpr int CG = WP >= 4 ? (CT % 3 == 0 ? 3 : 2) : CT;   // WP 2: all weight operands of the stage up front (one LDS round trip)
#pragma unroll
        for (int c0 = 0; c0 < CT; c0 += CG) {
            xr_h8 bh[CG], bl[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float* w = wb + (c0 + c) * 512 + lane * 2;
                bh[c] = xr_cat(*reinterpret_cast<const xr_h4*>(w), *reinterpret_cast<const xr_h4*>(w + 128));
                bl[c] = xr_cat(*reinterpret_cast<const xr_h4*>(w + 256), *reinterpret_cast<const xr_h4*>(w + 384));
            }
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int p = 0; p < WP; ++p)
                    acc[p][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[p], bh[c], acc[p][c0 + c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int p = 0; p < WP; ++p)
                    acc[p][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[p], bl[c], acc[p][c0 + c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int p = 0; p < WP; ++p)
                    acc[p][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[p], bh[c], acc[p][c0 + c], 0, 0, 0);
        }

        if (++t == KP) {
            const int mt0 = ci * CT;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = (mt0 + c) * 16 + r;
                const bool row_ok = mt0 + c < a.mtiles && co < a.M;
                const float bv = (a.bias && row_ok) ? a.bias[co] : 0.0f;
#pragma unroll
                for (int p = 0; p < WP; ++p) {
                    float4 v = make_float4(acc[p][c][0] + bv, acc[p][c][1] + bv, acc[p][c][2] + bv, acc[p][c][3] + bv);
                    if (a.act != IRM_ACT_NONE) {
                        v.x = xr_act(v.x, a.act); v.y = xr_act(v.y, a.act); v.z = xr_act(v.z, a.act); v.w = xr_act(v.w, a.act);
                    }
                    if (row_ok && pixs[p] < a.N && !IRM_DBG(a.dbg, 4)) *reinterpret_cast<float4*>(Y + (long)co * a.N + pixs[p]) = v;
                    acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            t = 0;
            ++ci;
        }
    }
}

template <int KT, int CT, int NS = 3, int NPT = 16, int WP = 2>
static int xres_launch(const XresArgs& a, int B, hipStream_t stream) {
    const size_t lds = ((size_t)2 * KT * NPT * 128 + (size_t)NS * CT * 512 + 32 * KT) * sizeof(float);
    IRM_ALLOW_BIG_LDS((&gemm_xres_kernel<KT, CT, NS, WP, NPT>));
    hipLaunchKernelGGL((gemm_xres_kernel<KT, CT, NS, WP, NPT>), dim3((a.N + NPT * 16 - 1) / (NPT * 16), 1, B),
                       dim3((NPT / WP + 1) * 64), lds, stream, a);
    return irm_launch_status();
}

// called by irm_gemm1x1_f16x3_f32 (gemm_pw.hip) for K <= 384; returns IRM_EINVAL for shapes it does not cover
int irm_gemm_xres_dispatch(const float* wp, const float* x, long x_bs, float* y, long y_bs, const float* bias,
                           const float* stats, const float* lnw, const float* lnb, int ln_mode, int act, int B, int M,
                           int K, int N, hipStream_t stream) {
    XresArgs a;
    a.Wp = wp; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.bias = bias; a.stats = stats; a.lnw = lnw; a.lnb = lnb;
    a.M = M; a.K = K; a.N = N; a.mtiles = (M + 15) / 16; a.stages = (K + 15) / 16; a.ln_mode = ln_mode; a.act = act;
    a.dbg = irm_probe_int("IRM_XRES_DBG", 0);
    // tiles per pass: 9 when it divides the tile count (144, 288 channels), else 8
    const bool nine = a.mtiles % 9 == 0;
    switch (a.stages) {
        case 2: return nine ? xres_launch<2, 9>(a, B, stream) : xres_launch<2, 8>(a, B, stream);
        case 4: return nine ? xres_launch<4, 9>(a, B, stream) : xres_launch<4, 8>(a, B, stream);
        case 6:
            if (irm_probe_set("IRM_XRES_DEEP")) return xres_launch<6, 4, 6>(a, B, stream);      // experiment: 4-tile stages, 6-deep ring
            return nine ? xres_launch<6, 9>(a, B, stream) : xres_launch<6, 8>(a, B, stream);
        case 12:
            if (M < 512) return IRM_EINVAL;      // few output tiles per converted input: the streaming kernel is faster
            return nine ? xres_launch<12, 9, 3, 8, 2>(a, B, stream) : xres_launch<12, 8, 3, 8, 2>(a, B, stream);
        default: return IRM_EINVAL;      // odd stage counts (K = 48), K = 384 (64-pixel tiles: measured slower than
                                         // streaming) and other sizes: the streaming kernel
    }
}
