// LayerNorm + 1x1 conv for the C >= 192 levels (K = 192 / 384 input channels, 576 ... 2042 output channels) as an fp32
// emulation on the fp16 matrix cores with PRE-SPLIT operands.
//
//   irm_ln_split_f16 : x [B][K][N] fp32 -> LayerNorm over the channels of every pixel (statistics in registers, two
//       passes) -> * 2^e -> fp16 hi + lo (x s = hi + lo up to 2^-22) written in MFMA A-operand FRAGMENT order:
//       xs[pixel tile of 16][k-step of 32 channels][hi | lo][64 lanes][8 halves], lane = 16 g + i holds pixel i,
//       channels 32 ks + 8 g + e.  A fragment is 1 KiB and contiguous: a wave reads or writes it with ONE 16-byte
//       access per lane (the same bytes as the fp32 tensor it replaces: 2 + 2 bytes per element).
//   irm_gemm_presplit_f16x3_f32 : y = (W xs) / (s_w s_x) + bias.  A workgroup (4 or 8 waves) owns 128 pixels; every
//       wave loads the fragments of ITS pixel tiles straight into registers (96 VGPRs: they are the A operand of every
//       MFMA the wave issues; no LDS, no conversion, no LayerNorm arithmetic in the GEMM) and then sweeps its range of
//       output tiles: the host-split weights (same fragment order, L2 resident) stream through a 4-deep LDS ring by
//       LDS-DMA, every wave issuing its share; per 32 input channels and (pixel tile, output tile) three
//       v_mfma_f32_16x16x32_f16 (lo*hi, hi*lo, hi*hi) accumulate in fp32.  The weight fragments of the next tile group
//       are read from LDS while the MFMAs of the current group run (two register sets), vmcnt waits are counted exactly
//       (stores included: fully masked tiles store to a dump page, so every wave issues the same instruction count).
//       The output-tile range of a pixel block is split over `mgroups` workgroups so that the launch fills whole rounds
//       of the 256 CUs; the groups of one pixel block get neighbouring slots of one XCD (shared L2 lines of xs).
//   replaces: LayerNorm (restormer.py:25-70) + Attention.qkv / FeedForward.project_in (restormer.py:82,105) at the
//   C >= 192 levels (gemm_xres.hip / gemm_pw.hip remain the entry points for every other shape).
#include "irm_common.h"
#include <type_traits>
#include <utility>

typedef _Float16 ps_h8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------------------
struct LnSplitArgs {
    const float* x; long x_bs;     // [B][K][N]
    const float* lnw; const float* lnb;
    _Float16* xs;                  // [B * N / 16][KS][2][64][8]
    int K, N, npt;                 // npt = B * N / 16
    int ln_mode;
    float eps, scale;
};

template <int KS>
__global__ __launch_bounds__(256) void ln_split_kernel(LnSplitArgs a) {
    __shared__ __attribute__((aligned(16))) float lw[KS * 32], lb[KS * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < KS * 32; k += blockDim.x) {
        // the power-of-two operand scale rides on the LayerNorm weight and bias (exact)
        lw[k] = k < a.K ? a.lnw[k] * a.scale : 0.0f;
        lb[k] = (a.ln_mode == IRM_LN_WITHBIAS && k < a.K) ? a.lnb[k] * a.scale : 0.0f;
    }
    __syncthreads();
    const int pt = blockIdx.x * (blockDim.x >> 6) + wave;
    if (pt >= a.npt) return;
    const int i = lane & 15, g = lane >> 4;
    const int ptl = a.N >> 4;
    const int b = pt / ptl;
    const int n = (pt - b * ptl) * 16 + i;
    const float* xp = a.x + (long)b * a.x_bs + n;
    float v[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 32 + 8 * g + e;
            v[ks][e] = xp[(long)min(k, a.K - 1) * a.N];
        }
    // two-pass statistics over the K channels of the pixel (4 lanes hold them: i, i + 16, i + 32, i + 48)
    float s = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (ks * 32 + 8 * g + e < a.K) ? v[ks][e] : 0.0f;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)a.K;
    float q = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[ks][e] - mean;
            q += (ks * 32 + 8 * g + e < a.K) ? d * d : 0.0f;
        }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q / (float)a.K + a.eps);
    _Float16* out = a.xs + (long)pt * KS * 1024 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float4 w0 = *reinterpret_cast<const float4*>(lw + ks * 32 + 8 * g);
        const float4 w1 = *reinterpret_cast<const float4*>(lw + ks * 32 + 8 * g + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(lb + ks * 32 + 8 * g);
        const float4 b1 = *reinterpret_cast<const float4*>(lb + ks * 32 + 8 * g + 4);
        const float we[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        ps_h8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y;
            if (a.ln_mode == IRM_LN_WITHBIAS) y = fmaf((v[ks][e] - mean) * rstd, we[e], be[e]);
            else y = v[ks][e] * rstd * we[e];             // BiasFree: the mean only enters the variance
            // finite for any input (the scale leaves 16x headroom over the typical bound; see _hip.ln_split_scale)
            y = fminf(fmaxf(y, -65000.0f), 65000.0f);
            // one opaque fp32 value: hi and lo must be derived from the SAME rounded product (a fused
            // v_fma_mixlo_f16 would round the exact product instead and break the split on double-rounding ties)
            asm volatile("" : "+v"(y));
            const _Float16 hh = (_Float16)y;
            h[e] = hh;
            l[e] = (_Float16)(y - (float)hh);
        }
        *reinterpret_cast<ps_h8*>(out + ks * 1024) = h;
        *reinterpret_cast<ps_h8*>(out + ks * 1024 + 512) = l;
    }
}

extern "C" int irm_ln_split_f16(const float* x, long x_bs, const float* lnw, const float* lnb, int ln_mode, float scale,
                                float eps, void* xs, int B, int K, int N, hipStream_t stream) {
    if (!x || !lnw || !xs || B <= 0 || K <= 0 || N <= 0) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if (ln_mode == IRM_LN_WITHBIAS && !lnb) return IRM_EINVAL;
    if ((N & 15) || (K & 31) || K > 384 || !irm_aligned16(xs) || !(scale > 0.0f)) return IRM_EINVAL;
    LnSplitArgs a{x, x_bs, lnw, lnb, reinterpret_cast<_Float16*>(xs), K, N, (int)((long)B * N / 16), ln_mode, eps, scale};
    // waves per workgroup: 4 on large planes, fewer when the launch would not fill the chip otherwise
    const int wpb = a.npt >= 4096 ? 4 : (a.npt >= 1024 ? 2 : 1);
    const dim3 grid((a.npt + wpb - 1) / wpb), block(64 * wpb);
    switch (K / 32) {
        case 6: hipLaunchKernelGGL(ln_split_kernel<6>, grid, block, 0, stream, a); break;
        case 12: hipLaunchKernelGGL(ln_split_kernel<12>, grid, block, 0, stream, a); break;
        default: return IRM_EINVAL;
    }
    return irm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
struct PsArgs {
    const _Float16* xs;            // [B * N / 16][KS][2][64][8]
    const _Float16* wps;           // [mtiles][KS][2][64][8]
    const float* bias;             // [M] or null
    float* y; long y_bs;           // [B][M][N]
    int M, N, mtiles;
    int npt;                       // pixel tiles in the launch: B * N / 16
    int nblk;                      // pixel blocks (workgroups per output-tile group): ceil(npt / (NW WP))
    int mgroups, cpg;              // output-tile chunks (of CT tiles) per workgroup
    float out_scale;               // 1 / (s_w s_x)
    int dbg;                       // IRM_PS_DBG (-DIRM_PROBES builds, timing only): 1 = stores to the dump page, 2 = no MFMAs, 4 = no weight DMA
};

__device__ float4 ps_dump[256 * 64];

template <int... I, class F>
__device__ __forceinline__ void ps_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression
template <int N, class F>
__device__ __forceinline__ void ps_static_for(F&& f) {
    ps_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int N>
__device__ __forceinline__ void ps_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KS = k-steps of 32 input channels, WP = pixel tiles per wave, CT = output tiles per chunk (one ring stage = the weight
// fragments of CT tiles for one k-step), CG = tiles per register group (two groups in flight: CT / CG must be even).
// NW = waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU: while one loads its resident operands or stores a
// chunk, the other one keeps the matrix cores busy - the phases of a single lock-stepped workgroup do not overlap).
template <int KS, int WP, int CT, int CG, int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_ps_kernel(PsArgs a) {
    constexpr int NS = 4;                          // ring depth
    constexpr int FR = 2 * CT;                     // 1 KiB fragments per stage
    constexpr int DPW = (FR + NW - 1) / NW;        // DMA instructions per wave and stage (duplicates fill the last round)
    constexpr int STG = FR * 1024;                 // bytes per stage
    constexpr int NG = CT / CG;
    constexpr int ST = WP * CT;                    // store instructions per wave and chunk
    static_assert(CT % CG == 0 && NG % 2 == 0, "two register groups alternate");
    static_assert(3 * DPW + 2 * ST <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + NS * STG);      // [cpg * CT * 16] bias of this workgroup's channels

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;

    // workgroup -> (pixel block, output-tile group).  Consecutive workgroup ids go to consecutive XCDs: XCD x takes the
    // pixel blocks [x nblk / 8, (x + 1) nblk / 8) in order, so the workgroups that are resident on one XCD at a time
    // write neighbouring runs of every output plane (DRAM page locality of the 400 MB store stream) and read one
    // contiguous range of xs; the groups of a pixel block sit 8 apart in the launch order (same XCD, same time: shared
    // L2 lines of xs)
    int pb, mgi;
    {
        const int id = blockIdx.x, mg = a.mgroups;
        if ((a.nblk & 7) == 0) {
            const int q = id / (8 * mg), rem = id - q * 8 * mg;
#ifdef PS_OLD_MAP
            pb = q * 8 + (rem & 7);
#else
            pb = (rem & 7) * (a.nblk >> 3) + q;
#endif
            mgi = rem >> 3;
        } else {
            pb = id / mg;
            mgi = id - pb * mg;
        }
    }
    // pixel tiles are numbered through the whole batch (xs is one array of B N / 16 tiles): a workgroup's NW x WP
    // tiles may straddle two images; tiles beyond the last one (tail workgroup) read the last tile and store to the dump page
    const int ptl = a.N >> 4;                      // pixel tiles per image
    int pt_idx[WP];
    float* ybase[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) {
        const int pt = (pb * NW + wave) * WP + p;
        pt_idx[p] = min(pt, a.npt - 1);
        const int bi = pt_idx[p] / ptl;
        ybase[p] = pt < a.npt ? a.y + (long)bi * a.y_bs + (pt_idx[p] - bi * ptl) * 16 + g * 4 : nullptr;
    }
    const int nchunks = (a.mtiles + CT - 1) / CT;
    const int c0 = mgi * a.cpg;
    const int nc = min(c0 + a.cpg, nchunks) - c0;  // chunks of this workgroup (>= 1: the host sends no empty group)

    // ---- resident A operands: the wave's WP pixel tiles, all KS k-steps, hi and lo
    ps_h8 xh[KS][WP], xl[KS][WP];
    {
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const _Float16* xp = a.xs + (long)pt_idx[p] * KS * 1024 + lane * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xh[ks][p] = *reinterpret_cast<const ps_h8*>(xp + ks * 1024);
                xl[ks][p] = *reinterpret_cast<const ps_h8*>(xp + ks * 1024 + 512);
            }
        }
    }

    // The workgroups sweep their output-tile chunks in ROTATED order (workgroup i starts at chunk i mod nc): started
    // together on the same order, all 256 CUs would request the same few KiB of the weight fragments at every moment -
    // one or two L2 channels per XCD serving everybody - instead of spreading their reads over the whole matrix.
#ifdef PS_NO_ROTATE
    const int rot = 0;
#else
    const int rot = pb % nc;
#endif
    auto cm = [&](int c) { const int cc = c + rot; return c0 + (cc >= nc ? cc - nc : cc); };

    // stage (c, ks) -> ring slot (c KS + ks) % NS
    auto issue = [&](int c, int ks) {
        char* dst = smem + ((c * KS + ks) & (NS - 1)) * STG;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int f = (wave + NW * j) % FR;
            const int mt = min(cm(c) * CT + (f >> 1), a.mtiles - 1);
            const _Float16* src = a.wps + (((long)mt * KS + ks) * 2 + (f & 1)) * 512 + lane * 8;
            if (IRM_DBG(a.dbg, 4)) src = a.wps + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[WP][CT];
#pragma unroll
    for (int p = 0; p < WP; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    ps_h8 wh[2][CG], wl[2][CG];
    auto read_group = [&](int slot, int jg, int set) {
        const char* base = smem + slot * STG + jg * CG * 2048 + lane * 16;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            wh[set][c] = *reinterpret_cast<const ps_h8*>(base + c * 2048);
            wl[set][c] = *reinterpret_cast<const ps_h8*>(base + c * 2048 + 1024);
        }
    };

    // ---- prologue: three stages in flight (KS >= 6), the bias of this workgroup's channels parked in LDS, then ST
    // stores to the dump page so that the vmcnt arithmetic of the first chunk is that of every other chunk
    issue(0, 0); issue(0, 1); issue(0, 2);
    for (int k = tid; k < nc * CT * 16; k += NW * 64) {
        const int co = c0 * CT * 16 + k;
        lbias[k] = (a.bias && co < a.M) ? a.bias[co] : 0.0f;
    }
    ps_wait_vmcnt<0>();
    {
        float4* d = ps_dump + ((blockIdx.x & 255) * 64 + lane);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < ST; ++j) {
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d), "v"(z) : "memory");
        }
    }
    __syncthreads();
    read_group(0, 0, 0);

    // one chunk: KS stages.  LAST: the tail of the ring (no stage beyond the last one is requested or awaited)
    auto chunk = [&](auto last_tag, int c) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int it0 = c * KS;
        ps_static_for<KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            const int it = it0 + ks;
#pragma unroll
            for (int jg = 0; jg < NG; ++jg) {
                if (jg < NG - 1) {
                    read_group(it & (NS - 1), jg + 1, (jg + 1) & 1);
                } else {
                    // own DMAs of stage it + 1 have landed?  Younger operations of this wave: the DMAs of stage it + 2
                    // and the ST stores of the chunk that ended one (ks == 0) or two (ks == 1) stages ago (chunk 0:
                    // the prologue's dump stores)
                    constexpr bool dma2 = !LAST || ks + 2 < KS;
                    constexpr int younger = (dma2 ? DPW : 0) + ((ks == 0 || ks == 1) ? ST : 0);
                    ps_wait_vmcnt<younger>();
                    asm volatile("s_barrier" ::: "memory");   // stage it + 1 is complete; nobody reads stage it - 1 any more
                    if (!LAST || ks + 3 < KS) issue(c + (ks + 3) / KS, (ks + 3) % KS);
                    read_group((it + 1) & (NS - 1), 0, 0);     // (after the last stage: a stale slot, never used)
                }
                const int set = jg & 1;
                if (IRM_DBG(a.dbg, 2)) continue;
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < WP; ++p)
                        acc[p][jg * CG + cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[ks][p], wh[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < WP; ++p)
                        acc[p][jg * CG + cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks][p], wl[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < WP; ++p)
                        acc[p][jg * CG + cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks][p], wh[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
            }
        });
        // ---- chunk epilogue: exactly ST store instructions per wave (masked rows go to the dump page)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int co = (cm(c) * CT + ct) * 16 + r;
            const bool ok = co < a.M && !IRM_DBG(a.dbg, 1);
            const float bv = lbias[((cm(c) - c0) * CT + ct) * 16 + r];
#pragma unroll
            for (int p = 0; p < WP; ++p) {
                const float4 v = make_float4(fmaf(acc[p][ct][0], a.out_scale, bv), fmaf(acc[p][ct][1], a.out_scale, bv),
                                             fmaf(acc[p][ct][2], a.out_scale, bv), fmaf(acc[p][ct][3], a.out_scale, bv));
                float4* dst = (ok && ybase[p]) ? reinterpret_cast<float4*>(ybase[p] + (long)co * a.N)
                                               : ps_dump + ((blockIdx.x & 255) * 64 + lane);
                *dst = v;
                acc[p][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    for (int c = 0; c < nc - 1; ++c) chunk(std::false_type{}, c);
    chunk(std::true_type{}, nc - 1);
    ps_wait_vmcnt<0>();                            // no LDS-DMA may be in flight when the workgroup's LDS is released
}

template <int KS, int WP, int CT, int CG, int NW>
static int ps_launch(PsArgs a, hipStream_t stream) {
    a.nblk = (a.npt + NW * WP - 1) / (NW * WP);
    const size_t lds = (size_t)4 * 2 * CT * 1024 + (size_t)a.cpg * CT * 64;
    IRM_ALLOW_BIG_LDS((&gemm_ps_kernel<KS, WP, CT, CG, NW>));
    hipLaunchKernelGGL((gemm_ps_kernel<KS, WP, CT, CG, NW>), dim3(a.nblk * a.mgroups), dim3(NW * 64), lds, stream, a);
    return irm_launch_status();
}

// xs from irm_ln_split_f16 (scale s_x), wps: W s_w split by the host in the same fragment order (Python:
// _hip.pack_gemm_weight_presplit), out_scale = 1 / (s_w s_x).  K in {192, 384}; N % 16 == 0; mgroups >= 1 workgroups
// share the output tiles of a pixel block (no empty group).  wg_shape = 10 waves + pixel tiles per wave:
//   K 192: 42 (ct 6 / 8; 128 pixels, two workgroups per CU), 32 (ct 6 / 8; 96 pixels), 43 (ct 4; 192 pixels);
//   K 384: 81 (ct 6 / 8; 128 pixels, one workgroup per CU); 0 = the default of the K.
extern "C" int irm_gemm_presplit_f16x3_f32(const void* wps, const void* xs, float* y, long y_bs, const float* bias,
                                           float out_scale, int act, int B, int M, int K, int N, int ct, int mgroups,
                                           int wg_shape, hipStream_t stream) {
    if (!wps || !xs || !y || B <= 0 || M <= 0 || N <= 0 || mgroups <= 0) return IRM_EINVAL;
    if (K != 192 && K != 384) return IRM_EINVAL;
    if (act != IRM_ACT_NONE) return IRM_EINVAL;      // (qkv and project_in have no activation)
    if ((N & 15) || (y_bs & 3) || !irm_aligned16(y) || !irm_aligned16(xs) || !irm_aligned16(wps)) return IRM_EINVAL;
    if (wg_shape == 0) wg_shape = K == 192 ? 42 : 81;
    if (ct != 4 && ct != 6 && ct != 8) return IRM_EINVAL;
    PsArgs a;
    a.xs = reinterpret_cast<const _Float16*>(xs); a.wps = reinterpret_cast<const _Float16*>(wps); a.bias = bias;
    a.y = y; a.y_bs = y_bs; a.M = M; a.N = N; a.mtiles = (M + 15) / 16;
    a.npt = (int)((long)B * N / 16); a.nblk = 0;
    const int nchunks = (a.mtiles + ct - 1) / ct;
    a.mgroups = mgroups; a.cpg = (nchunks + mgroups - 1) / mgroups;
    if ((long)(mgroups - 1) * a.cpg >= nchunks) return IRM_EINVAL;          // an empty group
    a.out_scale = out_scale; a.dbg = irm_probe_int("IRM_PS_DBG", 0);
    if (K == 192) {
        if (wg_shape == 42 && ct == 8) return ps_launch<6, 2, 8, 4, 4>(a, stream);
        if (wg_shape == 42 && ct == 6) return ps_launch<6, 2, 6, 3, 4>(a, stream);
        if (wg_shape == 32 && ct == 8) return ps_launch<6, 2, 8, 4, 3>(a, stream);
        if (wg_shape == 32 && ct == 6) return ps_launch<6, 2, 6, 3, 3>(a, stream);
        if (wg_shape == 43 && ct == 4) return ps_launch<6, 3, 4, 2, 4>(a, stream);
        return IRM_EINVAL;
    }
    if (wg_shape == 41 && ct == 8) return ps_launch<12, 1, 8, 4, 4>(a, stream);
    if (wg_shape == 41 && ct == 6) return ps_launch<12, 1, 6, 3, 4>(a, stream);
    if (wg_shape == 81 && ct == 8) return ps_launch<12, 1, 8, 4, 8>(a, stream);
    if (wg_shape == 81 && ct == 6) return ps_launch<12, 1, 6, 3, 8>(a, stream);
    return IRM_EINVAL;
}
