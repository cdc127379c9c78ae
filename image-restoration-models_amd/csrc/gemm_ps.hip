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

// LayerNorm + power-of-two scale + fp16 hi/lo split of ONE pixel's channels as lane (i, g) of a wave holds them (channels
// 32 ks + 8 g + e; the other three lanes of the pixel are 16, 32, 48 lanes away).  Every operation is an explicitly rounded
// intrinsic (no fma contraction left to the compiler): ln_split_kernel and the LN-fused GEMM run this same function.  lw / lb: LDS, LayerNorm weight / bias x operand scale.
template <int KS>
__device__ __forceinline__ void ps_ln_pixel(const float (&v)[KS][8], int K, int g, const float* lw, const float* lb, int ln_mode,
                                            float eps, ps_h8 (&hi)[KS], ps_h8 (&lo)[KS]) {
    float s = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s = __fadd_rn(s, (ks * 32 + 8 * g + e < K) ? v[ks][e] : 0.0f);
    s = __fadd_rn(s, __shfl_xor(s, 16));
    s = __fadd_rn(s, __shfl_xor(s, 32));
    const float mean = __fdiv_rn(s, (float)K);
    float q = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = __fsub_rn(v[ks][e], mean);
            q = __fadd_rn(q, (ks * 32 + 8 * g + e < K) ? __fmul_rn(d, d) : 0.0f);
        }
    q = __fadd_rn(q, __shfl_xor(q, 16));
    q = __fadd_rn(q, __shfl_xor(q, 32));
    const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(q, (float)K), eps)));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float4 w0 = *reinterpret_cast<const float4*>(lw + ks * 32 + 8 * g);
        const float4 w1 = *reinterpret_cast<const float4*>(lw + ks * 32 + 8 * g + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(lb + ks * 32 + 8 * g);
        const float4 b1 = *reinterpret_cast<const float4*>(lb + ks * 32 + 8 * g + 4);
        const float we[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float ys[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y;
            if (ln_mode == IRM_LN_WITHBIAS) y = __fmaf_rn(__fmul_rn(__fsub_rn(v[ks][e], mean), rstd), we[e], be[e]);
            else y = __fmul_rn(__fmul_rn(v[ks][e], rstd), we[e]);      // BiasFree: the mean only enters the variance
            // finite for any input (the scale leaves 16x headroom over the typical bound; see _hip.ln_split_scale)
            ys[e] = irm_sat_h(y);
        }
        // hi and lo from the SAME rounded fp32 value (irm_split2 takes it as an opaque register operand)
        irm_split8(ys, hi[ks], lo[ks]);
    }
}

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
    IRM_KERNEL_ENTRY();
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
    ps_h8 hi[KS], lo[KS];
    ps_ln_pixel<KS>(v, a.K, g, lw, lb, a.ln_mode, a.eps, hi, lo);
    _Float16* out = a.xs + (long)pt * KS * 1024 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<ps_h8*>(out + ks * 1024) = hi[ks];
        *reinterpret_cast<ps_h8*>(out + ks * 1024 + 512) = lo[ks];
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
    // LNF kernels (LayerNorm + split inside the GEMM's operand load: xs is not used)
    const float* x; long x_bs;     // [B][K][N] fp32
    const float* lnw; const float* lnb;
    int ln_mode; float eps, x_scale;
    int dbg;                       // IRM_PS_DBG (-DIRM_PROBES builds, timing only): 1 = stores to the dump page, 2 = no MFMAs, 4 = no weight DMA
    int H, W;                      // YCL kernels: the image is H x W (N = H W), whole 8 x 32 tiles
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
// LNF: the wave normalises and splits its own pixel tiles while loading them (ps_ln_pixel, the routine of ln_split_kernel)
// instead of reading fragments: no ln_split launch, no write + read of xs.  Pays where every workgroup does it once
// (one round, mgroups 1: the K 192 shapes).
// YCL (round 3): y tile-major channel-last in chunks of 64 channels [8 x 32 tile][M / 64][pixel][64] (M % 64 == 0; irm_hip.h) for the GDFN tail kernel
// (fused_tail.hip): the MFMA operands are swapped (weights = A), so lane (pixel r, g) ends up with channels 4 g .. 4 g + 3 of
// an output tile - one 16-byte store per tile and pixel; the store count per wave and chunk (the vmcnt arithmetic) is unchanged.
template <int KS, int WP, int CT, int CG, int NW, bool LNF = false, bool YCL = false>
__global__ __launch_bounds__(NW * 64, 2) void gemm_ps_kernel(PsArgs a) {
    IRM_KERNEL_ENTRY();
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
        if constexpr (YCL) {
            const int n = (pt_idx[p] - bi * ptl) * 16 + r, yy = n / a.W, xx = n - yy * a.W;
            // [tile][64-channel chunk][256 pixels][64 channels]: a workgroup's chunk of CT = 4 tiles is ONE contiguous run
            // of 256 bytes per pixel x its consecutive pixels (16 KiB per pixel tile), not 256-byte pieces 4 KiB apart
            const long tl = (long)((yy >> 3) * (a.W >> 5) + (xx >> 5)), pin = ((yy & 7) << 5) + (xx & 31);
            ybase[p] = pt < a.npt ? a.y + (long)bi * a.y_bs + tl * 256 * a.M + pin * 64 + g * 4 : nullptr;
        }
    }
    const int nchunks = (a.mtiles + CT - 1) / CT;
    const int c0 = mgi * a.cpg;
    const int nc = min(c0 + a.cpg, nchunks) - c0;  // chunks of this workgroup (>= 1: the host sends no empty group)

    // ---- resident A operands: the wave's WP pixel tiles, all KS k-steps, hi and lo
    ps_h8 xh[KS][WP], xl[KS][WP];
    if constexpr (LNF) {
        float* lw = lbias + a.cpg * CT * 16;       // [2][32 KS] LayerNorm weight, bias x operand scale
        for (int k = tid; k < KS * 32; k += NW * 64) {
            lw[k] = a.lnw[k] * a.x_scale;
            lw[KS * 32 + k] = a.ln_mode == IRM_LN_WITHBIAS ? a.lnb[k] * a.x_scale : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int bi = pt_idx[p] / ptl;
            const float* xp = a.x + (long)bi * a.x_bs + (pt_idx[p] - bi * ptl) * 16 + r;
            float v[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[ks][e] = xp[(long)(ks * 32 + 8 * g + e) * a.N];
            ps_h8 hi[KS], lo[KS];
            ps_ln_pixel<KS>(v, KS * 32, g, lw, lw + KS * 32, a.ln_mode, a.eps, hi, lo);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { xh[ks][p] = hi[ks]; xl[ks][p] = lo[ks]; }
        }
    } else {
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
#elif defined(PS_ROTATE_PER_WG)
    const int rot = pb % nc;
#else
    // per XCD, not per workgroup: the workgroups of an XCD (neighbouring pixel blocks) keep one chunk order, so that
    // together they write long contiguous runs of every output row and share each weight line through their L2; the
    // eight XCDs start an eighth of the matrix apart
    const int rot = (a.nblk & 7) == 0 ? (int)(((long)(pb / (a.nblk >> 3)) * nc) >> 3) : pb % nc;
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
                        acc[p][jg * CG + cc] = YCL ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[set][cc], xl[ks][p], acc[p][jg * CG + cc], 0, 0, 0)
                                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[ks][p], wh[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < WP; ++p)
                        acc[p][jg * CG + cc] = YCL ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[set][cc], xh[ks][p], acc[p][jg * CG + cc], 0, 0, 0)
                                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks][p], wl[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < WP; ++p)
                        acc[p][jg * CG + cc] = YCL ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[set][cc], xh[ks][p], acc[p][jg * CG + cc], 0, 0, 0)
                                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks][p], wh[set][cc], acc[p][jg * CG + cc], 0, 0, 0);
            }
        });
        // ---- chunk epilogue: exactly ST store instructions per wave (masked rows go to the dump page)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (YCL) {
                const int cb = (cm(c) * CT + ct) * 16;                       // (M % 16 == 0: whole tiles)
                const bool okt = cb < a.M && !IRM_DBG(a.dbg, 1);
                const f32x4 bv4 = *reinterpret_cast<const f32x4*>(lbias + ((cm(c) - c0) * CT + ct) * 16 + 4 * g);
#pragma unroll
                for (int p = 0; p < WP; ++p) {
                    const float4 v = make_float4(fmaf(acc[p][ct][0], a.out_scale, bv4[0]), fmaf(acc[p][ct][1], a.out_scale, bv4[1]),
                                                 fmaf(acc[p][ct][2], a.out_scale, bv4[2]), fmaf(acc[p][ct][3], a.out_scale, bv4[3]));
                    float4* dst = (okt && ybase[p]) ? reinterpret_cast<float4*>(ybase[p] + (long)(cb >> 6) * (256 * 64) + (cb & 63))
                                                    : ps_dump + ((blockIdx.x & 255) * 64 + lane);
                    *dst = v;
                    acc[p][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                continue;
            }
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

template <int KS, int WP, int CT, int CG, int NW, bool LNF = false, bool YCL = false>
static int ps_launch(PsArgs a, hipStream_t stream) {
    a.nblk = (a.npt + NW * WP - 1) / (NW * WP);
    const size_t lds = (size_t)4 * 2 * CT * 1024 + (size_t)a.cpg * CT * 64 + (LNF ? (size_t)KS * 256 : 0);
    IRM_ALLOW_BIG_LDS((&gemm_ps_kernel<KS, WP, CT, CG, NW, LNF, YCL>));
    hipLaunchKernelGGL((gemm_ps_kernel<KS, WP, CT, CG, NW, LNF, YCL>), dim3(a.nblk * a.mgroups), dim3(NW * 64), lds, stream, a);
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
    a.x = nullptr; a.x_bs = 0; a.lnw = nullptr; a.lnb = nullptr; a.ln_mode = 0; a.eps = 0.0f; a.x_scale = 1.0f;
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

// LayerNorm + 1x1 conv in ONE launch for K = 192 (4 waves x 3 pixel tiles, ct 4): irm_ln_split_f16's arithmetic inside the
// operand load of irm_gemm_presplit_f16x3_f32 - no xs tensor (results agree with the pair to a few fp32 ulps).  x [B][192][N] fp32, N % 16 == 0.
extern "C" int irm_ln_gemm_presplit_f16x3_f32(const void* wps, const float* x, long x_bs, const float* lnw, const float* lnb,
                                              int ln_mode, float x_scale, float eps, float* y, long y_bs, const float* bias,
                                              float out_scale, int B, int M, int K, int N, int mgroups, hipStream_t stream) {
    if (!wps || !x || !lnw || !y || B <= 0 || M <= 0 || N <= 0 || mgroups <= 0 || K != 192) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if (ln_mode == IRM_LN_WITHBIAS && !lnb) return IRM_EINVAL;
    if ((N & 15) || (y_bs & 3) || !irm_aligned16(y) || !irm_aligned16(wps) || !(x_scale > 0.0f)) return IRM_EINVAL;
    PsArgs a;
    a.xs = nullptr; a.wps = reinterpret_cast<const _Float16*>(wps); a.bias = bias;
    a.y = y; a.y_bs = y_bs; a.M = M; a.N = N; a.mtiles = (M + 15) / 16;
    a.npt = (int)((long)B * N / 16); a.nblk = 0;
    const int nchunks = (a.mtiles + 3) / 4;
    a.mgroups = mgroups; a.cpg = (nchunks + mgroups - 1) / mgroups;
    if ((long)(mgroups - 1) * a.cpg >= nchunks) return IRM_EINVAL;
    a.out_scale = out_scale; a.dbg = 0;
    a.x = x; a.x_bs = x_bs; a.lnw = lnw; a.lnb = lnb; a.ln_mode = ln_mode; a.eps = eps; a.x_scale = x_scale; a.H = 0; a.W = 0;
    return ps_launch<6, 3, 4, 2, 4, true>(a, stream);
}

// The same launch with y written tile-major channel-last in chunks of 64 channels [H/8 * W/32 tiles][M / 64][256 pixels][64]
// (header): N = H W, H % 8 == 0, W % 32 == 0, M % 64 == 0 - the layout irm_gdfn_tail_f16x3_f32 reads.
extern "C" int irm_ln_gemm_presplit_cl_f16x3_f32(const void* wps, const float* x, long x_bs, const float* lnw, const float* lnb,
                                                 int ln_mode, float x_scale, float eps, float* y, long y_bs, const float* bias,
                                                 float out_scale, int B, int M, int K, int H, int W, int mgroups,
                                                 hipStream_t stream) {
    if (!wps || !x || !lnw || !y || B <= 0 || M <= 0 || H <= 0 || W <= 0 || mgroups <= 0 || K != 192) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if (ln_mode == IRM_LN_WITHBIAS && !lnb) return IRM_EINVAL;
    if ((H & 7) || (W & 31) || (M & 63) || (y_bs & 3) || !irm_aligned16(y) || !irm_aligned16(wps) || !(x_scale > 0.0f)) return IRM_EINVAL;
    PsArgs a;
    a.xs = nullptr; a.wps = reinterpret_cast<const _Float16*>(wps); a.bias = bias;
    a.y = y; a.y_bs = y_bs; a.M = M; a.N = H * W; a.mtiles = (M + 15) / 16; a.H = H; a.W = W;
    a.npt = (int)((long)B * a.N / 16); a.nblk = 0;
    const int nchunks = (a.mtiles + 3) / 4;
    a.mgroups = mgroups; a.cpg = (nchunks + mgroups - 1) / mgroups;
    if ((long)(mgroups - 1) * a.cpg >= nchunks) return IRM_EINVAL;
    a.out_scale = out_scale; a.dbg = 0;
    a.x = x; a.x_bs = x_bs; a.lnw = lnw; a.lnb = lnb; a.ln_mode = ln_mode; a.eps = eps; a.x_scale = x_scale;
    return ps_launch<6, 3, 4, 2, 4, true, true>(a, stream);
}

// ===============================================================================================================
// GDFN tail of the C >= 192 levels with pre-split operands (late round 2):
//   irm_dwconv3x3_gate_split_f16 : g = gelu(dw(h1)) * dw(h2) (restormer.py:84, 89-91) written as fp16 hi + lo of g * 2^-4 in
//       MFMA fragment order gs[pixel tile][k-step of 32 gate channels][hi | lo][64 lanes][8 halves] (channels beyond
//       hid: zero) - the bytes of the planar fp32 tensor irm_dwconv3x3_gate_f32 writes, but the GEMM that consumes them
//       needs no conversion and reads 1 KiB per instruction.  A workgroup = 32 gate channels x 8 rows x 32 columns:
//       thread (channel, column quad) slides the 3-row window of irm_dwconv3x3_f32 down 8 rows (128-byte row segments
//       per channel), parks its 32 results in LDS [pixel][channel], the waves then emit whole fragments.
//   irm_gemm_presplit_res_f16x3_f32 : y = res + bias + out_scale * (W gs)  (FeedForward.project_out + the block's
//       residual, restormer.py:86, 92, 148), K streamed: every wave owns WP pixel tiles and ALL output tiles (12 per
//       ring stage); its own gs fragments come straight from global memory into a register rotation (two k-steps
//       ahead), the weight fragments of a stage (12 tiles x hi/lo = 24 KiB) through a 3-deep LDS-DMA ring.
//       Counted vmcnt: at the top of a stage the wave's younger operations are one stage of DMA and one or two
//       k-steps of fragment loads - compile-time constants (stages past the end re-request the last one).
struct GateSplitArgs {
    const float* x; long x_bs;     // [B][2 hid][H][W]
    const float* w;                // [2 hid][9]
    const float* bias;             // [2 hid] or null
    _Float16* gs;                  // [B * N / 16][KS][2][64][8]
    int hid, H, W, KS, tiles_x;
    float scale;
};

__device__ __forceinline__ void gs_row(const float* plane, int row, int H, int W, int col, float (&r)[6]) {
    if (row < 0 || row >= H) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = 0.0f;
        return;
    }
    const float* p = plane + (long)row * W;
    const float4 v = *reinterpret_cast<const float4*>(p + col);
    r[0] = col > 0 ? p[col - 1] : 0.0f;
    r[1] = v.x; r[2] = v.y; r[3] = v.z; r[4] = v.w;
    r[5] = col + 4 < W ? p[col + 4] : 0.0f;
}

__device__ __forceinline__ void gs_apply(const float (&k)[9], const float (&r0)[6], const float (&r1)[6],
                                         const float (&r2)[6], float bias, float (&o)[4]) {
    // (the operation order of dw_apply in elementwise.hip: the two kernels give identical values)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = bias;
        s += k[0] * r0[i] + k[1] * r0[i + 1] + k[2] * r0[i + 2];
        s += k[3] * r1[i] + k[4] * r1[i + 1] + k[5] * r1[i + 2];
        s += k[6] * r2[i] + k[7] * r2[i + 1] + k[8] * r2[i + 2];
        o[i] = s;
    }
}

// CH gate channels x (1024 / CH) columns x 8 rows per workgroup: thread = (channel, column quad).  CH 8: a wave reads
// 512-byte row segments of one plane (the access pattern of irm_dwconv3x3_gate_f32) and the workgroup emits quarter
// fragments (256-byte runs); CH 32: 128-byte row segments, whole fragments.
template <int CH>
__global__ __launch_bounds__(256) void dwconv3x3_gate_split_kernel(GateSplitArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int CGS = 256 / CH, COLS = 4 * CGS;  // column quads / columns per workgroup
    constexpr int PXS = 8 * COLS + 4;              // floats per channel row of the LDS tile [channel][pixel] (+ pad)
    __shared__ __attribute__((aligned(16))) float tile[CH * PXS];
    const int tid = threadIdx.x;
    const int cgl = tid % CGS, chl = tid / CGS;
    const int b = blockIdx.z;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * 8, x0 = tx * COLS;
    const int c = blockIdx.y * CH + chl;           // gate channel
    const long plane = (long)a.H * a.W;
    const int col = x0 + 4 * cgl;
    const bool live = c < a.hid && col < a.W;
    if (live) {
        const float* xa = a.x + (long)b * a.x_bs + (long)c * plane;
        const float* xb = xa + (long)a.hid * plane;
        float ka[9], kb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { ka[i] = a.w[c * 9 + i]; kb[i] = a.w[(c + a.hid) * 9 + i]; }
        const float ba = a.bias ? a.bias[c] : 0.0f, bb = a.bias ? a.bias[c + a.hid] : 0.0f;
        float a0[6], a1[6], a2[6], b0[6], b1[6], b2[6];
        gs_row(xa, y0 - 1, a.H, a.W, col, a0);
        gs_row(xa, y0, a.H, a.W, col, a1);
        gs_row(xb, y0 - 1, a.H, a.W, col, b0);
        gs_row(xb, y0, a.H, a.W, col, b1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int y = y0 + r;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.H) {
                gs_row(xa, y + 1, a.H, a.W, col, a2);
                gs_row(xb, y + 1, a.H, a.W, col, b2);
                float u[4], v[4];
                gs_apply(ka, a0, a1, a2, ba, u);
                gs_apply(kb, b0, b1, b2, bb, v);
                o = make_float4(irm_gelu(u[0]) * v[0], irm_gelu(u[1]) * v[1], irm_gelu(u[2]) * v[2], irm_gelu(u[3]) * v[3]);
#pragma unroll
                for (int i = 0; i < 6; ++i) { a0[i] = a1[i]; a1[i] = a2[i]; b0[i] = b1[i]; b1[i] = b2[i]; }
            }
            *reinterpret_cast<float4*>(tile + chl * PXS + r * COLS + 4 * cgl) = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            *reinterpret_cast<float4*>(tile + chl * PXS + r * COLS + 4 * cgl) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // emission: unit = (pixel tile, lane group g8 of the workgroup's CH / 8, pixel i): 16 bytes of the hi and of the lo
    // fragment; consecutive threads -> consecutive pixels, then lane groups: 256-byte runs per (tile, part, lane group)
    constexpr int G8 = CH / 8, TILES = 8 * COLS / 16;
    const long ptl = plane >> 4;
    const int ks = (blockIdx.y * CH) >> 5, g8_0 = ((blockIdx.y * CH) & 31) >> 3;
#pragma unroll
    for (int t = 0; t < TILES * G8 * 16 / 256; ++t) {
        const int u = tid + 256 * t;
        const int i16 = u & 15, g8l = (u >> 4) % G8, lt = u / (16 * G8);
        const int row = lt / (COLS / 16), seg = lt - row * (COLS / 16);
        if (y0 + row >= a.H || x0 + seg * 16 >= a.W) continue;
        const float* src = tile + (8 * g8l) * PXS + row * COLS + seg * 16 + i16;
        ps_h8 h, l;
        float ys[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ys[e] = irm_sat_h(__fmul_rn(src[e * PXS], a.scale));
        irm_split8(ys, h, l);                      // one rounded fp32 value for both parts (irm_common.h)
        const long pt = (long)b * ptl + (((long)(y0 + row) * a.W + x0 + seg * 16) >> 4);
        _Float16* out = a.gs + ((pt * a.KS + ks) * 2) * 512 + ((g8_0 + g8l) * 16 + i16) * 8;
        *reinterpret_cast<ps_h8*>(out) = h;
        *reinterpret_cast<ps_h8*>(out + 512) = l;
    }
}

template <int CH>
static int gate_split_launch(GateSplitArgs a, int B, hipStream_t stream) {
    constexpr int COLS = 1024 / CH;
    a.tiles_x = (a.W + COLS - 1) / COLS;
    const int cgroups = a.KS * (32 / CH);          // channel groups incl. the zero channels up to 32 KS
    if (cgroups > 65535) return IRM_EINVAL;
    const dim3 grid((unsigned)(a.tiles_x * ((a.H + 7) / 8)), cgroups, B);
    hipLaunchKernelGGL(dwconv3x3_gate_split_kernel<CH>, grid, dim3(256), 0, stream, a);
    return irm_launch_status();
}

// ch: gate channels per workgroup (8, 16 or 32; 0 = by row length: whole 512-byte row segments where W >= 128)
extern "C" int irm_dwconv3x3_gate_split_f16(const float* x, long x_bs, const float* w, const float* bias, void* gs,
                                            float scale, int B, int hid, int H, int W, int ch, hipStream_t stream) {
    if (!x || !w || !gs || B <= 0 || hid <= 0 || H <= 0 || W <= 0 || B > 65535) return IRM_EINVAL;
    if ((W & 15) || (x_bs & 3) || !irm_aligned16(x) || !irm_aligned16(gs) || !(scale > 0.0f)) return IRM_EINVAL;
    GateSplitArgs a{x, x_bs, w, bias, reinterpret_cast<_Float16*>(gs), hid, H, W, (hid + 31) / 32, 0, scale};
    if (ch == 0) ch = W >= 128 ? 8 : (W >= 64 ? 16 : 32);
    switch (ch) {
        case 8: return gate_split_launch<8>(a, B, stream);
        case 16: return gate_split_launch<16>(a, B, stream);
        case 32: return gate_split_launch<32>(a, B, stream);
        default: return IRM_EINVAL;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct Ps2Args {
    const _Float16* xs;            // [npt][KS][2][64][8]
    const _Float16* wps;           // [mtiles][KS][2][64][8]
    const float* bias;             // [M] or null
    const float* res; long r_bs;   // [B][M][N] or null (may alias y)
    float* y; long y_bs;           // [B][M][N]
    int M, N, KS, mtiles, npt;
    float out_scale;
};

// NH = ring stages per k-step (12 output tiles each: M <= 192 NH, zero tiles beyond M are clamped), WP pixel tiles per
// wave, NW waves.
template <int NH, int WP, int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_ps2_kernel(Ps2Args a) {
    IRM_KERNEL_ENTRY();
    constexpr int NS = 3, MTS = 12, FR = 2 * MTS, STG = FR * 1024;
    constexpr int DPW = FR / NW;                   // DMA instructions per wave and stage
    constexpr int CG = 2, NG = MTS / CG;           // weight fragments in register groups of 2 tiles, two groups in flight
    static_assert(FR % NW == 0 && NG % 2 == 0, "shape");
    // Operation order of a wave: every stage issues [DMA(s + 2)] and, on the first stage of k-step k, [A(k + 2)] behind
    // it.  At the top of stage s the wave needs DMA(s) and A(k) - whichever was issued later has exactly one stage of
    // DMA and one k-step of fragment loads behind it (NH 1: D(s) A(s) | D(s+1) A(s+1); NH 2, first half: D(s) A(k+1)
    // D(s+1) with A(k) older; second half: D(s) | D(s+1) A(k+2)).
    constexpr int YOUNGER = DPW + 2 * WP;
    static_assert(NS * DPW + 8 * WP <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int nblk = gridDim.x;
    int pb = blockIdx.x;
    if ((nblk & 7) == 0) pb = (pb & 7) * (nblk >> 3) + (pb >> 3);     // XCD x: one contiguous eighth of the pixel range
    const int ptl = a.N >> 4;
    int pt_idx[WP];
    bool pt_ok[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) {
        const int pt = (pb * NW + wave) * WP + p;
        pt_ok[p] = pt < a.npt;
        pt_idx[p] = min(pt, a.npt - 1);
    }
    const int TOT = a.KS * NH;                     // ring stages; a.KS % 4 == 0 (host)

    auto issue = [&](int s) {                      // stage s = (k-step s / NH, tile half s % NH); past the end: the last one
        const int sc = min(s, TOT - 1);
        const int ks = sc / NH, half = sc - ks * NH;
        char* dst = smem + (s % NS) * STG;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int f = wave + NW * j;
            const int mt = min(half * MTS + (f >> 1), a.mtiles - 1);
            const _Float16* src = a.wps + (((long)mt * a.KS + ks) * 2 + (f & 1)) * 512 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
        }
    };
    ps_h8 xh[4][WP], xl[4][WP];                    // k-step ks lives in set ks % 4
    auto load_a = [&](int ks, int set) {
        const int kc = min(ks, a.KS - 1);
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const _Float16* xp = a.xs + ((long)pt_idx[p] * a.KS + kc) * 1024 + lane * 8;
            xh[set][p] = *reinterpret_cast<const ps_h8*>(xp);
            xl[set][p] = *reinterpret_cast<const ps_h8*>(xp + 512);
        }
    };
    f32x4 acc[WP][NH * MTS];
#pragma unroll
    for (int p = 0; p < WP; ++p)
#pragma unroll
        for (int c = 0; c < NH * MTS; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ps_h8 wh[2][CG], wl[2][CG];
    auto read_group = [&](int slot, int jg, int set) {
        const char* base = smem + slot * STG + jg * CG * 2048 + lane * 16;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            wh[set][c] = *reinterpret_cast<const ps_h8*>(base + c * 2048);
            wl[set][c] = *reinterpret_cast<const ps_h8*>(base + c * 2048 + 1024);
        }
    };

    // prologue in the operation order of the steady state: A(0) D(0) A(1) D(1)
    load_a(0, 0);
    issue(0);
    load_a(1, 1);
    issue(1);
    int s = 0;
    for (int k4 = 0; k4 < a.KS; k4 += 4) {
        ps_static_for<4>([&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            ps_static_for<NH>([&](auto h_c) {
                constexpr int half = decltype(h_c)::value;
                // own DMAs of stage s (and, older, the fragments of this k-step) have landed; everybody is done with s - 1
                ps_wait_vmcnt<YOUNGER>();
                asm volatile("s_barrier" ::: "memory");
                issue(s + 2);
                if (half == 0) load_a(k4 + kk + 2, (kk + 2) & 3);
                const int slot = s % NS;
                read_group(slot, 0, 0);
#pragma unroll
                for (int jg = 0; jg < NG; ++jg) {
                    if (jg + 1 < NG) read_group(slot, jg + 1, (jg + 1) & 1);
                    const int set = jg & 1;
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                        for (int p = 0; p < WP; ++p) {
                            f32x4& ac = acc[p][half * MTS + jg * CG + cc];
                            ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[kk][p], wh[set][cc], ac, 0, 0, 0);
                        }
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                        for (int p = 0; p < WP; ++p) {
                            f32x4& ac = acc[p][half * MTS + jg * CG + cc];
                            ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[kk][p], wl[set][cc], ac, 0, 0, 0);
                        }
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                        for (int p = 0; p < WP; ++p) {
                            f32x4& ac = acc[p][half * MTS + jg * CG + cc];
                            ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[kk][p], wh[set][cc], ac, 0, 0, 0);
                        }
                }
                ++s;
            });
        });
    }
    ps_wait_vmcnt<0>();                            // the two re-requested stages: no LDS-DMA in flight at exit
    // ---- epilogue: + bias + residual, 16-byte stores (pixels on the MFMA row index)
#pragma unroll
    for (int p = 0; p < WP; ++p) {
        if (!pt_ok[p]) continue;
        const int bi = pt_idx[p] / ptl;
        const long poff = (long)(pt_idx[p] - bi * ptl) * 16 + g * 4;
        float* Y = a.y + (long)bi * a.y_bs + poff;
        const float* R = a.res ? a.res + (long)bi * a.r_bs + poff : nullptr;
#pragma unroll
        for (int c = 0; c < NH * MTS; ++c) {
            const int co = c * 16 + r;
            if (co >= a.M) continue;
            const float bv = a.bias ? a.bias[co] : 0.0f;
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R) rv = *reinterpret_cast<const float4*>(R + (long)co * a.N);
            *reinterpret_cast<float4*>(Y + (long)co * a.N) =
                make_float4(fmaf(acc[p][c][0], a.out_scale, bv) + rv.x, fmaf(acc[p][c][1], a.out_scale, bv) + rv.y,
                            fmaf(acc[p][c][2], a.out_scale, bv) + rv.z, fmaf(acc[p][c][3], a.out_scale, bv) + rv.w);
        }
    }
}

template <int NH, int WP, int NW>
static int ps2_launch(const Ps2Args& a, hipStream_t stream) {
    const int nblk = (a.npt + NW * WP - 1) / (NW * WP);
    const size_t lds = (size_t)3 * 24 * 1024;
    IRM_ALLOW_BIG_LDS((&gemm_ps2_kernel<NH, WP, NW>));
    hipLaunchKernelGGL((gemm_ps2_kernel<NH, WP, NW>), dim3(nblk), dim3(NW * 64), lds, stream, a);
    return irm_launch_status();
}

// xs: fragments of the K = 32 KS input channels (irm_dwconv3x3_gate_split_f16; channels beyond the real ones zero);
// wps: W s_w (columns padded to K) split by the host in fragment order (_hip.pack_gemm_weight_presplit); out_scale =
// 1 / (s_w s_x).  M <= 384, KS % 4 == 0, N % 16 == 0.  wg_shape = 10 waves + pixel tiles per wave: M <= 192: 42 (default)
// or 82; M <= 384: 81 (default) or 41.
extern "C" int irm_gemm_presplit_res_f16x3_f32(const void* wps, const void* xs, float* y, long y_bs, const float* res,
                                               long r_bs, const float* bias, float out_scale, int B, int M, int KS, int N,
                                               int wg_shape, hipStream_t stream) {
    if (!wps || !xs || !y || B <= 0 || M <= 0 || M > 384 || KS <= 0 || (KS & 3) || N <= 0) return IRM_EINVAL;
    if ((N & 15) || (y_bs & 3) || (r_bs & 3) || !irm_aligned16(y) || !irm_aligned16(res) || !irm_aligned16(xs)
        || !irm_aligned16(wps)) return IRM_EINVAL;
    Ps2Args a{reinterpret_cast<const _Float16*>(xs), reinterpret_cast<const _Float16*>(wps), bias, res, r_bs, y, y_bs,
              M, N, KS, (M + 15) / 16, (int)((long)B * N / 16), out_scale};
    if (M <= 192) {
        if (wg_shape == 0 || wg_shape == 42) return ps2_launch<1, 2, 4>(a, stream);
        if (wg_shape == 82) return ps2_launch<1, 2, 8>(a, stream);
        return IRM_EINVAL;
    }
    if (wg_shape == 41) return ps2_launch<2, 1, 4>(a, stream);
    if (wg_shape == 0 || wg_shape == 81) return ps2_launch<2, 1, 8>(a, stream);      // (measured: 66 vs 73 us, M 384 on 6 x 64^2)
    return IRM_EINVAL;
}
