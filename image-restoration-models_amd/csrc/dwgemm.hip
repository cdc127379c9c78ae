// Depth-wise 3x3 conv fused into the 1x1 conv that consumes it:
//     g[k] = gate ? gelu(dw(x[k])) * dw(x[k + K]) : dw(x[k]),      y = W g + bias (+ res)
// Restormer's GDFN tail (dwconv -> chunk -> gelu(x1)*x2 -> project_out, restormer.py:84-93) and the
// value half of MDTA (qkv_dwconv on v -> attn @ v -> project_out, restormer.py:118-131, with the folded
// per-sample matrix as W) without ever writing the depth-wise result to HBM.
//
// One workgroup owns an 8 x 32 pixel tile (a 4 x 16 patch per wave) and every output channel (M <= 96).  Stages of 4 input channels
// (one MFMA k-step; 8 planes when gated) arrive through an LDS-DMA ring exactly like conv3x3_ring_kernel:
// the halo image is 10 rows x 10 aligned 16-byte chunks per plane, border chunks read a zero page, and
// the stage's packed 1x1 weights and depth-wise coefficients ride along.  Each lane then evaluates the
// depth-wise stencil for the 4 vertically neighbouring pixels it feeds to the MFMA as A operands, so the gated
// activations live only in registers.
#include "irm_common.h"

__device__ __attribute__((aligned(16))) float dg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct DwGemmArgs {
    const float* Wp; long w_bs;        // packed [mtiles][ksteps][64] (per sample when w_bs != 0)
    const float* dwp;                  // [4*S][DWS] depth-wise coefficients, see irm_hip.h
    const float* X; long x_bs;         // [B][K or 2K][H][W]
    float* Y; long y_bs;               // [B][M][H][W]
    const float* R; long r_bs;
    const float* bias;
    float* stats_out; float eps;
    int M, K, H, W, mtiles, ksteps, tiles_x, tiles;
    int dbg;                           // IRM_DWGEMM_DBG: 1 = no DMA, 2 = no stencil/MFMA (profiling only)
};

template <int N>
__device__ __forceinline__ void dg_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LayerNorm statistics of the finished output pixels (same contract as irm_stats_from_acc in gemm_pw.hip):
// lane (g, j) holds channel 16c + j of PT pixel quads; pix[q] < 0 marks a quad outside the image.
template <int CT, int PT>
__device__ __forceinline__ void dg_stats(const float4 (&t)[PT][CT], int M, long N, int j, const long (&pix)[PT],
                                         float* st, float eps) {
    float sum[PT][4], sq[PT][4];
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sum[q][e] = 0.f; sq[q][e] = 0.f; }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const bool ok = c * 16 + j < M;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const float v[4] = {t[q][c].x, t[q][c].y, t[q][c].z, t[q][c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[q][e] += ok ? v[e] : 0.f;
        }
    }
    const float inv = 1.0f / (float)M;
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum[q][e] += __shfl_xor(sum[q][e], o);
            sum[q][e] *= inv;
        }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const bool ok = c * 16 + j < M;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const float v[4] = {t[q][c].x, t[q][c].y, t[q][c].z, t[q][c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[e] - sum[q][e];
                sq[q][e] += ok ? d * d : 0.f;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sq[q][e] += __shfl_xor(sq[q][e], o);
            sq[q][e] = 1.0f / sqrtf(sq[q][e] * inv + eps);
        }
        if (j == 0 && pix[q] >= 0) {
            *reinterpret_cast<float4*>(st + pix[q]) = make_float4(sum[q][0], sum[q][1], sum[q][2], sum[q][3]);
            *reinterpret_cast<float4*>(st + N + pix[q]) = make_float4(sq[q][0], sq[q][1], sq[q][2], sq[q][3]);
        }
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Packed-fp32 (v_pk_fma_f32) stencil: PT+2 rows x 3 columns around PT vertically neighbouring pixels, the
// outputs as register pairs (o0,o1)[, (o2,o3)].  kk[t] holds tap t twice (k,k); bias first, then the taps row
// by row, as dw_apply in elementwise.hip.  Lanes of a wave read consecutive columns of the halo image: no LDS
// bank conflicts.
template <int PT, int RF>
__device__ __forceinline__ void dg_stencil(const float* img, const v2f (&kk)[9], v2f bias, v2f (&o)[PT / 2]) {
    v2f rp[PT + 1][3];                             // rp[d][dx] = rows d and d+1 at column dx
#pragma unroll
    for (int d = 0; d < PT + 1; ++d)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) rp[d][dx] = (v2f){img[d * RF + dx], img[(d + 1) * RF + dx]};
#pragma unroll
    for (int h = 0; h < PT / 2; ++h) {
        v2f s = bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) s = kk[dy * 3 + dx] * rp[2 * h + dy][dx] + s;
        o[h] = s;
    }
}

// GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level): branch free,
// packed fp32 except the two transcendentals per element; the kernel is VALU bound on this function.
__device__ __forceinline__ v2f dg_gelu2(v2f x) {
    const v2f z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const v2f d = z * 0.3275911f + 1.0f;
    const v2f t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    v2f p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const v2f q = z * z * -1.4426950408889634f;
    const v2f e = p * t * (v2f){__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};   // 1 - erf(|z|)
    const v2f w = 1.0f - e;
    const v2f sg = {copysignf(w.x, x.x), copysignf(w.y, x.y)};
    const v2f h = x * 0.5f;
    return sg * h + h;
}

// PT = pixels per lane: 4 -> 4 waves, each a 4 x 16 patch; 2 -> 8 waves, each a 2 x 16 patch (half the
// accumulators per wave: <= 128 VGPRs, 16 waves per CU, for the 96-channel outputs)
// F16: the 1x1 part as an fp32 emulation on the fp16 matrix cores.  The lane's stencil outputs of 4 consecutive
// stages (channels 16G + 4j + g, j = 0..3 - exactly the k-slots lane (g, i) owns in a 16x16x16 MFMA) are kept in
// registers and split into fp16 hi + lo (scaled by 2^-4 so that gated activations up to 10^6 stay in range);
// every 4th stage three MFMAs per tile (lo*hi, hi*lo, hi*hi) accumulate in fp32.  The split weights of a
// 16-channel group (host packed: irm_gemm1x1_f16x3_f32's order) arrive in quarters with the 4 stages through one
// extra, partly masked DMA instruction per wave and stage, into a double-buffered area next to the ring.
typedef _Float16 dg_h4 __attribute__((ext_vector_type(4)));

// TW = tile width (32 or 64 pixels; 8 rows): the halo columns cost a 64-byte sector per row and side for one
// pixel each, so a 64-wide tile (8 waves) moves 1.5x instead of 2x the bytes of its rows
template <int CT, bool GATE, int NS, int PT, bool F16 = false, int TW = 32>
__global__ __launch_bounds__((PT == 4 ? 256 : 512) * (TW / 32), TW == 64 ? 1 : (PT == 4 ? (CT <= 3 ? 3 : 2) : 2))
void dwgemm_kernel(DwGemmArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int CH = TW / 4 + 2;                 // 16-byte chunks per halo row
    constexpr int RF = CH * 4, PL = 10 * RF;       // floats per halo row / plane
    constexpr int NWX = TW / 16;                   // wave columns
    constexpr int NT = (PT == 4 ? 256 : 512) * (TW / 32);   // threads per workgroup
    constexpr int NP = GATE ? 8 : 4;               // halo planes per stage
    constexpr int DWS = GATE ? 40 : 20;            // depth-wise coefficients per channel (floats, each twice)
    constexpr int XC = NP * 10 * CH, WC = F16 ? 0 : CT * 16;   // 16-byte chunks per stage (F16: weights bypass the ring)
    constexpr int TC = XC + WC + DWS;
    constexpr int R = (TC + NT - 1) / NT;          // DMA instructions per lane per stage
    constexpr int STG = R * NT * 4;                // floats per stage
    constexpr int RW = R + (F16 ? 1 : 0);          // DMA instructions per wave and stage
    static_assert((NS - 2) * RW <= 63, "vmcnt field");
    constexpr int LW = CT * 16 / (NT / 64);        // F16: weight pieces per wave and stage
    static_assert(!F16 || (PT == 4 && (CT * 16) % (NT / 64) == 0 && LW <= 64), "F16 variant: one masked weight instruction per wave");
    static_assert(TW == 32 || (PT == 4 && F16), "64-wide tiles: emulated variant only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* warea = smem + NS * STG;                // F16: [group parity][quarter][CT * 64] floats

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int b = blockIdx.y;
    // neighbouring tiles share halo rows: keep runs of consecutive tiles on one XCD (blocks are dealt
    // round-robin to the 8 XCDs) so that the shared rows hit in that XCD's L2
    const int per = (a.tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= a.tiles) return;
    const int ty0 = (tile / a.tiles_x) * 8, tx0 = (tile % a.tiles_x) * TW;
    const long plane = (long)a.H * a.W;
    const float* X = a.X + (long)b * a.x_bs;
    const float* Wp = a.Wp + (long)b * a.w_bs;
    const int S = (a.K + 3) >> 2;
    const int SL = F16 ? ((S + 3) & ~3) : S;       // F16: whole 16-channel groups (stages beyond S carry zeros)
    const int groups = a.ksteps >> 2;              // 16-channel groups of the packed weights

    // per-lane DMA sources: src(s) = s < lim ? base + s * stride : zero page
    const float* base[R];
    long stride[R];
    int lim[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int q = j * NT + tid;
        base[j] = dg_zero_page; stride[j] = 0; lim[j] = 0;
        if (q < XC) {
            const int pl = q / (10 * CH), rem = q - pl * (10 * CH), row = rem / CH, chunk = rem - row * CH;
            const int gy = ty0 - 1 + row, gx = tx0 - 4 + chunk * 4;
            const int ch = pl & 3, set = pl >> 2;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                base[j] = X + ((long)set * a.K + ch) * plane + (long)gy * a.W + gx;
                stride[j] = 4 * plane;
                lim[j] = (a.K - ch + 3) >> 2;
            }
        } else if (q < XC + WC) {
            const int wq = q - XC, ct = wq >> 4;
            if (ct < a.mtiles) {
                base[j] = Wp + (long)ct * a.ksteps * 64 + (wq & 15) * 4;
                stride[j] = 64;
                lim[j] = S;
            }
        } else if (q < TC) {
            base[j] = a.dwp + (q - XC - WC) * 4;
            stride[j] = 4 * DWS;
            lim[j] = S;
        }
    }

    // running source pointers (border / unused lanes: zero page, stride 0); only a ragged last stage
    // (K % 4 != 0) has to look at lim
    const bool ragged = (a.K & 3) != 0 || SL != S;
    auto issue = [&](int s) {
        float* dst = smem + (s % NS) * STG + wave * 256;
        const bool tail = ragged && s >= S - 1;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float* src = (tail && s >= lim[j]) ? dg_zero_page : base[j];
            base[j] += stride[j];
            if (!IRM_DBG(a.dbg, 1)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * 4), 16, 0, 0);
        }
        if constexpr (F16) {
            // quarter (s & 3) of the split weights of group s >> 2: CT*16 pieces of 16 bytes, CT*4 lanes per wave
            if (lane < LW) {
                const int cq = wave * LW + lane, ct = cq >> 4;
                const float* src = ct < a.mtiles
                    ? Wp + ((long)ct * groups + (s >> 2)) * 256 + (s & 3) * 64 + (cq & 15) * 4 : dg_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)(warea + ((s >> 2) & 1) * (CT * 256) + (s & 3) * (CT * 64) +
                                                              wave * (LW * 4)), 16, 0, 0);
            }
        }
    };

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < SL) issue(j);

    // accumulators start from bias + residual: lane (g, i) holds channel 16c + i of the pixel quads q < PT,
    // row PT*(wave>>1) + q, columns 16*(wave&1) + 4*g + [0,4)
    float* Y = a.Y + (long)b * a.y_bs;
    const float* Rp = a.R ? a.R + (long)b * a.r_bs : nullptr;
    long pix[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int y = ty0 + (wave / NWX) * PT + q, x = tx0 + 16 * (wave % NWX) + 4 * g;
        pix[q] = (y < a.H && x < a.W) ? (long)y * a.W + x : -1;
    }
    f32x4 acc[PT][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int co = c * 16 + i;
        const bool row_ok = co < a.M;
        const float bv = (a.bias && row_ok) ? a.bias[co] : 0.0f;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Rp) {       // unconditional (clamped) loads: all of them are in flight together
                const bool ok = row_ok && pix[q] >= 0;
                rr = *reinterpret_cast<const float4*>(Rp + (ok ? (long)co * plane + pix[q] : 0));
                if (!ok) rr = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            constexpr float sc = F16 ? 0.0625f : 1.0f;     // F16 accumulates in units of 2^-4
            acc[q][c] = (f32x4){(rr.x + bv) * sc, (rr.y + bv) * sc, (rr.z + bv) * sc, (rr.w + bv) * sc};
        }
    }

    // A-operand pixels of this lane: column 16*(wave&1) + i, rows PT*(wave>>1) + p, p < PT (a wave owns a
    // PT x 16 patch); img_off = top-left tap of the first pixel in the halo image
    const int img_off = g * PL + (wave / NWX) * (PT * RF) + 16 * (wave % NWX) + i + 3;

    float ag[4][PT];                               // F16: stencil outputs of the 4 stages of a group
    for (int s = 0; s < SL; ++s) {
        const int rem = min(NS - 2, SL - 1 - s);
        if (rem >= NS - 2 && NS >= 3) dg_wait_vmcnt<(NS - 2) * RW>();
        else dg_wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");
        if (s + NS - 1 < SL) issue(s + NS - 1);

        if (IRM_DBG(a.dbg, 2)) continue;
        const float* xb = smem + (s % NS) * STG;
        const float* wb = xb + XC * 4;
        const v2f* dk = reinterpret_cast<const v2f*>(wb + WC * 4 + g * DWS);
        float af[PT];
        {
            v2f ka[9], oa[PT / 2];
#pragma unroll
            for (int t = 0; t < 9; ++t) ka[t] = dk[t];
            dg_stencil<PT, RF>(xb + img_off, ka, dk[9], oa);
            if (GATE) {
                v2f kb[9], ob[PT / 2];
#pragma unroll
                for (int t = 0; t < 9; ++t) kb[t] = dk[10 + t];
                dg_stencil<PT, RF>(xb + img_off + 4 * PL, kb, dk[19], ob);
#pragma unroll
                for (int h = 0; h < PT / 2; ++h) oa[h] = dg_gelu2(oa[h]) * ob[h];
            }
            #pragma unroll
            for (int h = 0; h < PT / 2; ++h) { af[2 * h] = oa[h].x; af[2 * h + 1] = oa[h].y; }
        }
        if constexpr (F16) {
            // park this stage's outputs in slot s & 3 (uniform select: the loop is not unrolled by 4)
            const int j4 = s & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < PT; ++p) ag[j][p] = j4 == j ? af[p] : (j4 < j ? 0.0f : ag[j][p]);
            if (j4 == 3) {
                dg_h4 ah[PT], al[PT];
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = irm_sat_h(ag[j][p] * 0.0625f);
                        ah[p][j] = (_Float16)x;
                        al[p][j] = (_Float16)(x - (float)ah[p][j]);
                    }
                const float* wa = warea + ((s >> 2) & 1) * (CT * 256) + (lane >> 5) * (CT * 64) + (lane & 31) * 2;
                dg_h4 bh[CT], bl[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    bh[c] = *reinterpret_cast<const dg_h4*>(wa + c * 64);
                    bl[c] = *reinterpret_cast<const dg_h4*>(wa + 2 * (CT * 64) + c * 64);
                }
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
            }
        } else {
            // (the f32 MFMA shares the SIMD's fp32 datapath with the VALU: interleaving the two streams inside
            // a wave or across waves buys nothing, the kernel costs VALU + MFMA time)
            float bf[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) bf[c] = wb[c * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][c] = irm_mfma16(af[p], bf[c], acc[p][c]);
        }
    }

    float4 t[PT][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            constexpr float sc = F16 ? 16.0f : 1.0f;
            t[q][c] = make_float4(acc[q][c][0] * sc, acc[q][c][1] * sc, acc[q][c][2] * sc, acc[q][c][3] * sc);
        }
    if (a.stats_out) dg_stats<CT, PT>(t, a.M, plane, i, pix, a.stats_out + (long)b * 2 * plane, a.eps);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int co = c * 16 + i;
        if (co >= a.M) continue;
#pragma unroll
        for (int q = 0; q < PT; ++q)
            if (pix[q] >= 0) *reinterpret_cast<float4*>(Y + (long)co * plane + pix[q]) = t[q][c];
    }
}

template <int CT, bool GATE, int PT, bool F16 = false, int TW = 32>
static int dg_launch(DwGemmArgs a, int B, hipStream_t stream) {
    constexpr int NS = 3;                          // deeper rings (4..6) measured no faster: occupancy matters more
    constexpr int NT = (PT == 4 ? 256 : 512) * (TW / 32);
    constexpr int TC = (GATE ? 8 : 4) * 10 * (TW / 4 + 2) + (F16 ? 0 : CT * 16) + (GATE ? 40 : 20);
    constexpr int R = (TC + NT - 1) / NT;
    const size_t lds = (size_t)NS * R * NT * 16 + (F16 ? 2 * CT * 1024 : 0);
    IRM_ALLOW_BIG_LDS((&dwgemm_kernel<CT, GATE, NS, PT, F16, TW>));
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles = a.tiles_x * ((a.H + 7) / 8);
    const int per = (a.tiles + 7) >> 3;
    hipLaunchKernelGGL((dwgemm_kernel<CT, GATE, NS, PT, F16, TW>), dim3(per * 8, B), dim3(NT), lds, stream, a);
    return irm_launch_status();
}

static int dwgemm_entry(const float* wp, long w_bs, const float* dwp, const float* x, long x_bs, float* y,
                        long y_bs, const float* res, long r_bs, const float* bias, int gate, int B, int M,
                        int K, int H, int W, float* stats_out, float eps, bool split, hipStream_t stream) {
    if (!wp || !dwp || !x || !y || B <= 0 || M <= 0 || K <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (M > 96 || (W & 3) || B > 65535) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || (r_bs & 3) || (w_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) ||
        !irm_aligned16(res) || !irm_aligned16(wp) || !irm_aligned16(dwp) || !irm_aligned16(stats_out))
        return IRM_EINVAL;
    DwGemmArgs a;
    a.Wp = wp; a.w_bs = w_bs; a.dwp = dwp; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.R = res; a.r_bs = r_bs;
    a.bias = bias; a.stats_out = stats_out; a.eps = eps;
    a.M = M; a.K = K; a.H = H; a.W = W;
    a.mtiles = (M + 15) / 16; a.ksteps = 4 * ((K + 15) / 16);
    a.tiles_x = (W + 31) / 32;
    a.tiles = a.tiles_x * ((H + 7) / 8);
    a.dbg = irm_probe_int("IRM_DWGEMM_DBG", 0);
    if (split) {
        // 64-wide tiles (IRM_DWGEMM_TW=64): fewer halo sectors, but measured 5-8 % slower than 32-wide ones
        static const int tw_env = irm_probe_int("IRM_DWGEMM_TW", 0);
        const bool wide = tw_env == 64;
        if (wide) {
            if (a.mtiles <= 3) return gate ? dg_launch<3, true, 4, true, 64>(a, B, stream) : dg_launch<3, false, 4, true, 64>(a, B, stream);
            return gate ? dg_launch<6, true, 4, true, 64>(a, B, stream) : dg_launch<6, false, 4, true, 64>(a, B, stream);
        }
        if (a.mtiles <= 3) return gate ? dg_launch<3, true, 4, true>(a, B, stream) : dg_launch<3, false, 4, true>(a, B, stream);
        return gate ? dg_launch<6, true, 4, true>(a, B, stream) : dg_launch<6, false, 4, true>(a, B, stream);
    }
    static const int pt_env = irm_probe_int("IRM_DWGEMM_PT", 0);
    const int pt = pt_env == 2 ? 2 : 4;      // 8 waves x 2 pixels: twice the occupancy, measured no faster
    if (a.mtiles <= 3) {
        if (pt == 4) return gate ? dg_launch<3, true, 4>(a, B, stream) : dg_launch<3, false, 4>(a, B, stream);
        return gate ? dg_launch<3, true, 2>(a, B, stream) : dg_launch<3, false, 2>(a, B, stream);
    }
    if (pt == 4) return gate ? dg_launch<6, true, 4>(a, B, stream) : dg_launch<6, false, 4>(a, B, stream);
    return gate ? dg_launch<6, true, 2>(a, B, stream) : dg_launch<6, false, 2>(a, B, stream);
}

extern "C" int irm_dwgemm_f32(const float* wp, long w_bs, const float* dwp, const float* x, long x_bs, float* y,
                              long y_bs, const float* res, long r_bs, const float* bias, int gate, int B, int M,
                              int K, int H, int W, float* stats_out, float eps, hipStream_t stream) {
    return dwgemm_entry(wp, w_bs, dwp, x, x_bs, y, y_bs, res, r_bs, bias, gate, B, M, K, H, W, stats_out, eps, false,
                        stream);
}

extern "C" int irm_dwgemm_f16x3_f32(const float* wp_split, long w_bs, const float* dwp, const float* x, long x_bs, float* y,
                                    long y_bs, const float* res, long r_bs, const float* bias, int gate, int B, int M,
                                    int K, int H, int W, float* stats_out, float eps, hipStream_t stream) {
    return dwgemm_entry(wp_split, w_bs, dwp, x, x_bs, y, y_bs, res, r_bs, bias, gate, B, M, K, H, W, stats_out, eps, true,
                        stream);
}
