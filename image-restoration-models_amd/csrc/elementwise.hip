// HBM-bound planar-NCHW kernels: per-pixel LayerNorm statistics and the
// depth-wise 3x3 convolutions (plain and GDFN-gated).
#include "irm_common.h"

// ---------------------------------------------------------------------------
// LayerNorm statistics over the channel axis (restormer.py:25-70): per pixel
// mean and rstd = 1/sqrt(biased var + eps).  One thread owns 4 consecutive
// pixels (16-byte loads) and runs Welford over its share of the C rows, so x is
// read exactly once.  A workgroup = PQ pixel quads x G = 256 / PQ channel groups
// (group g takes channels g, g + G, ...; the partials are merged in group order by
// Chan's formula).  PQ follows the plane size only (64 / 16 / 4 quads for large /
// medium / small planes - small planes need the channel axis for parallelism: a
// 32x32 level has 4 workgroups of 96-deep dependent loops otherwise), never the
// batch size: an image's statistics do not depend on what it is batched with.
template <bool VEC, int PQ>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, long x_bs,
                                                       float* __restrict__ stats, int C, int N, float eps) {
    IRM_KERNEL_ENTRY();
    constexpr int G = 256 / PQ;
    __shared__ float4 pm[G][PQ], pq[G][PQ];
    const int b = blockIdx.y;
    const int q = threadIdx.x % PQ, g = threadIdx.x / PQ;
    const int n = (blockIdx.x * PQ + q) * 4;
    const float* p = x + (long)b * x_bs;
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), m2 = mean;
    int cnt = 0;
    auto push = [&](const float4& v) {
        const float rc = 1.0f / (float)(++cnt);
        float d;
        d = v.x - mean.x; mean.x += d * rc; m2.x += d * (v.x - mean.x);
        d = v.y - mean.y; mean.y += d * rc; m2.y += d * (v.y - mean.y);
        d = v.z - mean.z; mean.z += d * rc; m2.z += d * (v.z - mean.z);
        d = v.w - mean.w; mean.w += d * rc; m2.w += d * (v.w - mean.w);
    };
    if (n < N) {
        int c = g;
        for (; c + 3 * G < C; c += 4 * G) {                  // four rows in flight
            const float4 v0 = irm_ld4<VEC>(p + (long)c * N, n, N);
            const float4 v1 = irm_ld4<VEC>(p + (long)(c + G) * N, n, N);
            const float4 v2 = irm_ld4<VEC>(p + (long)(c + 2 * G) * N, n, N);
            const float4 v3 = irm_ld4<VEC>(p + (long)(c + 3 * G) * N, n, N);
            push(v0); push(v1); push(v2); push(v3);
        }
        for (; c < C; c += G) push(irm_ld4<VEC>(p + (long)c * N, n, N));
    }
    pm[g][q] = mean;
    pq[g][q] = m2;
    __syncthreads();
    if (g != 0 || n >= N) return;
    // Chan's merge of (count, mean, M2) partials, fixed order 0, 1, ..., G - 1
    float na = (float)((C + G - 1) / G);             // channels seen by group 0
    for (int k = 1; k < G; ++k) {
        const int ck = (C - k + G - 1) / G;          // channels k, k + G, ... < C
        if (ck <= 0) break;
        const float nb = (float)ck, nt = na + nb;
        const float4 mb = pm[k][q], qb = pq[k][q];
        const float f = nb / nt, g2 = na * nb / nt;
        float d;
        d = mb.x - mean.x; mean.x += d * f; m2.x += qb.x + d * d * g2;
        d = mb.y - mean.y; mean.y += d * f; m2.y += qb.y + d * d * g2;
        d = mb.z - mean.z; mean.z += d * f; m2.z += qb.z + d * d * g2;
        d = mb.w - mean.w; mean.w += d * f; m2.w += qb.w + d * d * g2;
        na = nt;
    }
    const float inv = 1.0f / (float)C;
    float4 rstd;
    rstd.x = 1.0f / sqrtf(m2.x * inv + eps);
    rstd.y = 1.0f / sqrtf(m2.y * inv + eps);
    rstd.z = 1.0f / sqrtf(m2.z * inv + eps);
    rstd.w = 1.0f / sqrtf(m2.w * inv + eps);
    float* s = stats + (long)b * 2 * N;
    irm_st4<VEC>(s, n, N, mean);
    irm_st4<VEC>(s + N, n, N, rstd);
}

template <int PQ>
static void ln_stats_launch(const float* x, long x_bs, float* stats, int B, int C, int N, float eps, bool vec,
                            hipStream_t stream) {
    dim3 grid(((N + 3) / 4 + PQ - 1) / PQ, B);
    if (vec) hipLaunchKernelGGL((ln_stats_kernel<true, PQ>), grid, dim3(256), 0, stream, x, x_bs, stats, C, N, eps);
    else hipLaunchKernelGGL((ln_stats_kernel<false, PQ>), grid, dim3(256), 0, stream, x, x_bs, stats, C, N, eps);
}

extern "C" int irm_ln_stats_f32(const float* x, long x_bs, float* stats, int B, int C, int N, float eps,
                                hipStream_t stream) {
    if (!x || !stats || B <= 0 || C <= 0 || N <= 0 || B > 65535) return IRM_EINVAL;
    const bool vec = !(N & 3) && !(x_bs & 3) && irm_aligned16(x) && irm_aligned16(stats);
    if (N >= 65536) ln_stats_launch<64>(x, x_bs, stats, B, C, N, eps, vec, stream);
    else if (N >= 2048) ln_stats_launch<16>(x, x_bs, stats, B, C, N, eps, vec, stream);
    else ln_stats_launch<4>(x, x_bs, stats, B, C, N, eps, vec, stream);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
// Depth-wise 3x3, zero pad 1 (restormer.py:84,106; MaIR conv2d with bias+SiLU).
// A work item = 4 consecutive columns x RS rows of one channel plane; a
// 3-row register window slides down the strip so each input row is loaded once
// per strip (halo rows come from L2).  GATE fuses the GDFN gate
// (restormer.py:89-91): out[c] = gelu(dw(x[c])) * dw(x[c + hid]).
struct DwArgs {
    const float* x; long x_bs;
    const float* w;        // [C][9]
    const float* bias;     // [C] or null
    float* y; long y_bs;
    int C;                 // output channels (GATE: hid; input has 2*hid)
    int H, W, act;
};

template <bool VEC>
__device__ __forceinline__ void dw_load_row(const float* plane, int row, int H, int W, int col, float (&r)[6]) {
    if (row < 0 || row >= H) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = 0.0f;
        return;
    }
    const float* p = plane + (long)row * W;
    const float4 v = irm_ld4<VEC>(p, col, W);
    r[0] = col > 0 ? p[col - 1] : 0.0f;
    r[1] = v.x; r[2] = v.y; r[3] = v.z; r[4] = v.w;
    r[5] = col + 4 < W ? p[col + 4] : 0.0f;
}

// Branch-free row load (clamped address, values selected afterwards): all RS + 2 rows of a strip can be requested before
// the first one is used.
template <bool VEC>
__device__ __forceinline__ void dw_load_row_nb(const float* plane, int row, int H, int W, int col, float (&r)[6]) {
    const bool ok = row >= 0 && row < H;
    const float* p = plane + (long)min(max(row, 0), H - 1) * W;
    const float4 v = irm_ld4<VEC>(p, col, W);
    const float l = p[max(col - 1, 0)], rr = p[min(col + 4, W - 1)];
    r[0] = (ok && col > 0) ? l : 0.0f;
    r[1] = ok ? v.x : 0.0f; r[2] = ok ? v.y : 0.0f; r[3] = ok ? v.z : 0.0f; r[4] = ok ? v.w : 0.0f;
    r[5] = (ok && col + 4 < W) ? rr : 0.0f;
}

__device__ __forceinline__ float4 dw_apply(const float (&k)[9], const float (&r0)[6], const float (&r1)[6],
                                           const float (&r2)[6], float bias) {
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = bias;
        s += k[0] * r0[i] + k[1] * r0[i + 1] + k[2] * r0[i + 2];
        s += k[3] * r1[i] + k[4] * r1[i + 1] + k[5] * r1[i + 2];
        s += k[6] * r2[i] + k[7] * r2[i + 1] + k[8] * r2[i + 2];
        o[i] = s;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

template <bool GATE, int RS, bool VEC>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs a) {
    IRM_KERNEL_ENTRY();
    const int cgs = (a.W + 3) >> 2;
    const int strips = (a.H + RS - 1) / RS;
    const long total = (long)a.C * strips * cgs;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % cgs);
    const long t = idx / cgs;
    const int strip = (int)(t % strips);
    const int c = (int)(t / strips);
    const int b = blockIdx.y;
    const int col = cg * 4;
    const int y0 = strip * RS;
    const long plane = (long)a.H * a.W;

    const float* xa = a.x + (long)b * a.x_bs + (long)c * plane;
    float ka[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) ka[i] = a.w[c * 9 + i];
    const float ba = a.bias ? a.bias[c] : 0.0f;

    const float* xb = nullptr;
    float kb[9];
    float bb = 0.0f;
    if (GATE) {
        xb = xa + (long)a.C * plane;
#pragma unroll
        for (int i = 0; i < 9; ++i) kb[i] = a.w[(c + a.C) * 9 + i];
        bb = a.bias ? a.bias[c + a.C] : 0.0f;
    }
    float* yo = a.y + (long)b * a.y_bs + (long)c * plane;

    if constexpr (!GATE) {
        // Plain depth-wise conv: every input row of the strip in flight at once (RS + 2 rows x 3 branch-free loads, 124
        // registers), then the arithmetic - the sliding window below keeps two rows in flight per wave and waits on memory
        // 74 % of its cycles.  In the model (same box, alternating builds): 1.46 -> 1.30 ms per frame.  The gated kernel would
        // need 182 registers (two tensors): measured 2.15 -> 2.25 ms; two half-strips with 6 rows of both tensors in flight
        // (140 registers): 2.08 -> 2.13 ms.  It keeps the window.
        float ra[RS + 2][6];
#pragma unroll
        for (int i = 0; i < RS + 2; ++i) dw_load_row_nb<VEC>(xa, y0 - 1 + i, a.H, a.W, col, ra[i]);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int y = y0 + r;
            float4 o = dw_apply(ka, ra[r], ra[r + 1], ra[r + 2], ba);
            if (a.act != IRM_ACT_NONE) {
                o.x = irm_act(o.x, a.act); o.y = irm_act(o.y, a.act);
                o.z = irm_act(o.z, a.act); o.w = irm_act(o.w, a.act);
            }
            if (y < a.H) irm_st4<VEC>(yo + (long)y * a.W, col, a.W, o);
        }
        return;
    }
    float a0[6], a1[6], a2[6], b0[6], b1[6], b2[6];
    dw_load_row<VEC>(xa, y0 - 1, a.H, a.W, col, a0);
    dw_load_row<VEC>(xa, y0, a.H, a.W, col, a1);
    if (GATE) {
        dw_load_row<VEC>(xb, y0 - 1, a.H, a.W, col, b0);
        dw_load_row<VEC>(xb, y0, a.H, a.W, col, b1);
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int y = y0 + r;
        if (y >= a.H) break;
        dw_load_row<VEC>(xa, y + 1, a.H, a.W, col, a2);
        float4 o = dw_apply(ka, a0, a1, a2, ba);
        if (GATE) {
            dw_load_row<VEC>(xb, y + 1, a.H, a.W, col, b2);
            const float4 g = dw_apply(kb, b0, b1, b2, bb);
            o.x = irm_gelu(o.x) * g.x; o.y = irm_gelu(o.y) * g.y;
            o.z = irm_gelu(o.z) * g.z; o.w = irm_gelu(o.w) * g.w;
#pragma unroll
            for (int i = 0; i < 6; ++i) { b0[i] = b1[i]; b1[i] = b2[i]; }
        } else if (a.act != IRM_ACT_NONE) {
            o.x = irm_act(o.x, a.act); o.y = irm_act(o.y, a.act);
            o.z = irm_act(o.z, a.act); o.w = irm_act(o.w, a.act);
        }
        irm_st4<VEC>(yo + (long)y * a.W, col, a.W, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) { a0[i] = a1[i]; a1[i] = a2[i]; }
    }
}

static int dw_launch(bool gate, const DwArgs& a, int B, hipStream_t stream) {
    constexpr int RS = 8;
    const long total = (long)a.C * ((a.H + RS - 1) / RS) * ((a.W + 3) >> 2);
    const long blocks = (total + 255) / 256;
    if (blocks > 2147483647L || B > 65535) return IRM_EINVAL;
    dim3 grid((unsigned)blocks, B);
    const bool vec = !(a.W & 3) && !(a.x_bs & 3) && !(a.y_bs & 3) && irm_aligned16(a.x) && irm_aligned16(a.y);
    if (gate && vec) hipLaunchKernelGGL((dwconv3x3_kernel<true, RS, true>), grid, dim3(256), 0, stream, a);
    else if (gate) hipLaunchKernelGGL((dwconv3x3_kernel<true, RS, false>), grid, dim3(256), 0, stream, a);
    else if (vec) hipLaunchKernelGGL((dwconv3x3_kernel<false, RS, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((dwconv3x3_kernel<false, RS, false>), grid, dim3(256), 0, stream, a);
    return irm_launch_status();
}

extern "C" int irm_dwconv3x3_f32(const float* x, long x_bs, const float* w, const float* bias, float* y,
                                 long y_bs, int B, int C, int H, int W, int act, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (act < 0 || act > 3) return IRM_EINVAL;
    DwArgs a{x, x_bs, w, bias, y, y_bs, C, H, W, act};
    return dw_launch(false, a, B, stream);
}

extern "C" int irm_dwconv3x3_gate_f32(const float* x, long x_bs, const float* w, const float* bias, float* y,
                                      long y_bs, int B, int hid, int H, int W, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || hid <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    DwArgs a{x, x_bs, w, bias, y, y_bs, hid, H, W, IRM_ACT_NONE};
    return dw_launch(true, a, B, stream);
}
