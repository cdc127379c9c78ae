// DeblurGANv2 FPN-MobileNet support kernels (src/deblurganv2/models/mobilenet_v2.py:5-57,
// models/fpn_mobilenet.py:53-70, 121-146).  The generator runs in train mode in the reference
// (src/deblurganv2/__init__.py:38) on one tile at a time, so every BatchNorm2d / InstanceNorm2d is a
// per-(sample, channel) normalisation over H x W with the biased variance: irm_chan_stats_f32 +
// irm_chan_norm_act_f32.  The remaining pieces are the stride-2 stem conv, the stride-2 depth-wise conv
// and nearest-neighbour up-sampling (+ lateral add).  All HBM-bound, planar NCHW.
#include "irm_common.h"

#define IRM_ACT_RELU6 4

// ---------------------------------------------------------------------------
// per (b, c) plane: mean and 1/sqrt(biased var + eps); two passes (mean, then centred squares; the
// second pass re-reads the plane from L2), fixed-order block reduction.
__device__ __forceinline__ float block_sum256(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ st,
                                                         int C, int N, float eps) {
    IRM_KERNEL_ENTRY();
    __shared__ float sh[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* p = x + (long)b * x_bs + (long)c * N;
    float s = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) s += p[i];
    const float mean = block_sum256(s, sh) / (float)N;
    float q = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) { const float d = p[i] - mean; q += d * d; }
    const float var = block_sum256(q, sh) / (float)N;
    if (threadIdx.x == 0) {
        st[((long)b * C + c) * 2] = mean;
        st[((long)b * C + c) * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
}

extern "C" int irm_chan_stats_f32(const float* x, long x_bs, float* stats, int B, int C, int N, float eps,
                                  hipStream_t stream) {
    if (!x || !stats || B <= 0 || C <= 0 || N <= 0 || B > 65535) return IRM_EINVAL;
    hipLaunchKernelGGL(chan_stats_kernel, dim3(C, B), dim3(256), 0, stream, x, x_bs, stats, C, N, eps);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
// Large planes (a 1280x720 frame is ONE tile for this model): split every plane over several workgroups.
// Each thread folds 16-element register chunks (exact two-pass mean / M2 per chunk) into a running
// (n, mean, M2) triple with Chan's merge; lanes, waves and finally the slices are merged in a fixed order,
// so the result does not depend on scheduling.  ws: [B*C][nsplit][3] floats.
struct WF { float n, mean, m2; };
__device__ __forceinline__ WF wf_merge(WF a, WF b) {
    const float n = a.n + b.n;
    if (n == 0.0f) return a;
    const float d = b.mean - a.mean, f = b.n / n;
    WF r;
    r.n = n;
    r.mean = fmaf(d, f, a.mean);
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    return r;
}

__global__ __launch_bounds__(256) void chan_partial_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ ws,
                                                           int C, int N, int nsplit) {
    IRM_KERNEL_ENTRY();
    __shared__ WF sh[4];
    const int sp = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const float4* p = reinterpret_cast<const float4*>(x + (long)b * x_bs + (long)c * N);
    const int nv = N >> 2;                                   // float4 units (N % 4 == 0 on this path)
    const int per = ((nv + nsplit - 1) / nsplit + 3) & ~3;   // units per slice, multiple of 4
    const int beg = sp * per, end = min(beg + per, nv);
    WF acc{0.f, 0.f, 0.f};
    for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
        float v[16];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j < end) {
                const float4 t = p[i + j];
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
                cnt += 4;
            } else {
                v[4 * j] = v[4 * j + 1] = v[4 * j + 2] = v[4 * j + 3] = 0.0f;
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
        WF ch;
        ch.n = (float)cnt;
        ch.mean = s / ch.n;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float d = v[j] - ch.mean; q += j < cnt ? d * d : 0.0f; }
        ch.m2 = q;
        acc = wf_merge(acc, ch);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        WF other;
        other.n = __shfl_xor(acc.n, o); other.mean = __shfl_xor(acc.mean, o); other.m2 = __shfl_xor(acc.m2, o);
        // same operand order on both partners: the lower lane is always the left operand
        acc = (threadIdx.x & o) ? wf_merge(other, acc) : wf_merge(acc, other);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const WF r = wf_merge(wf_merge(sh[0], sh[1]), wf_merge(sh[2], sh[3]));
        float* o = ws + (((long)b * C + c) * nsplit + sp) * 3;
        o[0] = r.n; o[1] = r.mean; o[2] = r.m2;
    }
}

__global__ __launch_bounds__(64) void chan_finish_kernel(const float* __restrict__ ws, float* __restrict__ st, int planes,
                                                         int nsplit, float eps) {
    IRM_KERNEL_ENTRY();
    const int pl = blockIdx.x * 64 + threadIdx.x;
    if (pl >= planes) return;
    const float* w = ws + (long)pl * nsplit * 3;
    WF acc{w[0], w[1], w[2]};
    for (int i = 1; i < nsplit; ++i) acc = wf_merge(acc, WF{w[3 * i], w[3 * i + 1], w[3 * i + 2]});
    st[(long)pl * 2] = acc.mean;
    st[(long)pl * 2 + 1] = 1.0f / sqrtf(acc.m2 / acc.n + eps);
}

extern "C" int irm_chan_stats_ws_f32(const float* x, long x_bs, float* stats, float* ws, long ws_floats, int B, int C,
                                     int N, float eps, hipStream_t stream) {
    if (!x || !stats || B <= 0 || C <= 0 || N <= 0 || B > 65535 || C > 65535) return IRM_EINVAL;
    // slices per plane: about 1024 workgroups in total, at least 4096 elements each
    int nsplit = (1024 + B * C - 1) / (B * C);
    nsplit = min(nsplit, max(N / 4096, 1));
    const bool vec = !(N & 3) && !(x_bs & 3) && irm_aligned16(x);
    if (nsplit <= 1 || !vec || !ws || ws_floats < (long)B * C * nsplit * 3)
        return irm_chan_stats_f32(x, x_bs, stats, B, C, N, eps, stream);
    hipLaunchKernelGGL(chan_partial_kernel, dim3(nsplit, C, B), dim3(256), 0, stream, x, x_bs, ws, C, N, nsplit);
    int rc = irm_launch_status();
    if (rc != IRM_OK) return rc;
    hipLaunchKernelGGL(chan_finish_kernel, dim3((B * C + 63) / 64), dim3(64), 0, stream, ws, stats, B * C, nsplit, eps);
    return irm_launch_status();
}

// y = act((x - mean) * rstd * w[c] + b[c]) (+ res); w, b optional (InstanceNorm2d(affine=False)); in place ok
__global__ __launch_bounds__(256) void chan_norm_act_kernel(const float* __restrict__ x, long x_bs,
                                                            const float* __restrict__ st, const float* __restrict__ w,
                                                            const float* __restrict__ bi, const float* __restrict__ res,
                                                            long r_bs, float* __restrict__ y, long y_bs, int C, int N,
                                                            int act) {
    IRM_KERNEL_ENTRY();
    const int c = blockIdx.y, b = blockIdx.z;
    const float mean = st[((long)b * C + c) * 2], rstd = st[((long)b * C + c) * 2 + 1];
    const float g = w ? w[c] : 1.0f, be = bi ? bi[c] : 0.0f;
    const long off = (long)c * N;
    const float* xp = x + (long)b * x_bs + off;
    const float* rp = res ? res + (long)b * r_bs + off : nullptr;
    float* yp = y + (long)b * y_bs + off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        float v = (xp[i] - mean) * rstd * g + be;
        if (act == IRM_ACT_RELU) v = fmaxf(v, 0.0f);
        else if (act == IRM_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
        if (rp) v += rp[i];
        yp[i] = v;
    }
}

extern "C" int irm_chan_norm_act_f32(const float* x, long x_bs, const float* stats, const float* w, const float* b,
                                     const float* res, long r_bs, float* y, long y_bs, int B, int C, int N, int act,
                                     hipStream_t stream) {
    if (!x || !stats || !y || B <= 0 || C <= 0 || N <= 0 || B > 65535 || C > 65535) return IRM_EINVAL;
    if (act != IRM_ACT_NONE && act != IRM_ACT_RELU && act != IRM_ACT_RELU6) return IRM_EINVAL;
    int gx = (N + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(chan_norm_act_kernel, dim3(gx, C, B), dim3(256), 0, stream, x, x_bs, stats, w, b, res, r_bs, y,
                       y_bs, C, N, act);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
// dense 3x3, stride 2, zero pad 1, no bias, small Ci (the MobileNetV2 stem conv_bn(3, 32, 2),
// mobilenet_v2.py:5-10).  One thread = one output pixel x 8 output channels.
__global__ __launch_bounds__(256) void conv3x3_s2_kernel(const float* __restrict__ x, long x_bs,
                                                         const float* __restrict__ w, float* __restrict__ y, long y_bs,
                                                         int Ci, int Co, int H, int W, int Ho, int Wo) {
    IRM_KERNEL_ENTRY();
    const int b = blockIdx.z, cg = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= Ho * Wo) return;
    const int oy = o / Wo, ox = o % Wo;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const float* xp = x + (long)b * x_bs;
    for (int ci = 0; ci < Ci; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const float v = xp[((long)ci * H + iy) * W + ix];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int co = cg * 8 + j;
                    if (co < Co) acc[j] = fmaf(v, w[(((long)co * Ci + ci) * 3 + ky) * 3 + kx], acc[j]);
                }
            }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = cg * 8 + j;
        if (co < Co) y[(long)b * y_bs + ((long)co * Ho + oy) * Wo + ox] = acc[j];
    }
}

extern "C" int irm_conv3x3_s2_f32(const float* x, long x_bs, const float* w, float* y, long y_bs, int B, int Ci, int Co,
                                  int H, int W, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || B > 65535) return IRM_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;          // floor((H + 2 - 3) / 2) + 1
    dim3 grid((Ho * Wo + 255) / 256, (Co + 7) / 8, B);
    hipLaunchKernelGGL(conv3x3_s2_kernel, grid, dim3(256), 0, stream, x, x_bs, w, y, y_bs, Ci, Co, H, W, Ho, Wo);
    return irm_launch_status();
}

// depth-wise 3x3, stride 2, zero pad 1, no bias (InvertedResidual dw, mobilenet_v2.py:32-46)
__global__ __launch_bounds__(256) void dwconv3x3_s2_kernel(const float* __restrict__ x, long x_bs,
                                                           const float* __restrict__ w, float* __restrict__ y, long y_bs,
                                                           int H, int W, int Ho, int Wo) {
    IRM_KERNEL_ENTRY();
    const int c = blockIdx.y, b = blockIdx.z;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= Ho * Wo) return;
    const int oy = o / Wo, ox = o % Wo;
    const float* xp = x + (long)b * x_bs + (long)c * H * W;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + i];
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            acc = fmaf(xp[(long)iy * W + ix], k[ky * 3 + kx], acc);
        }
    }
    y[(long)b * y_bs + ((long)c * Ho + oy) * Wo + ox] = acc;
}

extern "C" int irm_dwconv3x3_s2_f32(const float* x, long x_bs, const float* w, float* y, long y_bs, int B, int C, int H,
                                    int W, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || C > 65535) return IRM_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    dim3 grid((Ho * Wo + 255) / 256, C, B);
    hipLaunchKernelGGL(dwconv3x3_s2_kernel, grid, dim3(256), 0, stream, x, x_bs, w, y, y_bs, H, W, Ho, Wo);
    return irm_launch_status();
}

// out[b][c][y][x] = (add ? add[b][c][y][x] : 0) + src[b][c][y / s][x / s]     (nearest, integer scale s)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ src, long s_bs,
                                                           const float* __restrict__ add, long a_bs,
                                                           float* __restrict__ out, long o_bs, int Hs, int Ws, int s) {
    IRM_KERNEL_ENTRY();
    const int c = blockIdx.y, b = blockIdx.z;
    const int Ho = Hs * s, Wo = Ws * s;
    const float* sp = src + (long)b * s_bs + (long)c * Hs * Ws;
    const float* ap = add ? add + (long)b * a_bs + (long)c * Ho * Wo : nullptr;
    float* op = out + (long)b * o_bs + (long)c * Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        const int y = i / Wo, x = i % Wo;
        const float v = sp[(long)(y / s) * Ws + x / s];
        op[i] = ap ? ap[i] + v : v;
    }
}

extern "C" int irm_upsample_add_f32(const float* src, long s_bs, const float* add, long a_bs, float* out, long o_bs,
                                    int B, int C, int Hs, int Ws, int scale, hipStream_t stream) {
    if (!src || !out || B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || scale <= 0 || B > 65535 || C > 65535) return IRM_EINVAL;
    int gx = (Hs * Ws * scale * scale + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(gx, C, B), dim3(256), 0, stream, src, s_bs, add, a_bs, out, o_bs, Hs, Ws,
                       scale);
    return irm_launch_status();
}
