// Dense 3x3 convs with very few channels on one side, exact fp32 on the vector pipe: they are memory streams (a 96 -> 3
// conv reads 96 planes and writes 3), an implicit GEMM pads the thin side to a 16-wide matrix tile and runs at a
// fraction of the stream rate (the emulated kernel: 414 us for Restormer's output conv on 6 x 512^2, 1.5 TB/s).
//   thin output (Co <= 4): Restormer `output` (+ inp_img, restormer.py:281), DnCNN's last conv + x - n
//     (network_dncnn.py:66-71), REDNet's last deconv + x (rednet.py:133-136), DeblurGANv2 `final` (fpn_mobilenet.py:68-70)
//   thin input (Ci <= 4): OverlapPatchEmbed (restormer.py:156-164), DnCNN / REDNet first conv
// Work item = 4 consecutive columns x RS rows (16-byte row accesses, 3-row register window); the weights of a wave are
// wave-uniform (scalar loads).  Zero pad 1, stride 1, W % 4 == 0, 16-byte aligned rows; epilogue as irm_conv3x3_f32
// (bias, relu1, res_mode 1 / 2 / 3, relu2), store_mode 0 only.
#include "irm_common.h"

struct ThinArgs {
    const float* w;                // [Co][Ci][3][3]
    const float* x; long x_bs;     // [B][Ci][H][W]
    float* y; long y_bs;           // [B][Co][H][W]
    const float* res; long r_bs;   // [B][Co][H][W] or null
    const float* bias;             // [Co] or null
    int Ci, Co, H, W;
    int relu1, res_mode, relu2;
    int cgs, strips;               // column groups of 4, row strips of RS
};

__device__ __forceinline__ void thin_row(const float* plane, int row, int H, int W, int col, float (&r)[6]) {
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

__device__ __forceinline__ float thin_epilogue(float v, float bias, float r, const ThinArgs& a) {
    v += bias;
    if (a.relu1) v = fmaxf(v, 0.0f);
    if (a.res_mode == 1) v += r;
    else if (a.res_mode == 2) v = r - v;
    else if (a.res_mode == 3) v = fminf(fmaxf(tanhf(v) + r, -1.0f), 1.0f);
    if (a.relu2) v = fmaxf(v, 0.0f);
    return v;
}

// ---- thin output: wave g of the workgroup sums input channels g, g + 4, ... into CO x RS x 4 accumulators; the four
// partial sums meet in LDS in wave order (bitwise reproducible), wave 0 applies the epilogue and stores.
template <int CO, int RS>
__global__ __launch_bounds__(256) void conv3x3_thin_out_kernel(ThinArgs a) {
    IRM_KERNEL_ENTRY();
    __shared__ float part[3][CO * RS * 4][64];
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 64 + lane;
    const long total = (long)a.strips * a.cgs;
    const bool live = item < total;
    const int cg = live ? (int)(item % a.cgs) : 0, strip = live ? (int)(item / a.cgs) : 0;
    const int b = blockIdx.y;
    const int col = cg * 4, y0 = strip * RS;
    const long plane = (long)a.H * a.W;
    const float* xb = a.x + (long)b * a.x_bs;
    float acc[CO][RS][4];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][r][i] = 0.0f;
    for (int ci = g; ci < a.Ci; ci += 4) {
        float k[CO][9];
#pragma unroll
        for (int c = 0; c < CO; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) k[c][t] = (c < a.Co) ? a.w[((long)c * a.Ci + ci) * 9 + t] : 0.0f;   // wave-uniform
        const float* xp = xb + (long)ci * plane;
        float r0[6], r1[6], r2[6];
        thin_row(xp, y0 - 1, a.H, a.W, col, r0);
        thin_row(xp, y0, a.H, a.W, col, r1);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            thin_row(xp, y0 + r + 1, a.H, a.W, col, r2);
#pragma unroll
            for (int c = 0; c < CO; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s = acc[c][r][i];
                    s = fmaf(k[c][0], r0[i], s); s = fmaf(k[c][1], r0[i + 1], s); s = fmaf(k[c][2], r0[i + 2], s);
                    s = fmaf(k[c][3], r1[i], s); s = fmaf(k[c][4], r1[i + 1], s); s = fmaf(k[c][5], r1[i + 2], s);
                    s = fmaf(k[c][6], r2[i], s); s = fmaf(k[c][7], r2[i + 1], s); s = fmaf(k[c][8], r2[i + 2], s);
                    acc[c][r][i] = s;
                }
#pragma unroll
            for (int i = 0; i < 6; ++i) { r0[i] = r1[i]; r1[i] = r2[i]; }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int c = 0; c < CO; ++c)
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) part[g - 1][(c * RS + r) * 4 + i][lane] = acc[c][r][i];
    }
    __syncthreads();
    if (g > 0 || !live) return;
    float* yb = a.y + (long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long)b * a.r_bs : nullptr;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        if (c >= a.Co) break;
        const float bv = a.bias ? a.bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int yy = y0 + r;
            if (yy >= a.H) break;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = (c * RS + r) * 4 + i;
                v[i] = ((acc[c][r][i] + part[0][e][lane]) + part[1][e][lane]) + part[2][e][lane];
            }
            const long off = (long)c * plane + (long)yy * a.W + col;
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rb) rv = *reinterpret_cast<const float4*>(rb + off);
            *reinterpret_cast<float4*>(yb + off) = make_float4(thin_epilogue(v[0], bv, rv.x, a), thin_epilogue(v[1], bv, rv.y, a),
                                                               thin_epilogue(v[2], bv, rv.z, a), thin_epilogue(v[3], bv, rv.w, a));
        }
    }
}

// ---- thin input: the CI x (RS + 2) x 6 input window of a work item stays in registers, the output channels of the
// workgroup's range are produced one after the other (their 9 CI weights are wave-uniform) and stored at once.
template <int CI, int RS>
__global__ __launch_bounds__(256) void conv3x3_thin_in_kernel(ThinArgs a, int co_per_group) {
    IRM_KERNEL_ENTRY();
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.strips * a.cgs;
    if (item >= total) return;
    const int cg = (int)(item % a.cgs), strip = (int)(item / a.cgs);
    const int b = blockIdx.z;
    const int col = cg * 4, y0 = strip * RS;
    const long plane = (long)a.H * a.W;
    const float* xb = a.x + (long)b * a.x_bs;
    float win[CI][RS + 2][6];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int r = 0; r < RS + 2; ++r) {
            if (ci < a.Ci) thin_row(xb + (long)ci * plane, y0 - 1 + r, a.H, a.W, col, win[ci][r]);
            else {
#pragma unroll
                for (int i = 0; i < 6; ++i) win[ci][r][i] = 0.0f;
            }
        }
    float* yb = a.y + (long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long)b * a.r_bs : nullptr;
    const int c0 = blockIdx.y * co_per_group, c1 = min(c0 + co_per_group, a.Co);
    for (int co = c0; co < c1; ++co) {
        float k[CI][9];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) k[ci][t] = (ci < a.Ci) ? a.w[((long)co * a.Ci + ci) * 9 + t] : 0.0f;   // uniform
        const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int yy = y0 + r;
            if (yy >= a.H) break;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.0f;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        s = fmaf(k[ci][dy * 3], win[ci][r + dy][i], s);
                        s = fmaf(k[ci][dy * 3 + 1], win[ci][r + dy][i + 1], s);
                        s = fmaf(k[ci][dy * 3 + 2], win[ci][r + dy][i + 2], s);
                    }
                v[i] = s;
            }
            const long off = (long)co * plane + (long)yy * a.W + col;
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rb) rv = *reinterpret_cast<const float4*>(rb + off);
            *reinterpret_cast<float4*>(yb + off) = make_float4(thin_epilogue(v[0], bv, rv.x, a), thin_epilogue(v[1], bv, rv.y, a),
                                                               thin_epilogue(v[2], bv, rv.z, a), thin_epilogue(v[3], bv, rv.w, a));
        }
    }
}

// w: the conv weight itself, [Co][Ci][3][3] fp32 (device).  Co <= 4 or Ci <= 4; W % 4 == 0; 16-byte aligned tensors.
extern "C" int irm_conv3x3_thin_f32(const float* w, const float* x, long x_bs, float* y, long y_bs, const float* res,
                                    long r_bs, const float* bias, int B, int Ci, int Co, int H, int W, int relu1,
                                    int res_mode, int relu2, hipStream_t stream) {
    if (!w || !x || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || B > 65535) return IRM_EINVAL;
    if (res_mode < 0 || res_mode > 3 || (res_mode && !res)) return IRM_EINVAL;
    if ((W & 3) || (x_bs & 3) || (y_bs & 3) || (r_bs & 3)) return IRM_EINVAL;
    if (!irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(res)) return IRM_EINVAL;
    if (Co > 4 && Ci > 4) return IRM_EINVAL;
    ThinArgs a{w, x, x_bs, y, y_bs, res_mode ? res : nullptr, r_bs, bias, Ci, Co, H, W, relu1, res_mode, relu2, W / 4, 0};
    if (Co <= 4 && Ci > 4) {
        constexpr int RS = 4;
        a.strips = (H + RS - 1) / RS;
        const long total = (long)a.strips * a.cgs;
        const dim3 grid((unsigned)((total + 63) / 64), B);
        if (Co == 1) hipLaunchKernelGGL((conv3x3_thin_out_kernel<1, RS>), grid, dim3(256), 0, stream, a);
        else if (Co == 2) hipLaunchKernelGGL((conv3x3_thin_out_kernel<2, RS>), grid, dim3(256), 0, stream, a);
        else if (Co == 3) hipLaunchKernelGGL((conv3x3_thin_out_kernel<3, RS>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv3x3_thin_out_kernel<4, RS>), grid, dim3(256), 0, stream, a);
        return irm_launch_status();
    }
    constexpr int RS = 2;
    a.strips = (H + RS - 1) / RS;
    const long total = (long)a.strips * a.cgs;
    const unsigned gx = (unsigned)((total + 255) / 256);
    // output-channel groups: enough workgroups to fill the chip on small images, at least 8 channels per group
    int groups = 1;
    while ((long)gx * B * groups < 1024 && Co / (groups * 2) >= 8) groups *= 2;
    const int per = (Co + groups - 1) / groups;
    const dim3 grid(gx, (Co + per - 1) / per, B);
    if (Ci == 1) hipLaunchKernelGGL((conv3x3_thin_in_kernel<1, RS>), grid, dim3(256), 0, stream, a, per);
    else if (Ci == 2) hipLaunchKernelGGL((conv3x3_thin_in_kernel<2, RS>), grid, dim3(256), 0, stream, a, per);
    else if (Ci == 3) hipLaunchKernelGGL((conv3x3_thin_in_kernel<3, RS>), grid, dim3(256), 0, stream, a, per);
    else hipLaunchKernelGGL((conv3x3_thin_in_kernel<4, RS>), grid, dim3(256), 0, stream, a, per);
    return irm_launch_status();
}
