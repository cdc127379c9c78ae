// Dense 3x3 convolution (stride 1, zero pad 1) as an implicit GEMM on the
// exact-f32 MFMA, planar NCHW.
//
// Replaces: Restormer's OverlapPatchEmbed / Downsample / Upsample / output convs
// (src/restormer/restormer.py:156-189, 243) with PixelUnshuffle / PixelShuffle
// and the global residual folded into the store; the conv3x3+bias+ReLU stacks
// of DnCNN (src/dncnn/models/network_dncnn.py:40-71) and REDNet
// (src/rednet/rednet.py:64-136; ConvTranspose2d(k3,s1,p1) is a conv with the
// weight transposed and flipped, done when the weights are packed).
//
// Mapping: one 256-thread workgroup = an 8 x 32 pixel tile; the input halo tile
// (10 x 34) of 8 input channels at a time is staged through double-buffered LDS
// (plane stride 400 floats = 16 mod 32, so the two k-rows a half-wave reads hit
// disjoint bank halves).  Each wave owns 4 pixel groups of 16 consecutive
// columns (2 rows x 2 halves) and CT output-channel tiles; for every tap (dy,dx)
// the A operand is the LDS tile shifted by (dy,dx), the B operand the packed
// weight of that tap:  Wp[tap][mtile][kstep][lane] =
// W[16 mtile + (lane&15)][4 kstep + (lane>>4)][tap/3][tap%3].
#include "irm_common.h"
#include <stdlib.h>

#define CV_TH 8
#define CV_TW 32
#define CV_CK 8
#define CV_ROWS (CV_TH + 2)
#define CV_TWP 40
#define CV_PLANE (CV_ROWS * CV_TWP)   // 400

__device__ __forceinline__ float cv_res(float v, float r, int mode) {
    if (mode == 1) return v + r;
    if (mode == 2) return r - v;
    return fminf(fmaxf(tanhf(v) + r, -1.0f), 1.0f);      // DeblurGANv2 output: fpn_mobilenet.py:68-70
}

struct ConvArgs {
    const float* Wp;              // packed [9][mtiles][ksteps][64]
    const float* X; long x_bs;    // [B][Ci][H][W]
    float* Y; long y_bs;
    const float* R; long r_bs;    // residual (normal store mode only) or null
    const float* bias;            // [Co] or null
    int Ci, Co, H, W;
    int mtiles, ksteps;           // ceil(Co/16), 2*ceil(Ci/8)
    int relu1;                    // relu right after bias
    int res_mode;                 // 0 none, 1: v += R, 2: v = R - v, 3: v = clamp(tanh(v) + R, -1, 1)
    int relu2;                    // relu after the residual
    int store_mode;               // 0 NCHW, 1 PixelUnshuffle(2), 2 PixelShuffle(2)
    int tiles_x;
    int vec;                      // 16-byte / 8-byte store fast paths are legal
};

template <int CT>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
    IRM_KERNEL_ENTRY();
    __shared__ float xs[2][CV_CK * CV_PLANE];
    // packed weights of one stage (9 taps x CT tiles x 2 k-steps x 64 lanes), double buffered like xs: read
    // from global once per stage with coalesced loads instead of one dependent load per tap inside the MFMA loop
    constexpr int WST = 9 * CT * 2 * 64;
    __shared__ float wsm[2][WST];
    constexpr int WLOAD = (WST + 255) / 256;
    float wr[WLOAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int ty0 = (blockIdx.x / a.tiles_x) * CV_TH;
    const int tx0 = (blockIdx.x % a.tiles_x) * CV_TW;
    const float* X = a.X + (long)b * a.x_bs;
    const long plane = (long)a.H * a.W;

    constexpr int NLOAD = (CV_CK * CV_ROWS * (CV_TW + 2) + 255) / 256;   // 11
    float xr[NLOAD];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int e = tid + j * 256;
            float v = 0.0f;
            if (e < CV_CK * CV_ROWS * (CV_TW + 2)) {
                const int ch = e / (CV_ROWS * (CV_TW + 2));
                const int rem = e % (CV_ROWS * (CV_TW + 2));
                const int yy = rem / (CV_TW + 2), xx = rem % (CV_TW + 2);
                const int ci = s * CV_CK + ch, gy = ty0 + yy - 1, gx = tx0 + xx - 1;
                if (ci < a.Ci && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = X[(long)ci * plane + (long)gy * a.W + gx];
            }
            xr[j] = v;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int e = tid + j * 256;
            if (e < CV_CK * CV_ROWS * (CV_TW + 2)) {
                const int ch = e / (CV_ROWS * (CV_TW + 2));
                const int rem = e % (CV_ROWS * (CV_TW + 2));
                const int yy = rem / (CV_TW + 2), xx = rem % (CV_TW + 2);
                xs[buf][ch * CV_PLANE + yy * CV_TWP + xx] = xr[j];
            }
        }
    };

    const int g = lane >> 4, r = lane & 15;
    const int nstages = a.ksteps / 2;
    const int nchunks = (a.mtiles + CT - 1) / CT;

    for (int chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        const int mt0 = chunk * CT;
        const int nct = min(CT, a.mtiles - mt0);
        auto load_w = [&](int s) {
#pragma unroll
            for (int j = 0; j < WLOAD; ++j) {
                const int e = tid + j * 256;                   // ((tap * CT + c) * 2 + kk) * 64 + lane
                float v = 0.0f;
                if (e < WST) {
                    const int ln = e & 63, kk = (e >> 6) & 1, tc = e >> 7, c = tc % CT, tap = tc / CT;
                    if (c < nct) v = a.Wp[(((long)tap * a.mtiles + mt0 + c) * a.ksteps + s * 2 + kk) * 64 + ln];
                }
                wr[j] = v;
            }
        };
        auto store_w = [&](int buf) {
#pragma unroll
            for (int j = 0; j < WLOAD; ++j) {
                const int e = tid + j * 256;
                if (e < WST) wsm[buf][e] = wr[j];
            }
        };
        f32x4 acc[4][CT];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        __syncthreads();                      // the previous chunk is done with both buffers
        load_stage(0);
        load_w(0);
        store_stage(0);
        store_w(0);
        __syncthreads();
        for (int s = 0; s < nstages; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstages) { load_stage(s + 1); load_w(s + 1); }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    float bf[CT], af[4];
#pragma unroll
                    for (int c = 0; c < CT; ++c) bf[c] = wsm[buf][((tap * CT + c) * 2 + kk) * 64 + lane];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int yy = wave * 2 + (p >> 1) + dy;
                        const int xx = (p & 1) * 16 + r + dx;
                        af[p] = xs[buf][(kk * 4 + g) * CV_PLANE + yy * CV_TWP + xx];
                    }
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        if (c < nct) {
#pragma unroll
                            for (int p = 0; p < 4; ++p) acc[p][c] = irm_mfma16(af[p], bf[c], acc[p][c]);
                        }
                    }
                }
            }
            if (s + 1 < nstages) { store_stage(buf ^ 1); store_w(buf ^ 1); }
            __syncthreads();
        }

        // epilogue: lane holds pixels (y, x..x+3) of channel co
        float* Y = a.Y + (long)b * a.y_bs;
        const float* R = a.R ? a.R + (long)b * a.r_bs : nullptr;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int co = (mt0 + c) * 16 + r;
            if (c < nct && co < a.Co) {
                const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int y = ty0 + wave * 2 + (p >> 1);
                    const int x = tx0 + (p & 1) * 16 + g * 4;
                    if (y >= a.H || x >= a.W) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[p][c][e] + bv;
                        if (a.relu1) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if (a.store_mode == 0) {
                        const long off = (long)co * plane + (long)y * a.W + x;
                        if (a.vec) {
                            if (a.res_mode) {
                                const float4 rr = *reinterpret_cast<const float4*>(R + off);
                                const float rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = cv_res(v[e], rv[e], a.res_mode);
                            }
                            if (a.relu2) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                            }
                            *reinterpret_cast<float4*>(Y + off) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (x + e < a.W) {
                                    float t = v[e];
                                    if (a.res_mode) t = cv_res(t, R[off + e], a.res_mode);
                                    if (a.relu2) t = fmaxf(t, 0.0f);
                                    Y[off + e] = t;
                                }
                            }
                        }
                    } else if (a.store_mode == 1) {
                        // PixelUnshuffle(2): out[co*4 + (y&1)*2 + (x&1)][y/2][x/2]; H, W even
                        const int oh = a.H >> 1, ow = a.W >> 1;
                        const long op = (long)oh * ow;
                        const int oc = co * 4 + (y & 1) * 2;
                        const long o = (long)(y >> 1) * ow + (x >> 1);
                        if (a.vec) {
                            *reinterpret_cast<float2*>(Y + (long)oc * op + o) = make_float2(v[0], v[2]);
                            *reinterpret_cast<float2*>(Y + (long)(oc + 1) * op + o) = make_float2(v[1], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (x + e < a.W) Y[(long)(oc + (e & 1)) * op + o + (e >> 1)] = v[e];
                        }
                    } else {
                        // PixelShuffle(2): out[co/4][2y + ((co>>1)&1)][2x + (co&1)]
                        const int ow = a.W * 2;
                        const long op = (long)a.H * 2 * ow;
                        const int oc = co >> 2, i = (co >> 1) & 1, jx = co & 1;
                        float* o = Y + (long)oc * op + (long)(2 * y + i) * ow + 2 * x + jx;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (x + e < a.W) o[2 * e] = v[e];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Fast path (W % 4 == 0, 16-byte aligned planes): the same implicit GEMM as an LDS-DMA ring, 4 input
// channels (one MFMA k-step) per stage.  The halo tile is fetched as aligned 16-byte chunks covering
// columns [tx0-4, tx0+36) (10 chunks x 10 rows x 4 channels, one contiguous LDS image with the same
// 400-float plane stride), out-of-image chunks read a 16-byte zero page, and the 9 taps' packed weights
// of the pass (9*CT*256 B) arrive by DMA as well.  NS-1 stages stay in flight behind a counted vmcnt;
// the pipeline runs across output-channel passes.
__device__ __attribute__((aligned(16))) float irm_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int N>
__device__ __forceinline__ void cv_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CT, int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_ring_kernel(ConvArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int XU = 8;                          // 1 KiB units of the input image per stage (400 chunks + pad:
                                                   // every wave issues the same number of DMA instructions)
    constexpr int WCH = 9 * CT * 16;               // 16-byte chunks of weights per stage
    constexpr int WU = (WCH + 63) / 64;            // 1 KiB units of weights per stage
    constexpr int WL = (WU + 3) / 4;               // weight DMA instructions per wave per stage
    constexpr int STG = (XU + WU) * 256;           // floats per stage
    constexpr int LPS = 2 + WL;
    static_assert((NS - 2) * LPS <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.z;
    const int ty0 = (blockIdx.x / a.tiles_x) * CV_TH;
    const int tx0 = (blockIdx.x % a.tiles_x) * CV_TW;
    const float* X = a.X + (long)b * a.x_bs;
    const long plane = (long)a.H * a.W;

    // per-lane source geometry of the two input DMA instructions of this wave (fixed over stages)
    long xoff[2];
    int xch[2];
    bool xok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (wave * 2 + j) * 64 + lane;             // chunk id within the stage image
        const int ch = q / 100, rem = q % 100, row = rem / 10, chunk = rem % 10;
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + chunk * 4;
        xch[j] = ch;
        xok[j] = q < 400 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        xoff[j] = (long)ch * plane + (long)gy * a.W + gx;
    }

    const int S = (a.Ci + 3) / 4;                             // stages = k-steps with real channels
    const int nchunks = (a.mtiles + CT - 1) / CT;
    const int my_chunks = (nchunks - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int TOT = my_chunks * S;

    auto issue = [&](int it) {
        const int ci = it / S, s = it - ci * S;
        const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * CT;
        float* xb = smem + (it % NS) * STG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = xok[j] && s * 4 + xch[j] < a.Ci;
            const float* src = ok ? X + (long)s * 4 * plane + xoff[j] : irm_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xb + (wave * 2 + j) * 256),
                                                 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int u = min(wave + 4 * i, WU - 1);          // surplus instructions repeat the last unit
            const int qq = u * 64 + lane;
            const int pair = qq >> 4, tap = pair / CT, ct = pair - tap * CT;
            const int mt = min(mt0 + ct, a.mtiles - 1);
            const float* src = qq < WCH
                ? a.Wp + (((long)tap * a.mtiles + mt) * a.ksteps + s) * 64 + (qq & 15) * 4 : irm_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xb + (XU + u) * 256), 16, 0, 0);
        }
    };

    f32x4 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < TOT) issue(j);

    int s = 0, ci = 0;
    for (int it = 0; it < TOT; ++it) {
        const int rem = min(NS - 2, TOT - 1 - it);
        if (rem >= 1 && NS >= 3) cv_wait_vmcnt<LPS>();
        else cv_wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");
        if (it + NS - 1 < TOT) issue(it + NS - 1);

        const float* xb = smem + (it % NS) * STG;
        const float* wb = xb + XU * 256;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            float af[4], bf[CT];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                af[p] = xb[g * CV_PLANE + (wave * 2 + (p >> 1) + dy) * CV_TWP + (p & 1) * 16 + r + dx + 3];
#pragma unroll
            for (int c = 0; c < CT; ++c) bf[c] = wb[(tap * CT + c) * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p][c] = irm_mfma16(af[p], bf[c], acc[p][c]);
        }

        if (++s == S) {
            const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * CT;
            float* Y = a.Y + (long)b * a.y_bs;
            const float* R = a.R ? a.R + (long)b * a.r_bs : nullptr;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = (mt0 + c) * 16 + r;
                const bool row_ok = mt0 + c < a.mtiles && co < a.Co;
                const float bv = (a.bias && row_ok) ? a.bias[co] : 0.0f;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int y = ty0 + wave * 2 + (p >> 1);
                    const int x = tx0 + (p & 1) * 16 + g * 4;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[p][c][e] + bv;
                        if (a.relu1) v[e] = fmaxf(v[e], 0.0f);
                    }
                    acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!row_ok || y >= a.H || x >= a.W) continue;
                    if (a.store_mode == 0) {
                        const long off = (long)co * plane + (long)y * a.W + x;
                        if (a.res_mode) {
                            const float4 rr = *reinterpret_cast<const float4*>(R + off);
                            const float rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = cv_res(v[e], rv[e], a.res_mode);
                        }
                        if (a.relu2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                        }
                        *reinterpret_cast<float4*>(Y + off) = make_float4(v[0], v[1], v[2], v[3]);
                    } else if (a.store_mode == 1) {
                        const int oh = a.H >> 1, ow = a.W >> 1;
                        const long op = (long)oh * ow;
                        const int oc = co * 4 + (y & 1) * 2;
                        const long o = (long)(y >> 1) * ow + (x >> 1);
                        *reinterpret_cast<float2*>(Y + (long)oc * op + o) = make_float2(v[0], v[2]);
                        *reinterpret_cast<float2*>(Y + (long)(oc + 1) * op + o) = make_float2(v[1], v[3]);
                    } else {
                        const int ow = a.W * 2;
                        const long op = (long)a.H * 2 * ow;
                        const int oc = co >> 2, i = (co >> 1) & 1, jx = co & 1;
                        float* o = Y + (long)oc * op + (long)(2 * y + i) * ow + 2 * x + jx;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[2 * e] = v[e];
                    }
                }
            }
            s = 0;
            ++ci;
        }
    }
}

template <int CT>
static int launch_conv_ring(const ConvArgs& a, int B, int ygroups, hipStream_t stream) {
    constexpr int NS = 3;
    constexpr int WU = (9 * CT * 16 + 63) / 64;
    const size_t lds = (size_t)NS * (8 + WU) * 1024;
    IRM_ALLOW_BIG_LDS((&conv3x3_ring_kernel<CT, NS>));
    const int tiles_y = (a.H + CV_TH - 1) / CV_TH;
    dim3 grid(a.tiles_x * tiles_y, ygroups, B);
    hipLaunchKernelGGL((conv3x3_ring_kernel<CT, NS>), grid, dim3(256), lds, stream, a);
    return irm_launch_status();
}

template <int CT>
static int launch_conv(const ConvArgs& a, int B, int ygroups, hipStream_t stream) {
    const int tiles_y = (a.H + CV_TH - 1) / CV_TH;
    dim3 grid(a.tiles_x * tiles_y, ygroups, B);
    hipLaunchKernelGGL((conv3x3_kernel<CT>), grid, dim3(256), 0, stream, a);
    return irm_launch_status();
}

extern "C" int irm_conv3x3_f32(const float* wp, const float* x, long x_bs, float* y, long y_bs, const float* res,
                               long r_bs, const float* bias, int B, int Ci, int Co, int H, int W, int relu1,
                               int res_mode, int relu2, int store_mode, int ct, int ygroups,
                               hipStream_t stream) {
    if (!wp || !x || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (res_mode < 0 || res_mode > 3 || (res_mode && !res) || store_mode < 0 || store_mode > 2) return IRM_EINVAL;
    if (store_mode != 0 && res_mode != 0) return IRM_EINVAL;
    if (store_mode == 1 && ((H & 1) || (W & 1))) return IRM_EINVAL;
    if (store_mode == 2 && (Co & 3)) return IRM_EINVAL;
    if (B > 65535) return IRM_EINVAL;
    ConvArgs a;
    a.Wp = wp; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.R = res; a.r_bs = r_bs; a.bias = bias;
    a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
    a.mtiles = (Co + 15) / 16; a.ksteps = 2 * ((Ci + 7) / 8);
    a.relu1 = relu1; a.res_mode = res_mode; a.relu2 = relu2; a.store_mode = store_mode;
    a.tiles_x = (W + CV_TW - 1) / CV_TW;
    a.vec = !(W & 3) && !(y_bs & 3) && !(r_bs & 3) && irm_aligned16(y) && irm_aligned16(res);
    if (ct <= 0) return IRM_EINVAL;
    const int nchunks = (a.mtiles + ct - 1) / ct;
    if (ygroups <= 0) ygroups = 1;
    if (ygroups > nchunks) ygroups = nchunks;
    const bool fast = a.vec && !(x_bs & 3) && irm_aligned16(x) && irm_aligned16(wp) && !irm_probe_set("IRM_CONV_GENERIC");
    if (fast) {
        switch (ct) {
            case 1: return launch_conv_ring<1>(a, B, ygroups, stream);
            case 2: return launch_conv_ring<2>(a, B, ygroups, stream);
            case 3: return launch_conv_ring<3>(a, B, ygroups, stream);
            case 4: return launch_conv_ring<4>(a, B, ygroups, stream);
            case 6: return launch_conv_ring<6>(a, B, ygroups, stream);
            default: return IRM_EINVAL;
        }
    }
    switch (ct) {
        case 1: return launch_conv<1>(a, B, ygroups, stream);
        case 2: return launch_conv<2>(a, B, ygroups, stream);
        case 3: return launch_conv<3>(a, B, ygroups, stream);
        case 4: return launch_conv<4>(a, B, ygroups, stream);
        case 6: return launch_conv<6>(a, B, ygroups, stream);
        default: return IRM_EINVAL;
    }
}
