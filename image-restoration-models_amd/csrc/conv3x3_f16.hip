// Dense 3x3 convolution (stride 1, zero pad 1) as an fp32 emulation on the fp16 matrix cores: the implicit GEMM of
// conv3x3.hip with three v_mfma_f32_16x16x32_f16 (lo*hi, hi*lo, hi*hi, fp32 accumulate) per k-step instead of eight
// f32-input MFMAs - the conv stacks of DnCNN / REDNet and the patch-embed / down / up / output convs of Restormer
// (restormer.py:156-189, 243; network_dncnn.py:40-71; rednet.py:64-136) are bound by the f32-MFMA rate there.
//
// One workgroup (8 waves) = an 8 x 32 pixel tile; wave w owns row w (2 MFMA tiles of 16 pixels) and CT output tiles.
// Per stage of 32 input channels:
//   1. the raw fp32 halo tile (32 planes of 10 x 40) arrives by LDS-DMA (16-byte chunks, border chunks read a zero page);
//   2. it is converted ONCE into a channel-minor image [340 pixels][32 ch fp16 hi | 32 ch fp16 lo] (x 2^-4 for range,
//      160-byte pixel stride: the operand reads below are bank-conflict free) - every element is split once and then
//      used by 9 taps x all output tiles;
//   3. the raw buffer is immediately refilled with the next stage while the 9 taps run: per tap and pixel tile two
//      ds_read_b128 give the A operands (lane (r, g) = pixel r, channels 8 g .. 8 g + 7); the split weights of 3 taps at a
//      time stream through a double-buffered LDS area (host packed, scaled by a power of two, L2 resident).
// Pixels are the MFMA row index, so a lane ends up with 4 consecutive pixels of one output channel: the epilogue (bias,
// ReLU, residual modes, PixelUnshuffle / PixelShuffle folded into the store) is the one of conv3x3.hip.
#include "irm_common.h"

typedef _Float16 cf_h8 __attribute__((ext_vector_type(8)));

#define CF_TH 8
#define CF_TW 32
#define CF_PLANE 400                    // raw plane: 10 rows x 40 floats (columns tx0 - 4 .. tx0 + 35)
#define CF_HC 34
#define CF_NP 340
#define CF_PXB 160                      // bytes per pixel of the fp16 image (64 hi + 64 lo + 32 pad)
#define CF_RAWB (7 * 512 * 16)           // 57344: 32 planes x 400 floats (51200 B) + the tail of the 7th DMA instruction
#define CF_IMGB (CF_NP * CF_PXB)        // 54400
#define CF_NRAW 7                       // raw DMA instructions per lane and stage (3200 chunks / 512 lanes)

__device__ __attribute__((aligned(16))) float cf_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float cf_res(float v, float r, int mode) {
    if (mode == 1) return v + r;
    if (mode == 2) return r - v;
    return fminf(fmaxf(tanhf(v) + r, -1.0f), 1.0f);
}

struct ConvF16Args {
    const float* Wp;              // [mtiles][S][9 taps][hi|lo][64 lanes][8 halves], see irm_hip.h
    const float* X; long x_bs;
    float* Y; long y_bs;
    const float* R; long r_bs;
    const float* bias;
    int Ci, Co, H, W, mtiles, S;
    int relu1, res_mode, relu2, store_mode, tiles_x;
    float inv_s;                  // 16 / weight scale
};

template <int N>
__device__ __forceinline__ void cf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NCH (round 3): a pass covers NCH chunks of CT output tiles (accumulators for all of them in registers: 8 CT NCH), so the
// input tile of a stage is fetched and converted ONCE per NCH x CT output tiles instead of once per CT - the up-sampling
// convs (Co = 2 Ci = 192 / 384 / 768: 12 / 24 / 48 output tiles) ran 3 / 6 / 4 passes over the same input.  The weights of a
// tap group still cover CT tiles (the LDS bound): NCH x 3 groups per stage instead of 3.
template <int CT, int NCH = 1>
__global__ __launch_bounds__(512, 2) void conv3x3_f16x3_kernel(ConvF16Args a) {
    IRM_KERNEL_ENTRY();
    constexpr int TT = CT * NCH;                   // output tiles per pass
    constexpr int NG = 3 * NCH;                    // weight groups per stage
    constexpr int WGB = 3 * CT * 2048;             // bytes of the weights of one tap group (3 taps)
    constexpr int NW = (WGB / 16 + 511) / 512;     // weight DMA instructions per lane and tap group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    char* raw = lds;
    char* img = lds + CF_RAWB;
    char* wbuf = img + CF_IMGB;                    // [2][WGB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.z;
    const int ty0 = (blockIdx.x / a.tiles_x) * CF_TH, tx0 = (blockIdx.x % a.tiles_x) * CF_TW;
    const float* X = a.X + (long)b * a.x_bs;
    const long plane = (long)a.H * a.W;
    const int S = a.S;
    const int nchunks = (a.mtiles + TT - 1) / TT;
    const int my_chunks = (nchunks - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;

    // raw DMA geometry of this lane (fixed over stages): chunk q = j * 512 + tid -> (channel, row, 16-byte column chunk)
    long xoff[CF_NRAW];
    int xch[CF_NRAW];
    bool xok[CF_NRAW];
#pragma unroll
    for (int j = 0; j < CF_NRAW; ++j) {
        const int q = j * 512 + tid;
        const int ch = q / 100, rem = q - ch * 100, row = rem / 10, chunk = rem - row * 10;
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + chunk * 4;
        xch[j] = ch;
        xok[j] = q < 3200 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        xoff[j] = (long)ch * plane + (long)gy * a.W + gx;
    }
    auto issue_raw = [&](int s) {
#pragma unroll
        for (int j = 0; j < CF_NRAW; ++j) {
            const bool ok = xok[j] && s * 32 + xch[j] < a.Ci;
            const float* src = ok ? X + (long)s * 32 * plane + xoff[j] : cf_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(raw + (j * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    // weight group n = (s NCH + chunk) 3 + tg (3 taps x CT tiles x hi|lo x 1 KiB) -> buffer n & 1
    auto issue_w = [&](int mt0, int n) {
        const int sc = n / 3, tg = n - 3 * sc, s = sc / NCH, chn = sc - s * NCH;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int cb = min(j * 512 + wave * 64, WGB / 16 - 64);          // surplus waves repeat the last 1 KiB (same bytes)
            const int q = cb + lane;
            const int ct = q / 384, rem = q - ct * 384;                       // 384 chunks = 3 taps x (hi | lo) x 1 KiB per output tile
            const int mt = min(mt0 + chn * CT + ct, a.mtiles - 1);
            const float* src = a.Wp + ((((long)mt * S + s) * 9 + tg * 3) * 2) * 256 + rem * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(wbuf + (n & 1) * WGB + cb * 16), 16, 0, 0);
        }
    };

    f32x4 acc[2][TT];
    float* Y = a.Y + (long)b * a.y_bs;
    const float* R = a.R ? a.R + (long)b * a.r_bs : nullptr;

    for (int ci = 0; ci < my_chunks; ++ci) {
        const int mt0 = ((int)blockIdx.y + ci * (int)gridDim.y) * TT;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < TT; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_barrier();                                        // previous pass: every LDS reader is done
        issue_raw(0);
        issue_w(mt0, 0);
        for (int s = 0; s < S; ++s) {
            cf_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                    // raw(s), W(3 s) landed; image free
            // ---- fp32 planes -> channel-minor fp16 hi / lo image
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int item = it * 512 + tid;
                if (item < 4 * CF_NP) {
                    const int cg = item / CF_NP, px = item - cg * CF_NP;
                    const int row = px / CF_HC, col = px - row * CF_HC;
                    const float* rp = reinterpret_cast<const float*>(raw) + (8 * cg) * CF_PLANE + row * 40 + col + 3;
                    cf_h8 hi, lo;
                    float xs[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xs[e] = irm_sat_h(__fmul_rn(rp[e * CF_PLANE], 0.0625f));
                    irm_split8(xs, hi, lo);                                  // one rounded value for hi and lo (irm_common.h)
                    *reinterpret_cast<cf_h8*>(img + px * CF_PXB + cg * 16) = hi;
                    *reinterpret_cast<cf_h8*>(img + px * CF_PXB + 64 + cg * 16) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // image complete; raw buffer free
            // the next requests, oldest first: the stage's second weight group, then the raw planes of the next stage
            issue_w(mt0, NG * s + 1);
            if (s + 1 < S) issue_raw(s + 1);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int n = NG * s + gi, tg = gi % 3, chn = gi / 3;
                if (gi == 1) {
                    if (s + 1 < S) cf_wait_vmcnt<CF_NRAW>(); else cf_wait_vmcnt<0>();      // group n landed (raw may be in flight)
                    __builtin_amdgcn_s_barrier();                                          // everybody is done with group n - 1
                    issue_w(mt0, n + 1);
                } else if (gi > 1) {
                    cf_wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    if (gi + 1 < NG || s + 1 < S) issue_w(mt0, n + 1);
                }
                const char* wb = wbuf + (n & 1) * WGB;
#pragma unroll
                for (int tl = 0; tl < 3; ++tl) {
                    const int dy = tg, dx = tl;
                    cf_h8 ah[2], al[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int pi = (wave + dy) * CF_HC + 16 * p + r + dx;
                        ah[p] = *reinterpret_cast<const cf_h8*>(img + pi * CF_PXB + 16 * g);
                        al[p] = *reinterpret_cast<const cf_h8*>(img + pi * CF_PXB + 64 + 16 * g);
                    }
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const cf_h8 wh = *reinterpret_cast<const cf_h8*>(wb + ((c * 3 + tl) * 2) * 1024 + lane * 16);
                        const cf_h8 wl = *reinterpret_cast<const cf_h8*>(wb + ((c * 3 + tl) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            f32x4& t = acc[p][chn * CT + c];
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p], wh, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], wl, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], wh, t, 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- epilogue (as conv3x3_ring_kernel)
#pragma unroll
        for (int c = 0; c < TT; ++c) {
            const int co = (mt0 + c) * 16 + r;
            const bool row_ok = mt0 + c < a.mtiles && co < a.Co;
            const float bv = (a.bias && row_ok) ? a.bias[co] : 0.0f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int y = ty0 + wave, x = tx0 + p * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(acc[p][c][e], a.inv_s, bv);
                    if (a.relu1) v[e] = fmaxf(v[e], 0.0f);
                }
                if (!row_ok || y >= a.H || x >= a.W) continue;
                if (a.store_mode == 0) {
                    const long off = (long)co * plane + (long)y * a.W + x;
                    if (a.res_mode) {
                        const float4 rr = *reinterpret_cast<const float4*>(R + off);
                        const float rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = cf_res(v[e], rv[e], a.res_mode);
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
    }
}

template <int CT, int NCH = 1>
static int launch_conv_f16(const ConvF16Args& a, int B, int ygroups, hipStream_t stream) {
    const size_t lds = (size_t)CF_RAWB + CF_IMGB + 2 * (3 * CT * 2048);
    static_assert((size_t)CF_RAWB + CF_IMGB + 2 * (3 * CT * 2048) <= 160 * 1024, "LDS");
    IRM_ALLOW_BIG_LDS((&conv3x3_f16x3_kernel<CT, NCH>));
    const int tiles_y = (a.H + CF_TH - 1) / CF_TH;
    dim3 grid(a.tiles_x * tiles_y, ygroups, B);
    hipLaunchKernelGGL((conv3x3_f16x3_kernel<CT, NCH>), grid, dim3(512), lds, stream, a);
    return irm_launch_status();
}

extern "C" int irm_conv3x3_f16x3_f32(const float* wp_split, float inv_scale, const float* x, long x_bs, float* y, long y_bs,
                                     const float* res, long r_bs, const float* bias, int B, int Ci, int Co, int H, int W,
                                     int relu1, int res_mode, int relu2, int store_mode, int ct, int ygroups,
                                     hipStream_t stream) {
    if (!wp_split || !x || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (res_mode < 0 || res_mode > 3 || (res_mode && !res) || store_mode < 0 || store_mode > 2) return IRM_EINVAL;
    if (store_mode != 0 && res_mode != 0) return IRM_EINVAL;
    if (store_mode == 1 && ((H & 1) || (W & 1))) return IRM_EINVAL;
    if (store_mode == 2 && (Co & 3)) return IRM_EINVAL;
    if (B > 65535 || (W & 3) || (x_bs & 3) || (y_bs & 3) || (r_bs & 3)) return IRM_EINVAL;
    if (!irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(res) || !irm_aligned16(wp_split)) return IRM_EINVAL;
    ConvF16Args a;
    a.Wp = wp_split; a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.R = res; a.r_bs = r_bs; a.bias = bias;
    a.Ci = Ci; a.Co = Co; a.H = H; a.W = W; a.mtiles = (Co + 15) / 16; a.S = (Ci + 31) / 32;
    a.relu1 = relu1; a.res_mode = res_mode; a.relu2 = relu2; a.store_mode = store_mode;
    a.tiles_x = (W + CF_TW - 1) / CF_TW; a.inv_s = inv_scale;
    const int nchunks = (a.mtiles + ct - 1) / (ct > 0 ? ct : 1);
    if (ygroups <= 0) ygroups = 1;
    if (ygroups > nchunks) ygroups = nchunks;
    switch (ct) {                     // output tiles per pass: 1 ... 4 (one weight chunk), 8 / 12 (2 / 3 chunks of 4)
        case 1: return launch_conv_f16<1>(a, B, ygroups, stream);
        case 2: return launch_conv_f16<2>(a, B, ygroups, stream);
        case 3: return launch_conv_f16<3>(a, B, ygroups, stream);
        case 4: return launch_conv_f16<4>(a, B, ygroups, stream);
        case 8: return launch_conv_f16<4, 2>(a, B, ygroups, stream);
        case 12: return launch_conv_f16<4, 3>(a, B, ygroups, stream);
        default: return IRM_EINVAL;
    }
}
