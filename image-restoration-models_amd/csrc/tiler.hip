// Device side of the tiled-patch loop (src/utils.py:353-454): cut an HWC
// uint8/uint16 image into equal-shape tiles (normalise, optional seeded noise,
// reflect pad to a multiple of 8 as utils.pad does), and blend the per-tile
// predictions back with the Gaussian window, divide by the weight map and
// requantise - all with the reference's float32 operation order so the result
// is bit-identical to the numpy code given identical predictions.
#include "irm_common.h"

struct ExtractArgs {
    const void* img;       // [H][W][C] u8 or u16
    const int* origins;    // [T][2] (y0, x0)
    const double* noise;   // [th][tw][C] float64 field (same for every tile) or null
    float* tiles;          // [T][C][ph][pw]
    int H, W, C, th, tw, ph, pw, T;
    int is_u16;
    float scale;           // 255 or 65535
    float mean, inv_std;   // DeblurGANv2 normalize: v = (raw - mean) * inv_std on the RAW integer value; 0,1 = off
    int pad_zero;          // 0: reflect pad (utils.pad), 1: zero pad (deblurganv2.pad)
};

__global__ __launch_bounds__(256) void tile_extract_kernel(ExtractArgs a) {
    IRM_KERNEL_ENTRY();
    const long total = (long)a.T * a.C * a.ph * a.pw;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % a.pw);
    long t = idx / a.pw;
    const int py = (int)(t % a.ph); t /= a.ph;
    const int c = (int)(t % a.C);
    const int tile = (int)(t / a.C);
    if (a.pad_zero && (py >= a.th || px >= a.tw)) {          // deblurganv2/__init__.py:16-24
        a.tiles[idx] = 0.0f;
        return;
    }
    // reflect (no edge repeat) for the padded rows/cols: utils.py:174-181
    const int sy = py < a.th ? py : 2 * a.th - 2 - py;
    const int sx = px < a.tw ? px : 2 * a.tw - 2 - px;
    const int gy = a.origins[tile * 2] + sy, gx = a.origins[tile * 2 + 1] + sx;
    const long src = ((long)gy * a.W + gx) * a.C + c;
    const float raw = a.is_u16 ? (float)reinterpret_cast<const unsigned short*>(a.img)[src]
                               : (float)reinterpret_cast<const unsigned char*>(a.img)[src];
    const bool albu = a.inv_std != 1.0f || a.mean != 0.0f;
    // utils.py:159-171, or albumentations Normalize (aug.py:31-39): (x - mean*255) * (1 / (std*255))
    float v = albu ? __fmul_rn(__fsub_rn(raw, a.mean), a.inv_std) : __fdiv_rn(raw, a.scale);
    if (a.noise) {                                          // utils.py:29-36
        const double d = (double)v + a.noise[((long)sy * a.tw + sx) * a.C + c];
        v = (float)fmin(fmax(d, 0.0), 1.0);
    }
    a.tiles[idx] = v;
}

extern "C" int irm_tile_extract(const void* img, int is_u16, const int* origins, const double* noise,
                                float* tiles, int H, int W, int C, int th, int tw, int ph, int pw, int T,
                                float mean, float inv_std, int pad_zero, hipStream_t stream) {
    if (!img || !origins || !tiles || H <= 0 || W <= 0 || C <= 0 || T <= 0) return IRM_EINVAL;
    if (th <= 0 || tw <= 0 || ph < th || pw < tw || th > H || tw > W) return IRM_EINVAL;
    if (!pad_zero && (ph - th >= th || pw - tw >= tw)) return IRM_EINVAL;    // reflect needs pad < extent
    ExtractArgs a{img, origins, noise, tiles, H, W, C, th, tw, ph, pw, T, is_u16,
                  is_u16 ? 65535.0f : 255.0f, mean, inv_std, pad_zero};
    const long total = (long)T * C * ph * pw;
    hipLaunchKernelGGL(tile_extract_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
struct BlendArgs {
    const float* pred;     // [T][Cp][ph][pw], only [:Co][:th][:tw] is used
    const int* origins;    // [T][2], in the reference's loop order
    const float* window;   // [ps][ps]
    void* out;             // [H][W][Co] u8 / u16
    const void* target;    // optional [H][W][Co] for the squared error
    unsigned long long* sse;   // optional single accumulator (integer: order independent)
    int H, W, Co, Cp, th, tw, ph, pw, ps, T;
    int is_u16;
    float post_scale, post_shift;   // postprocess v*scale+shift (DeblurGANv2 (x+1)/2); 1,0 = off
};

// One workgroup blends BLEND_PER_WG consecutive output elements (256 threads x 8 rounds) and adds its squared error
// with ONE integer atomic: a frame of 1280x720x3 bytes takes 1 350 atomics on the single accumulator instead of one
// per wave (43 200 same-address atomics serialised at the L2: 0.5 ms of a 57 ms frame).  Integer sums: order independent.
#define BLEND_PER_WG 2048

__global__ __launch_bounds__(256) void blend_kernel(BlendArgs a) {
    IRM_KERNEL_ENTRY();
    const long total = (long)a.H * a.W * a.Co;
    unsigned long long err = 0;
    const bool albu = a.post_scale != 1.0f || a.post_shift != 0.0f;
    const float peak = a.is_u16 ? 65535.0f : 255.0f;
#pragma unroll 1
    for (int round = 0; round < BLEND_PER_WG / 256; ++round) {
        const long idx = (long)blockIdx.x * BLEND_PER_WG + round * 256 + threadIdx.x;
        if (idx >= total) break;
        const int c = (int)(idx % a.Co);
        const long t = idx / a.Co;
        const int x = (int)(t % a.W), y = (int)(t / a.W);
        float acc = 0.0f, wsum = 0.0f;
        for (int i = 0; i < a.T; ++i) {            // same order as the h_idx / w_idx loops
            const int ly = y - a.origins[2 * i], lx = x - a.origins[2 * i + 1];
            if (ly < 0 || ly >= a.th || lx < 0 || lx >= a.tw) continue;
            float p = a.pred[(((long)i * a.Cp + c) * a.ph + ly) * a.pw + lx];
            if (albu) p = __fmul_rn(__fadd_rn(p, a.post_shift), a.post_scale);
            const float w = a.window[ly * a.ps + lx];
            acc = __fadd_rn(acc, __fmul_rn(p, w));          // utils.py:433
            wsum = __fadd_rn(wsum, w);                       // utils.py:434
        }
        float v = __fdiv_rn(acc, fmaxf(wsum, 1e-8f));        // utils.py:440
        v = rintf(fminf(fmaxf(__fmul_rn(v, peak), 0.0f), peak));   // clip, round half to even
        const unsigned q = (unsigned)v;
        if (a.is_u16) reinterpret_cast<unsigned short*>(a.out)[idx] = (unsigned short)q;
        else reinterpret_cast<unsigned char*>(a.out)[idx] = (unsigned char)q;
        if (a.target && a.sse) {
            const int tv = a.is_u16 ? reinterpret_cast<const unsigned short*>(a.target)[idx]
                                    : reinterpret_cast<const unsigned char*>(a.target)[idx];
            const long d = (long)q - tv;
            err += (unsigned long long)(d * d);
        }
    }
    if (a.target && a.sse) {
        // wave reduction, the four waves meet in LDS, one integer atomic per workgroup
        __shared__ unsigned long long part[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long s = part[0] + part[1] + part[2] + part[3];
            if (s) atomicAdd(a.sse, s);
        }
    }
}

extern "C" int irm_window_blend(const float* pred, const int* origins, const float* window, void* out,
                                int is_u16, const void* target, unsigned long long* sse, int H, int W, int Co,
                                int Cp, int th, int tw, int ph, int pw, int ps, int T, float post_scale,
                                float post_shift, hipStream_t stream) {
    if (!pred || !origins || !window || !out || H <= 0 || W <= 0 || Co <= 0 || Cp < Co || T <= 0) return IRM_EINVAL;
    if (th <= 0 || tw <= 0 || ph < th || pw < tw || th > ps || tw > ps) return IRM_EINVAL;
    BlendArgs a{pred, origins, window, out, target, sse, H, W, Co, Cp, th, tw, ph, pw, ps, T, is_u16,
                post_scale, post_shift};
    const long total = (long)H * W * Co;
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((total + BLEND_PER_WG - 1) / BLEND_PER_WG)), dim3(256), 0, stream, a);
    return irm_launch_status();
}
