// GDFN tail of the C = 192 level of Restormer in ONE kernel (round 3):
//
//   x += project_out( gelu(dw(h)[:hid]) * dw(h)[hid:] ) + bias2        (FeedForward.dwconv / chunk / gelu * / project_out and the
//                                                                       block's residual, restormer.py:84-93, 148)
//
// with h = project_in(LN(x)) written by irm_ln_gemm_presplit_cl_f16x3_f32 tile-major channel-last, its two halves padded to
// HP channels each, in chunks of 64 channels ([tile][2 HP / 64][256 pixels][64]: the producer's chunk of 64 channels is one
// contiguous run per pixel tile).  Before: a gated depth-wise kernel wrote g (hid planes) and a streaming GEMM
// read it back - 1.2 GB per 12-tile block at C = 192 - as two launches; here g never leaves the CU.
//
// This is the second half of the GDFN branch kernel of the C <= 96 levels (fused_block.hip) with the h image filled from
// HBM instead of by the project_in MFMAs: one persistent workgroup (8 waves) per CU walks (image, 8 x 32 tile) items; per
// stage of 16 gate pairs (32 channels of h) every lane fetches five or six 16-byte pieces (pixel, 4 channels) of the 10 x 34
// halo tile ONE STAGE AHEAD into registers (channel-last: whole lines, nothing wasted; halo pixels outside the image read a
// zero page - the reference zero-pads h) and parks them in the channel-minor LDS image [pixel][32 channels + pad] (160-byte
// pixel stride: the stencil's ds_read_b128 are conflict free) that fused_block.hip's stencil expects; the depth-wise 3x3,
// the erf-GELU gate, the fp16 hi/lo split of the gate (x 2^-4, saturating) and the fp32-emulated project_out (three
// v_mfma_f32_16x16x32_f16 per product, CT = C / 16 = 12 accumulator tiles x 2 rows per wave) are that kernel's.  The block's
// residual is read and written in place in the epilogue (a tile reads h with its halo, x only inside the tile).
#include "irm_common.h"
#include <utility>

typedef _Float16 ft_h8 __attribute__((ext_vector_type(8)));
typedef unsigned ft_u4 __attribute__((ext_vector_type(4)));

#define FT_TH 8
#define FT_TW 32
#define FT_HC (FT_TW + 2)
#define FT_NP ((FT_TH + 2) * FT_HC)       // 340 halo pixels
#define FT_PS 40                          // floats per pixel of the LDS image
#define FT_PLF ((FT_NP + 16) * FT_PS)     // + 16 junk pixels (lanes without a pixel park there)

__device__ __attribute__((aligned(16))) float ft_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct TailArgs {
    const float* Hh; long h_bs;           // [B][tiles][CB / 64][256][64]
    float* X; long x_bs;                  // [B][C][H][W], updated in place
    const float* rec;                     // [S][512]: depth-wise taps [10][32] of stage s (9 taps + bias; second half x 2^-4), pad
    const float* w2;                      // [ceil(S/2)][CT][hi|lo][64 lanes][8 halves], as irm_gdfn_fused_f16x3_f32
    const float* bias2;                   // [C] or null
    int C, H, W, S, HP, CB;
    float inv_s2;
    int tiles_x, tiles, items, gpx;
};

// gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2), Abramowitz-Stegun 7.1.26 with folded constants (fused_block.hip)
__device__ __forceinline__ float ft_gelu1(float x) {
    constexpr double CU = 0.84932180028801904272;
    constexpr double F = 0.5 / CU;
    constexpr float k = (float)(0.3275911 / 1.2011224087864498);
    constexpr float a1 = (float)(0.254829592 * F), a2 = (float)(-0.284496736 * F), a3 = (float)(1.421413741 * F),
                    a4 = (float)(-1.453152027 * F), a5 = (float)(1.061405429 * F);
    const float u = fabsf(x) * (float)CU;
    const float t = __builtin_amdgcn_rcpf(fmaf(u, k, 1.0f));
    float p = fmaf(t, a5, a4);
    p = fmaf(p, t, a3);
    p = fmaf(p, t, a2);
    p = fmaf(p, t, a1);
    const float w = (p * t) * __builtin_amdgcn_exp2f(-u * u);
    return fmaf(-u, w, __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f));
}

typedef __attribute__((address_space(3))) char ft_lc;
__device__ __forceinline__ unsigned ft_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
template <typename T>
__device__ __forceinline__ T ft_ld(const ft_lc* base, unsigned voff, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(base + voff + imm);
}
template <typename T>
__device__ __forceinline__ void ft_st(ft_lc* base, unsigned voff, int imm, T v) {
    *reinterpret_cast<__attribute__((address_space(3))) T*>(base + voff + imm) = v;
}
template <int IMM, typename T>
__device__ __forceinline__ void ft_dsr(T& d, unsigned voff) {
    static_assert(sizeof(T) == 16 && IMM >= 0 && IMM < 65536, "ds_read_b128");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(voff), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void ft_waitcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <typename T>
__device__ __forceinline__ void ft_tie1(T& r) { asm volatile("" : "+v"(r)); }
template <typename... T>
__device__ __forceinline__ void ft_tie(T&... r) { (ft_tie1(r), ...); }
template <int I> using ft_ic = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void ft_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ft_ic<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ft_for(F&& f) { ft_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int NP>
__device__ __forceinline__ void ft_dma(const float* src, float* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NP + 7) / 8; ++i) {
        const int pc = wave + 8 * i;
        if (pc < NP)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    }
}

template <int CT>
__global__ __launch_bounds__(512, 2) void gdfn_tail_kernel(TailArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int RECF = 512;                      // floats per tap record
    constexpr int W2F = CT * 512, W2P = CT * 2;
    constexpr int SLOT_B = RECF * 4, W2_OFF = 2 * SLOT_B, PL_OFF = W2_OFF + W2F * 4, PL_B = FT_PLF * 4;
    constexpr int PARK_OFF = PL_OFF + 2 * PL_B;
    constexpr int NK = (FT_NP * 8 + 511) / 512;    // 16-byte pieces per lane and stage (6; the last one partly)
    static_assert(PARK_OFF + 8192 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ft_lc* lds = (ft_lc*)smem;
    float* slots = smem;
    float* w2a = smem + W2_OFF / 4;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long plane = (long)a.H * a.W;
    const int S = a.S;
    const int per = (a.items + 7) >> 3;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    auto item_of = [&](int round) { const int i = round * a.gpx + pos; return i < per ? xcd * per + i : a.items; };

    int round = 0;
    int item = item_of(0);
    if (item >= a.items) return;

    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, g = lane >> 4, r = lane & 15;
        const int b = item / a.tiles, tile = item - b * a.tiles;
        const int ty0 = (tile / a.tiles_x) * FT_TH, tx0 = (tile % a.tiles_x) * FT_TW;
        const float* Hh = a.Hh + (long)b * a.h_bs;
        const unsigned vw = ft_opaque((unsigned)(lane * 16));
        const unsigned vc = ft_opaque((unsigned)(16 * g));
        const int sp0 = (2 * (wave >> 1)) * FT_HC + 16 * (wave & 1) + r;
        const unsigned vp0 = ft_opaque((unsigned)(PL_OFF + (sp0 * FT_PS + 4 * g) * 4));
        const unsigned vp1 = ft_opaque(vp0 + PL_B);

        // this lane's pieces of a stage: piece idx = k 512 + tid -> (halo pixel idx >> 3, 16-byte piece idx & 7: channels
        // 4 (piece & 3) .. + 3 of half piece >> 2).  goff: float offset of the piece in h for stage 0, or -1 (pixel outside the
        // image / beyond the tile: the zero page); loff: byte offset in LDS image 0
        int goff[NK];
        unsigned loff[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = k * 512 + tid, px = idx >> 3, pc = idx & 7;
            const int hr = px / FT_HC, hc = px - hr * FT_HC;
            const int gy = ty0 - 1 + hr, gx = tx0 - 1 + hc;
            const bool valid = px < FT_NP, inside = valid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            // h: [tile][CB / 64 chunks][256 pixels][64 channels]; the multiplier half starts HP / 64 chunks further on
            goff[k] = inside ? ((gy >> 3) * a.tiles_x + (gx >> 5)) * 256 * a.CB + ((gy & 7) * FT_TW + (gx & 31)) * 64 + (pc & 3) * 4 +
                               (pc >> 2) * (a.HP / 64) * (256 * 64)
                             : -1;
            loff[k] = (unsigned)(PL_OFF + ((valid ? px : FT_NP + (tid & 15)) * FT_PS) * 4 + (pc & 3) * 16 + (pc >> 2) * 64);
        }
        f32x4 hreg[NK];
        auto load_h = [&](int s) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
#ifdef FT_NO_LOAD                                  // (diagnostic: the kernel without its HBM reads of h)
                const float* src = ft_zero_page;
#else
                const float* src = goff[k] >= 0 ? Hh + goff[k] + (s >> 2) * (256 * 64) + (s & 3) * 16 : ft_zero_page;
#endif
                hreg[k] = *reinterpret_cast<const f32x4*>(src);
            }
        };
        auto park_h = [&](int img) {
#pragma unroll
            for (int k = 0; k < NK; ++k) ft_st<f32x4>(lds, loff[k], img * PL_B, hreg[k]);
        };

        const int nitem = item_of(round + 1);

        // the previous item's last barrier has passed: every LDS region is free
        ft_dma<2>(a.rec, slots, wave, lane);                                // taps of stage 0 -> slot 0
        ft_dma<W2P>(a.w2, w2a, wave, lane);
        load_h(0);
        f32x4 acc2[2][CT];
        const unsigned vpark = ft_opaque((unsigned)(PARK_OFF + threadIdx.x * 16));
#pragma unroll
        for (int c = 0; c < CT; ++c) { acc2[0][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[1][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        ft_st<f32x4>(lds, vpark, 0, acc2[1][CT - 1]);
        park_h(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        float o[2][2][4];
        f32x4 kprev[3];
        auto st_comp = [&](int hf, int dy, const f32x4 (&P)[3], const f32x4 (&kc)[3], const f32x4& kb) {
            if (dy == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[hf][q][e] = kb[e];
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (dy < 3) o[hf][0][e] = fmaf(kc[dx][e], P[dx][e], o[hf][0][e]);
                    if (dy > 0) o[hf][1][e] = fmaf(kprev[dx][e], P[dx][e], o[hf][1][e]);
                }
            }
            if (dy < 3) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) kprev[dx] = kc[dx];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(o[hf][q][e]));
        };

        ft_u4 Gh[2], Gl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { Gh[q] = (ft_u4){0u, 0u, 0u, 0u}; Gl[q] = (ft_u4){0u, 0u, 0u, 0u}; }

        // iteration it: stencil + gate of stage it (image it & 1, taps in slot it & 1); the pieces of stage it + 1 are requested
        // at its start and parked in the other image at its end; project_out every second stage
        auto iter = [&](auto PAR, auto MORE, int it) {
            constexpr int par = decltype(PAR)::value;
            constexpr bool more = decltype(MORE)::value;
            if constexpr (more) {
                load_h(it + 1);
                ft_dma<2>(a.rec + (long)(it + 1) * RECF, slots + (par ^ 1) * RECF, wave, lane);
            }
            if (par == 0 && it > 0) ft_dma<W2P>(a.w2 + (long)(it >> 1) * W2F, w2a, wave, lane);
            {
                const unsigned vp = par ? vp1 : vp0;
                f32x4 P[2][3], K[2][3], KB[1];
                auto loads = [&](auto IC, auto NC) {
                    constexpr int I = decltype(IC)::value, n = decltype(NC)::value;
                    constexpr int hf = I >> 2, dy = I & 3;
                    constexpr int cf = par * SLOT_B + hf * 64;
                    if constexpr (dy == 0) ft_dsr<cf + 9 * 128>(KB[0], vc);
                    if constexpr (dy < 3) {
                        ft_dsr<cf + (dy * 3 + 0) * 128>(K[n][0], vc);
                        ft_dsr<cf + (dy * 3 + 1) * 128>(K[n][1], vc);
                        ft_dsr<cf + (dy * 3 + 2) * 128>(K[n][2], vc);
                    }
                    ft_dsr<(dy * FT_HC + 0) * (FT_PS * 4) + hf * 64>(P[n][0], vp);
                    ft_dsr<(dy * FT_HC + 1) * (FT_PS * 4) + hf * 64>(P[n][1], vp);
                    ft_dsr<(dy * FT_HC + 2) * (FT_PS * 4) + hf * 64>(P[n][2], vp);
                };
                loads(ft_ic<0>{}, ft_ic<0>{});
                ft_for<8>([&](auto IC) {
                    constexpr int I = decltype(IC)::value, c = I & 1, n = c ^ 1;
                    constexpr int hf = I >> 2, dy = I & 3;
                    if constexpr (I + 1 < 8) loads(ft_ic<I + 1>{}, ft_ic<n>{});
                    constexpr int J = I + 1, jdy = J & 3;
                    constexpr int nnext = J < 8 ? 3 + (jdy < 3 ? 3 : 0) + (jdy == 0 ? 1 : 0) : 0;
                    ft_waitcnt<nnext>();
                    ft_tie(P[c][0], P[c][1], P[c][2]);
                    if constexpr (dy < 3) ft_tie(K[c][0], K[c][1], K[c][2]);
                    if constexpr (dy == 0) ft_tie(KB[0]);
                    st_comp(hf, dy, P[c], K[c], KB[0]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // (taps and bias of the second half are pre-scaled by 2^-4 on the host: o[1] = dw(h2) / 16, exact)
                    const float g0 = irm_sat_h(__fmul_rn(ft_gelu1(o[0][q][e]), o[1][q][e]));
                    const float g1 = irm_sat_h(__fmul_rn(ft_gelu1(o[0][q][e + 1]), o[1][q][e + 1]));
                    unsigned hh, ll;
                    irm_split2(g0, g1, hh, ll);
                    Gh[q][2 * par + e / 2] = hh;
                    Gl[q][2 * par + e / 2] = ll;
                }
#pragma unroll
            for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(Gh[q]), "+v"(Gl[q]));
            if (par == 1 || !more) {
                if constexpr (par == 0) {
                    // odd stage count: this super-stage's project_out weights were requested in this very iteration
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                f32x4 accp = ft_ld<f32x4>(lds, vpark, 0);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const ft_h8 bh = ft_ld<ft_h8>(lds, vw, W2_OFF + (c * 2) * 1024);
                    const ft_h8 bl = ft_ld<ft_h8>(lds, vw, W2_OFF + (c * 2 + 1) * 1024);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4& t = (q == 1 && c == CT - 1) ? accp : acc2[q][c];
                        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ft_h8, Gl[q]), bh, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ft_h8, Gh[q]), bl, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ft_h8, Gh[q]), bh, t, 0, 0, 0);
                    }
                }
                ft_st<f32x4>(lds, vpark, 0, accp);
#pragma unroll
                for (int q = 0; q < 2; ++q) { Gh[q] = (ft_u4){0u, 0u, 0u, 0u}; Gl[q] = (ft_u4){0u, 0u, 0u, 0u}; }
            }
            if constexpr (more) park_h(par ^ 1);            // (the other image: its stencil finished before the last barrier)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        {
            const std::integral_constant<int, 0> P0; const std::integral_constant<int, 1> P1;
            const std::true_type T; const std::false_type F;
            int it = 0;
            for (; it + 2 < S; it += 2) { iter(P0, T, it); iter(P1, T, it + 1); }
            if (it + 2 == S) { iter(P0, T, it); iter(P1, F, it + 1); }
            else iter(P0, F, it);
        }

        // ------------------------------------------------------------ epilogue: x += acc / s2 + bias2, 16-byte accesses, in place
        {
            int t3 = threadIdx.x;
            asm volatile("" : "+v"(t3));
            const int r3 = t3 & 15, ox = tx0 + 16 * (wave & 1) + 4 * ((t3 & 63) >> 4), oy0 = ty0 + 2 * (wave >> 1);
            float* X = a.X + (long)b * a.x_bs;
            acc2[1][CT - 1] = ft_ld<f32x4>(lds, vpark, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = 16 * c + r3;
                if (co >= a.C || ox >= a.W) continue;
                const float bv = a.bias2 ? a.bias2[co] : 0.0f;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (oy0 + q >= a.H) continue;
                    float4* px = reinterpret_cast<float4*>(X + (long)co * plane + (long)(oy0 + q) * a.W + ox);
                    const float4 xv = *px;
                    *px = make_float4(fmaf(acc2[q][c][0], a.inv_s2, xv.x + bv), fmaf(acc2[q][c][1], a.inv_s2, xv.y + bv),
                                      fmaf(acc2[q][c][2], a.inv_s2, xv.z + bv), fmaf(acc2[q][c][3], a.inv_s2, xv.w + bv));
                }
            }
        }
        if (nitem >= a.items) break;
        item = nitem;
        ++round;
        // (no barrier: behind the last iteration's barrier a wave touches only its own parked tile)
    }
}

extern "C" int irm_gdfn_tail_f16x3_f32(const float* h_cl, long h_bs, const float* rec, const float* w2, const float* bias2,
                                       float* x, long x_bs, float inv_s2, int B, int C, int hid, int hid_pad, int H, int W,
                                       hipStream_t stream) {
    if (!h_cl || !rec || !w2 || !x || B <= 0 || C <= 0 || hid <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (C != 192 || (H & 7) || (W & 31) || hid_pad < hid || (hid_pad & 63)) return IRM_EINVAL;
    if ((h_bs & 3) || (x_bs & 3) || !irm_aligned16(h_cl) || !irm_aligned16(x) || !irm_aligned16(rec) || !irm_aligned16(w2)) return IRM_EINVAL;
    if ((long)(H / 8) * (W / 32) * 256 * 2 * hid_pad >= (1L << 31)) return IRM_EINVAL;       // 32-bit piece offsets
    TailArgs a;
    a.Hh = h_cl; a.h_bs = h_bs; a.X = x; a.x_bs = x_bs; a.rec = rec; a.w2 = w2; a.bias2 = bias2;
    a.C = C; a.H = H; a.W = W; a.S = (hid + 15) / 16; a.HP = hid_pad; a.CB = 2 * hid_pad; a.inv_s2 = inv_s2;
    a.tiles_x = W / FT_TW; a.tiles = a.tiles_x * (H / FT_TH); a.items = B * a.tiles;
    constexpr int CT = 12;
    const size_t lds = (size_t)(2 * 512 + CT * 512 + 2 * FT_PLF + 2048) * sizeof(float);
    IRM_ALLOW_BIG_LDS((&gdfn_tail_kernel<CT>));
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return IRM_ELAUNCH;
    const int per = (a.items + 7) >> 3;
    a.gpx = (n + 7) / 8;
    if (a.gpx > per) a.gpx = per;
    hipLaunchKernelGGL((gdfn_tail_kernel<CT>), dim3(a.gpx * 8), dim3(512), lds, stream, a);
    return irm_launch_status();
}
