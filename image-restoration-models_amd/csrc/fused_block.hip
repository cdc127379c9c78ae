// Whole-branch fusion for the C <= 96 levels of Restormer: LayerNorm -> 1x1 conv -> depth-wise 3x3 -> ... in ONE
// kernel, so that the wide intermediate tensor (2*hid or 3*C channels per pixel) never reaches HBM.
//
//   gdfn_fused_kernel   y = x + project_out( gelu(dw(h)[:hid]) * dw(h)[hid:] ),  h = project_in(LN(x))
//                       (FeedForward + norm2 + residual, restormer.py:25-70, 76-93, 148)
//
// One workgroup (8 waves) owns an 8 x 32 pixel tile.  Its 10 x 34 halo pixels x C input channels are read ONCE,
// normalised (statistics computed in registers: all channels of a pixel sit in 4 lanes), split into fp16 hi + lo
// and kept in REGISTERS as MFMA operands (22 tiles of 16 pixels over 8 waves; lane (r, g) holds pixel r, channels
// 32 ks + 8 g + j).  The hidden channels are produced in stages of 16 gate pairs (32 channels): stage weights
// (host packed, L2 resident) arrive by LDS-DMA, three v_mfma_f32_16x16x32_f16 per product (lo*hi, hi*lo, hi*hi,
// fp32 accumulate - the emulation of irm_gemm1x1_f16x3_f32) with the WEIGHTS as the A operand, so that a lane
// receives 4 consecutive hidden channels of one pixel and parks them with one ds_write_b128 in a channel-minor
// LDS image [pixel][32 channels + pad] (160-byte pixel stride: the stencil's ds_read_b128 are conflict free).
// The depth-wise stencil then reads, per lane, pixel r of two vertically adjacent output rows and 4 + 4 channels
// (packed-fp32 FMAs over channel pairs), gates them, splits the result (x 2^-4) into fp16 hi/lo: exactly the
// k-slots lane (r, g) owns in the 16x16x32 MFMA of project_out, accumulated over all stages in registers.  The
// halo pixels outside the image read as zero in the stencil (the reference zero-pads h, not x): their image slots are
// zeroed once per item and never written.  The 1x1
// conv of stage s+1 and the stencil of stage s sit in one barrier interval (double-buffered LDS image), so the
// matrix and vector pipes overlap.
#include "irm_common.h"
#include <utility>
#ifdef FB_STAMP
#include <stdlib.h>
#endif

typedef _Float16 fb_h8 __attribute__((ext_vector_type(8)));
typedef float fb_v2 __attribute__((ext_vector_type(2)));
typedef unsigned fb_u4 __attribute__((ext_vector_type(4)));

#ifndef FB_INTERLEAVE
#define FB_INTERLEAVE 3      // plain VALU instructions scheduled behind each MFMA of a GEMM unit
#endif
#define FB_TH 8
#define FB_TW 32
#define FB_HC (FB_TW + 2)                 // halo columns
#define FB_NP ((FB_TH + 2) * FB_HC)       // halo pixels (340)
#define FB_PT ((FB_NP + 15) / 16)         // 16-pixel MFMA tiles of the halo (22)
#define FB_PS 40                          // floats per pixel in the LDS image (32 channels + 8 pad)
#define FB_PLF ((FB_NP + 16) * FB_PS)     // floats per LDS image (+ 16 junk pixels: lanes without a pixel store there, no branch)

struct FusedArgs {
    const float* X; long x_bs;            // [B][C][H][W]
    float* Y; long y_bs;                  // [B][C][H][W], Y != X (neighbouring tiles read each other's halo)
    const float* rec;                     // [S + 1] iteration records, see irm_hip.h
    const float* w2;                      // [ceil(S/2)] project_out super-stages
    const float* bias2;                   // [C] or null
    int C, H, W, S, M;                    // S = ceil(hid / 16) (GATE) or ceil(M / 32) stages; M = output channels (!GATE)
    int ln_mode; float eps, inv_s1, inv_s2;
    int tiles_x, tiles;                   // tiles per image
    int items, gpx;                       // B * tiles; workgroups per XCD
    // APPLY (irm_attn_gdfn_fused_f16x3_f32): the attention branch's last step in this kernel's prologue,
    // x' = x + bias_o + Mfold[b] v, so that x' never reaches HBM
    const float* V; long v_bs;            // [B][C][H][W] values (after qkv_dwconv)
    const float* mf; long mf_bs;          // [B][CT][KS][hi|lo][64 lanes][8 halves] folded per-image matrix (mdta_finalize, fragment order)
    const float* bias_o;                  // [C] attention project_out bias or null
    int tm;                               // !GATE: q, k (channels < 2C) stored tile-major [tile][2C][256] inside Y's q, k part
    // Tile-major, channel-LAST activations between the kernels of a stage (whole 8 x 32 tiles): element (channel, y, x) of an
    // image at (((y >> 3) tiles_x + (x >> 5)) 256 + (y & 7) 32 + (x & 31)) Cb + channel, Cb = channels of that tensor.  A pixel's
    // channels are contiguous: lane (pixel, g) fetches its 8 channels of a k-step as two 16-byte loads (planar: eight 4-byte
    // loads from eight planes), every fetched 128-byte line is used whole (planar: the 34-pixel halo rows start one pixel
    // before a line boundary - ~1.9x the bytes), and a tile's stores stay inside one contiguous Cb KiB block.
    int x_tm, v_tm, y_tm;                 // X; V (APPLY input / !GATE output channels >= 2C); Y (GATE && YCL output)
#ifdef FB_STAMP
    unsigned long long* dbg;
#endif
};

__device__ __forceinline__ fb_v2 fb_gelu2(fb_v2 x) {
    // erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), branch free (as dwgemm.hip)
    const fb_v2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const fb_v2 d = z * 0.3275911f + 1.0f;
    const fb_v2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    fb_v2 p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const fb_v2 q = z * z * -1.4426950408889634f;
    const fb_v2 e = p * t * (fb_v2){__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const fb_v2 w = 1.0f - e;
    const fb_v2 sg = {copysignf(w.x, x.x), copysignf(w.y, x.y)};
    const fb_v2 h = x * 0.5f;
    return sg * h + h;
}

// gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2), erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): with
// u = |x| sqrt(log2(e) / 2) the exponential is exp2(-u u), the rational argument 1 + p z = 1 + (p / sqrt(log2 e)) u, and
// the factor 0.5 |x| = u * (0.5 / c_u) is folded into the polynomial's coefficients: 14 instructions (two of them
// transcendental) instead of 18 for the sign-select form 0.5 x (1 + copysign(1 - e, x)); no cancellation beyond a factor
// of two anywhere (x > 0: x - [<= x / 2]).
__device__ __forceinline__ float fb_gelu1(float x) {
    constexpr double CU = 0.84932180028801904272;            // sqrt(log2(e) / 2)
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
    const float w = (p * t) * __builtin_amdgcn_exp2f(-u * u);     // 0.5 / c_u * erfc(|x| / sqrt 2)
    return fmaf(-u, w, __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f));      // (med3: max(x, 0) without fmaxf's canonicalising v_max)
}

// LDS accesses as (per-lane byte offset in a VGPR) + (compile-time immediate < 64 KiB): the offsets are made opaque
// once, otherwise the compiler materialises one address register per distinct constant beyond the 16-bit DS
// offset field of the 150 KiB layout and spills them.
typedef __attribute__((address_space(3))) char fb_lc;
__device__ __forceinline__ unsigned fb_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
template <typename T>
__device__ __forceinline__ T fb_ld(const fb_lc* base, unsigned voff, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(base + voff + imm);
}
template <typename T>
__device__ __forceinline__ void fb_st(fb_lc* base, unsigned voff, int imm, T v) {
    *reinterpret_cast<__attribute__((address_space(3))) T*>(base + voff + imm) = v;
}

// Hand-counted LDS reads for the main loop: hipcc (ROCm 7.2) waits lgkmcnt(0) before the first use of ANY pending
// ds_read there, which serialises the prefetch of the next chunk behind the lock-step read burst of all 8 waves.
// These reads are invisible to its bookkeeping; fb_wait<N>(regs...) waits until at most N newer LDS operations
// are outstanding and ties the registers to the wait (consumers cannot be scheduled above it).
template <int IMM, typename T>
__device__ __forceinline__ void fb_dsr(T& d, unsigned voff) {
    static_assert(sizeof(T) == 16 && IMM >= 0 && IMM < 65536, "ds_read_b128");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(voff), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void fb_waitcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <typename T>
__device__ __forceinline__ void fb_tie1(T& r) { asm volatile("" : "+v"(r)); }
template <typename... T>
__device__ __forceinline__ void fb_tie(T&... r) { (fb_tie1(r), ...); }
template <int I> using fb_ic = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void fb_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(fb_ic<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void fb_for(F&& f) { fb_for_impl(f, std::make_integer_sequence<int, N>{}); }

// NP pieces of 1 KiB, piece i issued by wave i % 8 (LDS destination = wave-uniform base + lane * 16)
#ifdef FB_STAMP
// diagnostic build (tools/build_variant.sh -DFB_STAMP): phase cycle sums of wave 0 of every workgroup -> a.dbg; never in the product
#define FB_T(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     stamp[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define FB_T(i) do { } while (0)
#endif

// input loads of the branch kernels (-DFB_NT: non-temporal, an experiment)
__device__ __forceinline__ float fb_gld(const float* p) {
#ifdef FB_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <int NP>
__device__ __forceinline__ void fb_dma(const float* src, float* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NP + 7) / 8; ++i) {
        const int pc = wave + 8 * i;
        if (pc < NP)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    }
}

// GATE: the GDFN branch (header).  !GATE: the front half of MDTA - y[3C] = dw3x3(qkv(LN(x))) (restormer.py:105-106,
// 116-117 + norm1), stages of 32 output channels, the stencil outputs go straight to HBM (one iteration late, so that
// the iteration-end vmcnt(0) never waits for a fresh store).
//
// Persistent: one workgroup per CU walks its share of the (image, tile) items; the raw input of the NEXT item is
// requested during the last stage of the current one (the operand registers of the finished GEMM are free by then),
// so its HBM latency and the epilogue stores overlap compute, and nothing is paid per tile for workgroup launch.
//
// APPLY (GATE only): the tile's input is x' = x + bias_o + Mfold[b] v (restormer.py:131, 147: project_out(attn @ v) + x with
// the per-image matrix Mfold = W_out blockdiag(softmax) folded by mdta_finalize), computed here for the 340 halo pixels
// instead of by a 1x1-conv launch that writes x' and this kernel reading it back.  v arrives like x (pixel on the
// lane, channels 32 ks + 8 g + e in the registers: the B operand of the MFMA, x 2^-4, split into fp16 hi/lo), the
// matrix fragments of the tile's image by LDS-DMA into the still-free image area; the accumulators start from
// (x + bias_o) / 16.  The MFMA leaves lane (r, g) with channels 16 t + 4 g + e of pixel r, so in this variant x is
// loaded in that order: k-slot (ks, g, e) of project_in <-> channel 16 (2 ks + (e >> 2)) + 4 g + (e & 3) (the host
// packs project_in's columns accordingly; LayerNorm does not care about the order).  C % 16 == 0.
// YCL (APPLY only): y is written in the tile-major channel-last layout (FusedArgs).  project_out then runs with its
// operands swapped (weights = A: the MFMA leaves lane (pixel, g) with channels 16 c + 4 g + e - 16-byte stores), and the
// residual x' reaches the accumulators through a pixel-major LDS image instead of the transposed one.
// XVCL: x (and v) are channel-last at compile time - the instantiation of the blocks inside a stage carries no planar load
// path (its 24 + 24 plane base addresses live in scalar registers across the item loop and spill into vector lanes).
template <int KS, int CT, bool GATE, bool APPLY = false, bool YCL = false, bool XVCL = false>
__global__ __launch_bounds__(512, 2) void lnpw_dw_fused_kernel(FusedArgs a) {
    static_assert(!APPLY || GATE, "APPLY is a variant of the GDFN kernel");
    static_assert(!YCL || APPLY, "YCL is a variant of the APPLY kernel");
    IRM_KERNEL_ENTRY();
    constexpr int W1F = KS * 1024;                 // floats of project_in weights per record (2 tiles x KS x hi/lo x 1 KiB)
    constexpr int RECF = W1F + 512;                // + depth-wise taps [10][32], bias [32], pad
    constexpr int RECP = KS * 4 + 2;               // 1 KiB pieces per record
    constexpr int W2F = GATE ? CT * 512 : 0, W2P = CT * 2;
    // LDS map (bytes): [2 record slots][project_out super-stage][2 images of 340 pixels x 160 B]
    constexpr int SLOT_B = RECF * 4, W2_OFF = 2 * SLOT_B, PL_OFF = W2_OFF + W2F * 4, PL_B = FB_PLF * 4;
    constexpr int CF_OFF = W1F * 4;                // taps within a slot
    constexpr int RT = 264;                        // floats per channel row of the residual transpose (inside the image area)
    constexpr int PARK_OFF = PL_OFF + 2 * PL_B;    // GATE: one accumulator tile per thread (8 KiB)
    static_assert(!GATE || 32 * KS * RT * 4 <= 2 * PL_B, "residual transpose must fit the image area");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fb_lc* lds = (fb_lc*)smem;
    float* slots = smem;
    float* w2a = smem + W2_OFF / 4;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long plane = (long)a.H * a.W;
    const int S = a.S;
    // item order: workgroups with equal blockIdx % 8 share an XCD (round-robin placement, speed only): they walk one
    // contiguous eighth of the items, 'a.gpx' neighbours per round
    const int per = (a.items + 7) >> 3;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    auto item_of = [&](int round) { const int i = round * a.gpx + pos; return i < per ? xcd * per + i : a.items; };

#ifdef FB_STAMP
    unsigned long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
    int round = 0;
    int item = item_of(0);
    if (item >= a.items) return;
#ifdef FB_STAGGER      // experiment (variant build): workgroups start FB_STAGGER x 64 cycles apart in 8 phases
    for (int d = 0; d < (int)((blockIdx.x >> 3) & 7) * FB_STAGGER; ++d) __builtin_amdgcn_s_sleep(1);
#endif
    float xr[3][KS][8];

    for (;;) {
        // Everything derived from the lane id is recomputed per item from an opaque copy: hoisted out of this loop,
        // the lane-constant addresses and masks (dozens of registers) stay live across the whole item and spill.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, g = lane >> 4, r = lane & 15;
        // lane geometry inside a tile (item independent)
        int hr[3], hc[3];
        bool pv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 16 * (wave + 8 * j) + r;
            pv[j] = p < FB_NP;
            hr[j] = p / FB_HC;
            hc[j] = p - hr[j] * FB_HC;
        }
        int klim[KS];                                  // channel 32 ks + 8 g + e exists <=> e < klim[ks] (compared where used:
#pragma unroll                                     // 24 lane masks held in scalar registers spill)
        for (int ks = 0; ks < KS; ++ks)
            klim[ks] = APPLY ? min(8, max(0, 4 * (a.C / 16 - 2 * ks))) : a.C - 32 * ks - 8 * g;
        const unsigned vw = fb_opaque((unsigned)(lane * 16));             // MFMA weight operands: lane-linear 16-byte pieces
        const unsigned vc = fb_opaque((unsigned)(16 * g));                // taps / bias of the lane's channel quad
        const int sp0 = (2 * (wave >> 1)) * FB_HC + 16 * (wave & 1) + r;  // top-left tap in halo coordinates
        const unsigned vp0 = fb_opaque((unsigned)(PL_OFF + (sp0 * FB_PS + 4 * g) * 4));
        const unsigned vp1 = fb_opaque(vp0 + PL_B);

        // raw input of an item: lane (r, g) -> halo pixel r of its 3 MFMA tiles, channels 32 ks + 8 g + e.  Addresses are
        // (wave-uniform 64-bit base of channel 32 ks + e) + (32-bit lane offset): 12 offset registers instead of 72 pointers.
        // Pixels outside the image read a clamped address (zeroed after the GEMM), channel groups beyond C a clamped
        // group (masked by kin).
        auto load_x = [&](int item) {
            // its own opaque lane id: this runs in the last iteration, and none of its lane geometry may be live (in
            // registers) across the iterations before it
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int r2 = t2 & 15, g2 = (t2 & 63) >> 4;
            const int b = item / a.tiles, tile = item - b * a.tiles;
            const int ty0 = (tile / a.tiles_x) * FB_TH, tx0 = (tile % a.tiles_x) * FB_TW;
            const float* X = a.X + (long)b * a.x_bs;
            const long plane = (long)a.H * a.W;
            if (XVCL || a.x_tm) {                          // tile-major, channel-last: two 16-byte loads per k-step
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int p = 16 * (wave + 8 * j) + r2, ph = p / FB_HC, pc = p - ph * FB_HC;
                    const int gy = min(max(ty0 - 1 + ph, 0), a.H - 1), gx = min(max(tx0 - 1 + pc, 0), a.W - 1);
                    const float* xp = X + (unsigned)((((gy >> 3) * a.tiles_x + (gx >> 5)) * 256 + (gy & 7) * FB_TW + (gx & 31)) * a.C);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        int c0, c1;
                        if constexpr (APPLY) {
                            const int tmax = a.C / 16 - 1;
                            c0 = 16 * min(2 * ks, tmax) + 4 * g2;
                            c1 = 16 * min(2 * ks + 1, tmax) + 4 * g2;
                        } else {
                            c0 = 32 * ks + 8 * min(g2, max((a.C - 32 * ks - 8) / 8, 0));
                            c1 = c0 + 4;
                        }
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(xp + c0), hi4 = *reinterpret_cast<const f32x4*>(xp + c1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { xr[j][ks][e] = lo4[e]; xr[j][ks][4 + e] = hi4[e]; }
                    }
                }
                return;
            }
            if constexpr (!XVCL)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int p = 16 * (wave + 8 * j) + r2, ph = p / FB_HC, pc = p - ph * FB_HC;
                const int gy = min(max(ty0 - 1 + ph, 0), a.H - 1), gx = min(max(tx0 - 1 + pc, 0), a.W - 1);
                const unsigned pix = (unsigned)(gy * a.W + gx);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned off = pix + (unsigned)(8 * min(g2, max((a.C - 32 * ks - 8) / 8, 0)) * plane);
                    if constexpr (APPLY) {
                        // x in the order the apply MFMA leaves it: channel 16 (2 ks + (e >> 2)) + 4 g + (e & 3); v in k-slot order
                        const unsigned offx = pix + (unsigned)(4 * g2 * plane);
                        const int tmax = a.C / 16 - 1;                 // 16-channel tiles beyond C: a clamped tile (masked)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            xr[j][ks][e] = fb_gld((X + (long)(16 * min(2 * ks + (e >> 2), tmax) + (e & 3)) * plane) + offx);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xr[j][ks][e] = fb_gld((X + (long)(32 * ks + e) * plane) + off);
                    }
                }
            }
        };

        // APPLY: v of an item (k-slot order: channel 32 ks + 8 g + e), requested at the start of the item: kept across the
        // item boundary beside x (144 registers live around the loop's back edge) the compiler spills a third of x.
        float vr[APPLY ? 3 : 1][KS][8];
        auto load_v = [&](int item) {
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int r2 = t2 & 15, g2 = (t2 & 63) >> 4;
            const int b = item / a.tiles, tile = item - b * a.tiles;
            const int ty0 = (tile / a.tiles_x) * FB_TH, tx0 = (tile % a.tiles_x) * FB_TW;
            const float* V = a.V + (long)b * a.v_bs;
            const long plane = (long)a.H * a.W;
            if (XVCL || a.v_tm) {
#pragma unroll
                for (int j = 0; j < (APPLY ? 3 : 0); ++j) {
                    const int p = 16 * (wave + 8 * j) + r2, ph = p / FB_HC, pc = p - ph * FB_HC;
                    const int gy = min(max(ty0 - 1 + ph, 0), a.H - 1), gx = min(max(tx0 - 1 + pc, 0), a.W - 1);
                    const float* vp = V + (unsigned)((((gy >> 3) * a.tiles_x + (gx >> 5)) * 256 + (gy & 7) * FB_TW + (gx & 31)) * a.C);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int c0 = 32 * ks + 8 * min(g2, max((a.C - 32 * ks - 8) / 8, 0));
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(vp + c0), hi4 = *reinterpret_cast<const f32x4*>(vp + c0 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vr[j][ks][e] = lo4[e]; vr[j][ks][4 + e] = hi4[e]; }
                    }
                }
                return;
            }
            if constexpr (!XVCL)
#pragma unroll
            for (int j = 0; j < (APPLY ? 3 : 0); ++j) {
                const int p = 16 * (wave + 8 * j) + r2, ph = p / FB_HC, pc = p - ph * FB_HC;
                const int gy = min(max(ty0 - 1 + ph, 0), a.H - 1), gx = min(max(tx0 - 1 + pc, 0), a.W - 1);
                const unsigned pix = (unsigned)(gy * a.W + gx);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned off = pix + (unsigned)(8 * min(g2, max((a.C - 32 * ks - 8) / 8, 0)) * plane);
#pragma unroll
                    for (int e = 0; e < 8; ++e) vr[j][ks][e] = fb_gld((V + (long)(32 * ks + e) * plane) + off);
                }
            }
        };
        if (round == 0) load_x(item);
        const int b = item / a.tiles, tile = item - b * a.tiles;
        const int ty0 = (tile / a.tiles_x) * FB_TH, tx0 = (tile % a.tiles_x) * FB_TW;
        float* Y = a.Y + (long)b * a.y_bs;
        const int nitem = item_of(round + 1);

        FB_T(0);
        // the previous item's last barrier has passed: every LDS region is free
        fb_dma<RECP>(a.rec, slots + RECF, wave, lane);                  // record 0 (prologue GEMM) -> slot 1
        fb_dma<RECP>(a.rec + RECF, slots, wave, lane);                  // record 1 (iteration 0)   -> slot 0
        if constexpr (GATE) fb_dma<W2P>(a.w2, w2a, wave, lane);

        // Where lane (r, g) parks the hidden channels of its halo pixels (byte offset of (pixel, channel quad g) in LDS
        // image 0).  The reference zero-pads h, not x: halo pixels OUTSIDE the image must read as zero in the stencil.
        // Their image slots are zeroed once per item (below, behind the barrier that frees the image area) and the
        // lane's stores of every stage go to the junk pixels behind the image instead (like the lanes without a
        // pixel) - no select per value in the GEMM epilogue.
        unsigned vq[3];
        bool inside[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 16 * (wave + 8 * j) + r;
            const int gy = ty0 - 1 + hr[j], gx = tx0 - 1 + hc[j];
            inside[j] = pv[j] && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            vq[j] = fb_opaque((unsigned)(PL_OFF + ((inside[j] ? p : FB_NP + r) * FB_PS + 4 * g) * 4));
        }

        if constexpr (APPLY) {
            // ---------------------------------------------------- x' = x + bias_o + Mfold[b] v  (restormer.py:131, 147)
            constexpr int MFP = 2 * KS * KS * 2;       // 1 KiB pieces: [2 KS tiles][KS][hi|lo]
            static_assert(MFP * 1024 <= 2 * PL_B, "the folded matrix must fit the image area");
            fb_dma<MFP>(a.mf + (long)b * a.mf_bs, smem + PL_OFF / 4, wave, lane);
            load_v(item);
            const int nt = a.C / 16;
            // accumulators = the x registers, in the operand scale of v (2^-4: exact)
#pragma unroll
            for (int t = 0; t < 2 * KS; ++t) {
                f32x4 bo = {0.f, 0.f, 0.f, 0.f};
                if (a.bias_o && t < nt) bo = *reinterpret_cast<const f32x4*>(a.bias_o + 16 * t + 4 * g);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        xr[j][t >> 1][4 * (t & 1) + e] = t < nt ? (xr[j][t >> 1][4 * (t & 1) + e] + bo[e]) * 0.0625f : 0.f;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            FB_T(8);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                fb_h8 vh[3], vl[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float sv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) sv[e] = irm_sat_h(vr[j][ks][e] * 0.0625f);
                    irm_split8(sv, vh[j], vl[j]);
                }
#pragma unroll
                for (int t = 0; t < 2 * KS; ++t) {
                    if (t < nt) {
                        const fb_h8 ah = fb_ld<fb_h8>(lds, vw, PL_OFF + ((t * KS + ks) * 2) * 1024);
                        const fb_h8 al = fb_ld<fb_h8>(lds, vw, PL_OFF + ((t * KS + ks) * 2 + 1) * 1024);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            f32x4 acc = {xr[j][t >> 1][4 * (t & 1)], xr[j][t >> 1][4 * (t & 1) + 1],
                                         xr[j][t >> 1][4 * (t & 1) + 2], xr[j][t >> 1][4 * (t & 1) + 3]};
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vh[j], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vl[j], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vh[j], acc, 0, 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) xr[j][t >> 1][4 * (t & 1) + e] = acc[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[j][ks][e] *= 16.0f;
            // every wave is done with the matrix before the residual transpose overwrites the image area
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            FB_T(9);
        }
        constexpr int RP = 32 * KS + 4;                // YCL: floats per pixel of the residual image (16-byte aligned; the 16
        static_assert(!YCL || 256 * RP * 4 <= 2 * PL_B, "residual image");        // pixels of a lane group on 16 bank quads)
        if constexpr (YCL) {
            // x' of the 256 interior pixels pixel-major [pixel][channel] into the free image area: the swapped project_out
            // wants the residual as (pixel on the lane, channels 16 c + 4 g + e) - the order the registers already have,
            // but for the pixels of the lane's STENCIL position
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ip = (hr[j] - 1) * FB_TW + hc[j] - 1;
                if (pv[j] && hr[j] >= 1 && hr[j] <= FB_TH && hc[j] >= 1 && hc[j] <= FB_TW) {
                    const unsigned vt = (unsigned)(PL_OFF + (ip * RP + 4 * g) * 4);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (32 * ks < a.C)
                            fb_st<f32x4>(lds, vt, 32 * ks * 4, (f32x4){xr[j][ks][0], xr[j][ks][1], xr[j][ks][2], xr[j][ks][3]});
                        if (32 * ks + 16 < a.C)
                            fb_st<f32x4>(lds, vt, (32 * ks + 16) * 4, (f32x4){xr[j][ks][4], xr[j][ks][5], xr[j][ks][6], xr[j][ks][7]});
                    }
                }
            }
        } else if constexpr (GATE) {
            // The residual is the tile's own input, already in registers (xr: pixel on the lane, channels in the
            // registers); project_out's accumulators want it transposed (channel on the lane, 4 pixels in the registers).
            // The image area is still free: park the 256 interior pixels channel-major there (row stride RT floats = 66
            // bank slots: the 16 rows a ds_read_b128 lane group touches fall on 16 different slots) and read them back
            // as accumulator tiles behind one barrier - no second global read of x (PMC reads 2.05x).
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ip = (hr[j] - 1) * FB_TW + hc[j] - 1;
                if (pv[j] && hr[j] >= 1 && hr[j] <= FB_TH && hc[j] >= 1 && hc[j] <= FB_TW) {
                    const unsigned vt = (unsigned)(PL_OFF + ((APPLY ? 4 : 8) * g * RT + ip) * 4);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const unsigned vtk = vt + (unsigned)(32 * ks * RT * 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e)      // (APPLY: register e holds channel 32 ks + 16 (e >> 2) + 4 g + (e & 3))
                            fb_st<float>(lds, vtk, (APPLY ? 16 * (e >> 2) + (e & 3) : e) * RT * 4, xr[j][ks][e]);
                    }
                }
            }
        }
        // ------------------------------------------------------------ resident input: LayerNorm + fp16 split
        fb_h8 xh[3][KS], xl[3][KS];
        float osc[3];                              // per pixel: 1 / (operand scales) of the project_in accumulators
        // FULL: C == 32 KS (no channel masks), WBK: -1 = flavour read at run time, 0 = BiasFree, 1 = WithBias.  The common
        // shapes (C = 96 / 64 / 32) take copies without the ~6 selects per value of the general one (round 3: 830 -> 400
        // vector instructions per item and wave in this phase).
        auto ln_phase = [&](auto FULL_, auto WBK_) {
            constexpr bool FULL = decltype(FULL_)::value;
            constexpr int WBK = decltype(WBK_)::value;
            const float invC = 1.0f / (float)a.C;
            const bool wb = WBK < 0 ? a.ln_mode == IRM_LN_WITHBIAS : WBK == 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float v[KS][8];
                float s = 0.f;
                int kl[KS];                            // opaque here: otherwise the 24 lane masks are hoisted out of the
#pragma unroll                                         // item loop into scalar registers, which then spill
                for (int ks = 0; ks < KS; ++ks) { kl[ks] = FULL ? 8 : klim[ks]; if (!FULL) asm volatile("" : "+v"(kl[ks])); }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[ks][e] = xr[j][ks][e]; s += (FULL || e < kl[ks]) ? v[ks][e] : 0.f; }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s * invC;
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[ks][e] - mean;
                        q += (FULL || e < kl[ks]) ? d * d : 0.f;
                        v[ks][e] = (FULL || e < kl[ks]) ? (wb ? d : v[ks][e]) : 0.f;
                    }
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                // The fp16 operand is the input normalised to UNIT RMS (x 2^4), whatever eps does to the LayerNorm
                // output: where var << eps (near-constant or tiny inputs) LN(x) is tiny and its lo part would fall
                // into the fp16 subnormals.  The remaining factor rms / sqrt(var + eps) <= 1 is applied in fp32 to the
                // accumulator of this pixel (osc[j]).
                const float var = q * invC;
                const float ms = wb ? var : fmaf(mean, mean, var);         // mean square of the operand (d or x)
                const float rs = ms > 0.f ? 16.0f / sqrtf(ms) : 0.f;
                osc[j] = a.inv_s1 * sqrtf(ms / (var + a.eps));
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        // the product must be ONE rounded fp32 value for both parts (irm_split2: opaque register operands)
                        unsigned hh, ll;
                        irm_split2(__fmul_rn(v[ks][e], rs), __fmul_rn(v[ks][e + 1], rs), hh, ll);
                        reinterpret_cast<unsigned*>(&xh[j][ks])[e / 2] = hh;
                        reinterpret_cast<unsigned*>(&xl[j][ks])[e / 2] = ll;
                    }
            }
        };
        {
            const std::true_type T_; const std::false_type F_;
            if (a.C == 32 * KS) {
                if (a.ln_mode == IRM_LN_WITHBIAS) ln_phase(T_, std::integral_constant<int, 1>{});
                else ln_phase(T_, std::integral_constant<int, 0>{});
            } else {
                ln_phase(F_, std::integral_constant<int, -1>{});
            }
        }

        FB_T(1);
        __builtin_amdgcn_sched_barrier(0);
        // project_out accumulators start from (residual + bias) in their own scale (s2 / 16, a power of two: exact).
        // One of the 2 CT tiles lives in LDS between the project_out phases (PARK_OFF, lane-linear): the stencil
        // phase is 4 registers over the 256 the two waves per SIMD allow, and hipcc's own choice is to spill an
        // accumulator tile to scratch with its store right in front of the iteration-end vmcnt(0).
        f32x4 acc2[2][CT];
        const unsigned vpark = fb_opaque((unsigned)(PARK_OFF + threadIdx.x * 16));
        if constexpr (GATE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float rsc = 1.0f / a.inv_s2;
            if constexpr (YCL) {
                const unsigned vt = (unsigned)(PL_OFF + (((2 * (wave >> 1)) * FB_TW + 16 * (wave & 1) + r) * RP + 4 * g) * 4);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias2 && 16 * c < a.C) bv = *reinterpret_cast<const f32x4*>(a.bias2 + 16 * c + 4 * g);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                        if (16 * c < a.C) rr = fb_ld<f32x4>(lds, vt, (q * FB_TW * RP + 16 * c) * 4);
                        acc2[q][c] = (f32x4){(rr[0] + bv[0]) * rsc, (rr[1] + bv[1]) * rsc, (rr[2] + bv[2]) * rsc, (rr[3] + bv[3]) * rsc};
                    }
                }
            } else {
            const unsigned vt = (unsigned)(PL_OFF + (r * RT + 64 * (wave >> 1) + 16 * (wave & 1) + 4 * g) * 4);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = 16 * c + r;
                const float bv = (a.bias2 && co < a.C) ? a.bias2[co] : 0.0f;
                const unsigned vtc = vt + (unsigned)(16 * c * RT * 4);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 rr = fb_ld<f32x4>(lds, vtc, q * FB_TW * 4);
                    acc2[q][c] = (f32x4){(rr[0] + bv) * rsc, (rr[1] + bv) * rsc, (rr[2] + bv) * rsc, (rr[3] + bv) * rsc};
                }
            }
            }
            fb_st<f32x4>(lds, vpark, 0, acc2[1][CT - 1]);
        }

        // One unit of the project_in GEMM of a stage = (16-channel tile hct, k-step ks): the weights are the A operand,
        // so lane (r, g) receives hidden channels 4 g .. 4 g + 3 of pixel r; after the last k-step the tile goes to
        // the LDS image img (+ bias, zero outside the image).  slot / img are compile-time constants at every call.
        f32x4 acc1[3];
        auto g1_comp = [&](int hct, int ks, int img, const fb_h8& ah, const fb_h8& al, const f32x4& b1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ks == 0) acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[j][ks], acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[j][ks], acc1[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[j][ks], acc1[j], 0, 0, 0);
            }
            if (ks == KS - 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaf(acc1[j][e], osc[j], b1[e]);
                    fb_st<f32x4>(lds, vq[j], img * PL_B + hct * 64, h);
                }
            }
        };
        auto gemm1_unit = [&](int hct, int ks, int slot, int img) {
            const fb_h8 ah = fb_ld<fb_h8>(lds, vw, slot * SLOT_B + ((hct * KS + ks) * 2) * 1024);
            const fb_h8 al = fb_ld<fb_h8>(lds, vw, slot * SLOT_B + ((hct * KS + ks) * 2 + 1) * 1024);
            const f32x4 b1 = fb_ld<f32x4>(lds, vc, slot * SLOT_B + CF_OFF + 320 * 4 + hct * 64);
            g1_comp(hct, ks, img, ah, al, b1);
        };

        // One chunk of the depth-wise stencil = (channel half hf, halo row dy): lane (r, g) of wave w -> output rows
        // 2 (w >> 1) + q, column 16 (w & 1) + r, channels 4 g + i of the half; row dy feeds tap row dy of q = 0 and
        // tap row dy - 1 of q = 1.
        float o[2][2][4];                              // [half][row q][channel]
        f32x4 kprev[3];
        auto st_comp = [&](int hf, int dy, const f32x4 (&P)[3], const f32x4 (&kc)[3], const f32x4& kb) {
            // scalar v_fma_f32 (the file is built with -fno-slp-vectorize): as many issue slots as packed FMAs, and
            // they pair with the MFMAs of the interleaved GEMM unit
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
            // pin the partial sums here: otherwise the compiler sinks the whole FMA chain to its consumer (the next
            // iteration's project_out) and keeps every LDS read of the stage alive until then
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(o[hf][q][e]));
        };

        FB_T(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FB_T(3);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (pv[j] && !inside[j]) {                 // border tiles only: out-of-image halo pixels read as zero
                const unsigned vz = (unsigned)(PL_OFF + ((16 * (wave + 8 * j) + r) * FB_PS + 4 * g) * 4);
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                fb_st<f32x4>(lds, vz, 0, z4);
                fb_st<f32x4>(lds, vz, 64, z4);
                fb_st<f32x4>(lds, vz, PL_B, z4);
                fb_st<f32x4>(lds, vz, PL_B + 64, z4);
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * KS; ++u) gemm1_unit(u / KS, u % KS, 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FB_T(4);

        fb_u4 Gh[2], Gl[2];                            // packed fp16 pairs: the k-slots of lane (r, g) in project_out's MFMA
#pragma unroll
        for (int q = 0; q < 2; ++q) { Gh[q] = (fb_u4){0u, 0u, 0u, 0u}; Gl[q] = (fb_u4){0u, 0u, 0u, 0u}; }

        float oprev[2][2][4];
        auto store_stage = [&](int st) {               // !GATE: stage st of the depth-wise outputs (oprev) -> Y
            int t4 = threadIdx.x;
            asm volatile("" : "+v"(t4));
            const int g = (t4 & 63) >> 4, sy = ty0 + 2 * (wave >> 1), sx = tx0 + 16 * (wave & 1) + (t4 & 15);
            if (a.tm && 32 * st < 2 * a.C) {
                // q, k tile-major (irm_qkv_dw_fused_tm_f16x3_f32: full tiles, 2C % 32 == 0): the tile's 2C x 256 floats are one
                // contiguous block, read back by irm_mdta_gram_tm_f16x3_f32 only
                float* yt = Y + (long)tile * (2L * a.C * 256) + (long)(32 * st) * 256;
                const unsigned tvoff = (unsigned)(4 * g * 256 + (2 * (wave >> 1)) * FB_TW + 16 * (wave & 1) + (t4 & 15));
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int q = 0; q < 2; ++q) yt[tvoff + (16 * hf + e) * 256 + q * FB_TW] = oprev[hf][q][e];
                return;
            }
            if (a.v_tm && 32 * st >= 2 * a.C) {
                // v tile-major channel-last inside the v part of Y (channels [2C, 3C): C N floats per image): the lane's 4
                // channels of a pixel are one 16-byte store
                float* yt = Y + 2L * a.C * plane + (long)tile * (a.C * 256L) + (32 * st - 2 * a.C) + 4 * g;
                const unsigned px = (unsigned)((2 * (wave >> 1)) * FB_TW + 16 * (wave & 1) + (t4 & 15));
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    if (32 * st + 16 * hf + 4 * g + 3 < a.M) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            *reinterpret_cast<f32x4*>(yt + (px + q * FB_TW) * a.C + 16 * hf) =
                                (f32x4){oprev[hf][q][0], oprev[hf][q][1], oprev[hf][q][2], oprev[hf][q][3]};
                    }
                return;
            }
            const unsigned svoff = (unsigned)((4 * g) * plane + (long)sy * a.W + sx);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* yu = Y + (long)(32 * st + 16 * hf + e) * plane;          // wave-uniform part
                    const bool cok = 32 * st + 16 * hf + 4 * g + e < a.M && sx < a.W;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (cok && sy + q < a.H) yu[svoff + q * a.W] = oprev[hf][q][e];
                }
        };

        // iteration it: stencil of stage it (LDS image it & 1) interleaved with the GEMM of stage it + 1 (into the other
        // image), chunk by chunk; project_out every second stage.  The last iteration (more = false) has no GEMM: its
        // operand registers take the raw input of the next item.
        auto iter = [&](auto PAR, auto MORE, int it) {
            constexpr int par = decltype(PAR)::value;
            constexpr bool more = decltype(MORE)::value;
            if constexpr (!GATE) { if (it > 0) store_stage(it - 1); }
            if constexpr (more) fb_dma<RECP>(a.rec + (long)(it + 2) * RECF, slots + (par ^ 1) * RECF, wave, lane);
            if (GATE && par == 0 && it > 0) fb_dma<W2P>(a.w2 + (long)(it >> 1) * W2F, w2a, wave, lane);
            if constexpr (!more && (par == 1 || !GATE)) {
                // unconditional (the last round re-reads a valid item for nothing): under "if (there is a next item)"
                // the old values would have to stay alive through every iteration of this item, 72 registers.
                // (GATE with an odd stage count ends on the par == 0 instance, which also holds the first half of the
                // project_out operands: there the request waits until the iteration is over.  The qkv branch has no
                // such operands: its last iteration requests the input whatever its parity - round 3; before, the odd
                // stage counts of C = 96 (9 stages) and C = 48 (5) requested it after the last stencil.)
                load_x(min(nitem, a.items - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const unsigned vp = par ? vp1 : vp0;
                f32x4 P[2][3], K[2][3], KB[1], B1[KS == 1 ? 2 : 1];   // (tap / GEMM bias: not in consecutive chunks unless KS == 1)
                fb_h8 AH[2], AL[2];
                // chunk I: stencil (half I >> 2, halo row I & 3) + GEMM unit (tile I / KS, k-step I % KS)
                auto loads = [&](auto IC, auto NC) {
                    constexpr int I = decltype(IC)::value, n = decltype(NC)::value;
                    constexpr int hf = I >> 2, dy = I & 3;
                    constexpr int cf = par * SLOT_B + CF_OFF + hf * 64;
#ifdef FB_ABLATE_TAPS      // timing-only ablation (wrong results): what the 20 tap reads per stage cost
                    if constexpr (dy == 0) asm volatile("" : "=v"(KB[0]));
                    if constexpr (dy < 3) { asm volatile("" : "=v"(K[n][0])); asm volatile("" : "=v"(K[n][1])); asm volatile("" : "=v"(K[n][2])); }
#else
                    if constexpr (dy == 0) fb_dsr<cf + 9 * 128>(KB[0], vc);
                    if constexpr (dy < 3) {
                        fb_dsr<cf + (dy * 3 + 0) * 128>(K[n][0], vc);
                        fb_dsr<cf + (dy * 3 + 1) * 128>(K[n][1], vc);
                        fb_dsr<cf + (dy * 3 + 2) * 128>(K[n][2], vc);
                    }
#endif
                    fb_dsr<(dy * FB_HC + 0) * (FB_PS * 4) + hf * 64>(P[n][0], vp);
                    fb_dsr<(dy * FB_HC + 1) * (FB_PS * 4) + hf * 64>(P[n][1], vp);
                    fb_dsr<(dy * FB_HC + 2) * (FB_PS * 4) + hf * 64>(P[n][2], vp);
                    if constexpr (more && I < 2 * KS) {
                        constexpr int hct = I / KS, ks = I % KS;
                        fb_dsr<par * SLOT_B + ((hct * KS + ks) * 2) * 1024>(AH[n], vw);
                        fb_dsr<par * SLOT_B + ((hct * KS + ks) * 2 + 1) * 1024>(AL[n], vw);
                        if constexpr (ks == KS - 1) fb_dsr<par * SLOT_B + CF_OFF + 320 * 4 + hct * 64>(B1[KS == 1 ? n : 0], vc);
                    }
                };
                // (the last iteration carries the next item's raw input in registers instead of the second operand
                // buffer: its chunks load and wait in place)
                if constexpr (more) loads(fb_ic<0>{}, fb_ic<0>{});
                fb_for<8>([&](auto IC) {
                    constexpr int I = decltype(IC)::value, c = more ? (I & 1) : 0, n = c ^ 1;
                    constexpr int hf = I >> 2, dy = I & 3;
                    if constexpr (more && I + 1 < 8) loads(fb_ic<I + 1>{}, fb_ic<n>{});
                    if constexpr (!more) loads(fb_ic<I>{}, fb_ic<0>{});
                    // LDS operations issued after the loads of chunk I: the loads of chunk I + 1
                    constexpr int J = I + 1, jdy = J & 3;
#ifdef FB_ABLATE_TAPS
                    constexpr int nnext = (more && J < 8) ? 3 +
#else
                    constexpr int nnext = (more && J < 8) ? 3 + (jdy < 3 ? 3 : 0) + (jdy == 0 ? 1 : 0) +
#endif
                                                      ((more && J < 2 * KS) ? 2 + (J % KS == KS - 1 ? 1 : 0) : 0) : 0;
                    fb_waitcnt<nnext>();
                    fb_tie(P[c][0], P[c][1], P[c][2]);
                    if constexpr (dy < 3) fb_tie(K[c][0], K[c][1], K[c][2]);
                    if constexpr (dy == 0) fb_tie(KB[0]);
                    st_comp(hf, dy, P[c], K[c], KB[0]);
                    if constexpr (more && I < 2 * KS) {
                        constexpr int hct = I / KS, ks = I % KS;
                        fb_tie(AH[c], AL[c]);
                        if constexpr (ks == KS - 1) fb_tie(B1[KS == 1 ? c : 0]);
                        g1_comp(hct, ks, par ^ 1, AH[c], AL[c], B1[KS == 1 ? c : 0]);
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, FB_INTERLEAVE, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            if constexpr (!GATE) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) oprev[hf][q][e] = o[hf][q][e];
            }
#pragma unroll
            for (int q = 0; q < (GATE ? 2 : 0); ++q)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // (the taps and the bias of the second half are pre-scaled by 2^-4 on the host: o[1] = dw(h2) / 16, exact)
                    const float g0 = irm_sat_h(__fmul_rn(fb_gelu1(o[0][q][e]), o[1][q][e]));
                    const float g1 = irm_sat_h(__fmul_rn(fb_gelu1(o[0][q][e + 1]), o[1][q][e + 1]));
                    unsigned hh, ll;
                    irm_split2(g0, g1, hh, ll);
                    Gh[q][2 * par + e / 2] = hh;
                    Gl[q][2 * par + e / 2] = ll;
                }
#pragma unroll
            for (int q = 0; q < (GATE ? 2 : 0); ++q) asm volatile("" : "+v"(Gh[q]), "+v"(Gl[q]));
            if (GATE && (par == 1 || !more)) {
                if constexpr (par == 0) {
                    // odd stage count: this super-stage's project_out weights were requested in this very iteration
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                f32x4 accp = fb_ld<f32x4>(lds, vpark, 0);      // the parked tile (q = 1, c = CT - 1)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const fb_h8 bh = fb_ld<fb_h8>(lds, vw, W2_OFF + (c * 2) * 1024);
                    const fb_h8 bl = fb_ld<fb_h8>(lds, vw, W2_OFF + (c * 2 + 1) * 1024);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4& t = (q == 1 && c == CT - 1) ? accp : acc2[q][c];
                        if constexpr (YCL) {        // weights as the A operand: the lane receives 4 channels of its pixel
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, __builtin_bit_cast(fb_h8, Gl[q]), t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, __builtin_bit_cast(fb_h8, Gh[q]), t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, __builtin_bit_cast(fb_h8, Gh[q]), t, 0, 0, 0);
                        } else {
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fb_h8, Gl[q]), bh, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fb_h8, Gh[q]), bl, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fb_h8, Gh[q]), bh, t, 0, 0, 0);
                        }
                    }
                }
                fb_st<f32x4>(lds, vpark, 0, accp);
#pragma unroll
                for (int q = 0; q < 2; ++q) { Gh[q] = (fb_u4){0u, 0u, 0u, 0u}; Gl[q] = (fb_u4){0u, 0u, 0u, 0u}; }
            }
            // the weight DMAs of this iteration must have landed; the last iteration issued none (only the next
            // item's input, which must NOT be waited for here)
            if constexpr (more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        {
            const std::integral_constant<int, 0> P0; const std::integral_constant<int, 1> P1;
            const std::true_type T; const std::false_type F;
            int it = 0;
            for (; it + 2 < S; it += 2) { iter(P0, T, it); iter(P1, T, it + 1); }
            if (it + 2 == S) { iter(P0, T, it); FB_T(5); iter(P1, F, it + 1); }
            else { FB_T(5); iter(P0, F, it); if constexpr (GATE) load_x(min(nitem, a.items - 1)); }
            FB_T(6);
        }

        if constexpr (!GATE) {
            store_stage(S - 1);
        } else {
            // -------------------------------------------------------- epilogue: 16-byte stores
            // (lane geometry from a fresh opaque lane id: kept from the prologue it would sit in registers - or in
            // scratch - through all the iterations)
            int t3 = threadIdx.x;
            asm volatile("" : "+v"(t3));
            const int r = t3 & 15, ox = tx0 + 16 * (wave & 1) + 4 * ((t3 & 63) >> 4), oy0 = ty0 + 2 * (wave >> 1);
            acc2[1][CT - 1] = fb_ld<f32x4>(lds, vpark, 0);
            if constexpr (YCL) {
                // lane (pixel r, g) holds channels 16 c + 4 g + e of its two pixels: 16-byte stores, channel-last
                float* yt = Y + (long)tile * (a.C * 256L) + 4 * ((t3 & 63) >> 4);
                const unsigned px = (unsigned)((2 * (wave >> 1)) * FB_TW + 16 * (wave & 1) + r);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (16 * c >= a.C) continue;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        *reinterpret_cast<f32x4*>(yt + (px + q * FB_TW) * a.C + 16 * c) =
                            (f32x4){acc2[q][c][0] * a.inv_s2, acc2[q][c][1] * a.inv_s2, acc2[q][c][2] * a.inv_s2, acc2[q][c][3] * a.inv_s2};
                }
            } else
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int co = 16 * c + r;
                if (co >= a.C || ox >= a.W) continue;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (oy0 + q >= a.H) continue;
                    const float4 v = make_float4(acc2[q][c][0] * a.inv_s2, acc2[q][c][1] * a.inv_s2,
                                                 acc2[q][c][2] * a.inv_s2, acc2[q][c][3] * a.inv_s2);
                    *reinterpret_cast<float4*>(Y + (long)co * plane + (long)(oy0 + q) * a.W + ox) = v;
                }
            }
        }
        FB_T(7);
        if (nitem >= a.items) break;
        item = nitem;
        ++round;
    }
#ifdef FB_STAMP
    if (threadIdx.x == 0 && a.dbg)
        for (int i = 0; i < 10; ++i) a.dbg[blockIdx.x * 10 + i] = stamp[i];
#endif
}

template <int KS, int CT, bool GATE = true, bool APPLY = false, bool YCL = false, bool XVCL = false>
static int gdfn_launch(FusedArgs a, int B, hipStream_t stream) {
    const size_t lds = ((size_t)2 * FB_PLF + 2 * (KS * 1024 + 512) + (GATE ? CT * 512 + 2048 : 0)) * sizeof(float);
    static_assert(((size_t)2 * FB_PLF + 2 * (KS * 1024 + 512) + CT * 512 + 2048) * sizeof(float) <= 160 * 1024, "LDS");
    static int configured_dev[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return IRM_ELAUNCH;
    if (!configured_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lnpw_dw_fused_kernel<KS, CT, GATE, APPLY, YCL, XVCL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return IRM_ELAUNCH;
        configured_dev[dev] = 1;
    }
    a.tiles_x = (a.W + FB_TW - 1) / FB_TW;
    a.tiles = a.tiles_x * ((a.H + FB_TH - 1) / FB_TH);
    a.items = B * a.tiles;
    static int cus_dev[64] = {0};
    if (!cus_dev[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return IRM_ELAUNCH;
        cus_dev[dev] = n;
    }
    // one workgroup per CU (the kernel holds ~150 KiB of LDS), in whole groups of 8 (one per XCD)
    const int per = (a.items + 7) >> 3;
    a.gpx = (cus_dev[dev] + 7) / 8;
    if (a.gpx > per) a.gpx = per;
    hipLaunchKernelGGL((lnpw_dw_fused_kernel<KS, CT, GATE, APPLY, YCL, XVCL>), dim3(a.gpx * 8), dim3(512), lds, stream, a);
    return irm_launch_status();
}

extern "C" int irm_gdfn_fused_f16x3_f32(const float* rec, const float* w2, const float* bias2, const float* x, long x_bs,
                                        float* y, long y_bs, int ln_mode, float eps, float inv_s1, float inv_s2, int B,
                                        int C, int hid, int H, int W, hipStream_t stream) {
    if (!rec || !w2 || !x || !y || x == y || B <= 0 || C <= 0 || hid <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (C > 96 || (W & 3)) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(rec) || !irm_aligned16(w2))
        return IRM_EINVAL;
    FusedArgs a;
    a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.rec = rec; a.w2 = w2; a.bias2 = bias2;
    a.C = C; a.H = H; a.W = W; a.S = (hid + 15) / 16; a.M = 0; a.ln_mode = ln_mode; a.eps = eps; a.inv_s1 = inv_s1; a.inv_s2 = inv_s2;
    a.V = nullptr; a.v_bs = 0; a.mf = nullptr; a.mf_bs = 0; a.bias_o = nullptr; a.tm = 0; a.x_tm = 0; a.v_tm = 0; a.y_tm = 0;
#ifdef FB_STAMP
    a.dbg = getenv("FB_DBG_PTR") ? (unsigned long long*)strtoull(getenv("FB_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    a.tiles_x = 0; a.tiles = 0;
    const int ks = (C + 31) / 32, ct = (C + 15) / 16;
    if (ks == 3) {
        if (ct <= 6) return gdfn_launch<3, 6>(a, B, stream);
    } else if (ks == 2) {
        if (ct <= 3) return gdfn_launch<2, 3>(a, B, stream);
        return gdfn_launch<2, 4>(a, B, stream);
    } else if (ks == 1) {
        return gdfn_launch<1, 2>(a, B, stream);
    }
    return IRM_EINVAL;
}

// The attention branch's last step + the GDFN branch: y = x' + GDFN(x'), x' = x + bias_o + Mfold[b] v (header).
extern "C" int irm_attn_gdfn_fused_f16x3_f32(const float* rec, const float* w2, const float* bias2, const float* x, long x_bs,
                                             const float* v, long v_bs, const float* mfold_frag, const float* bias_o,
                                             float* y, long y_bs, int ln_mode, float eps, float inv_s1, float inv_s2, int B,
                                             int C, int hid, int H, int W, int lay, hipStream_t stream) {
    if (lay < 0 || lay > 7 || (lay && ((H & 7) || (W & 31)))) return IRM_EINVAL;
    if (!rec || !w2 || !x || !v || !mfold_frag || !y || x == y || v == y || B <= 0 || C <= 0 || hid <= 0 || H <= 0 || W <= 0)
        return IRM_EINVAL;
    if (C > 96 || (C & 15) || (W & 3)) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || (v_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(v) ||
        !irm_aligned16(rec) || !irm_aligned16(w2) || !irm_aligned16(mfold_frag) || (bias_o && !irm_aligned16(bias_o)))
        return IRM_EINVAL;
    FusedArgs a;
    a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.rec = rec; a.w2 = w2; a.bias2 = bias2;
    a.C = C; a.H = H; a.W = W; a.S = (hid + 15) / 16; a.M = 0; a.ln_mode = ln_mode; a.eps = eps; a.inv_s1 = inv_s1; a.inv_s2 = inv_s2;
    const int ks = (C + 31) / 32, ct = (C + 15) / 16;
    a.V = v; a.v_bs = v_bs; a.mf = mfold_frag; a.mf_bs = (long)2 * ks * ks * 512; a.bias_o = bias_o; a.tm = 0; a.x_tm = lay & 1; a.v_tm = (lay >> 1) & 1; a.y_tm = (lay >> 2) & 1;
#ifdef FB_STAMP
    a.dbg = getenv("FB_DBG_PTR") ? (unsigned long long*)strtoull(getenv("FB_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    a.tiles_x = 0; a.tiles = 0;
#ifndef FB_NO_XVCL          // (variant build for A/B: the general instantiations with run-time layout flags only)
    if (a.x_tm && a.v_tm && ks == 3) return a.y_tm ? ((bias2 && !irm_aligned16(bias2)) ? IRM_EINVAL : gdfn_launch<3, 6, true, true, true, true>(a, B, stream))
                                                   : gdfn_launch<3, 6, true, true, false, true>(a, B, stream);
    if (a.x_tm && a.v_tm && ks == 2 && ct <= 3) return a.y_tm ? ((bias2 && !irm_aligned16(bias2)) ? IRM_EINVAL : gdfn_launch<2, 3, true, true, true, true>(a, B, stream))
                                                              : gdfn_launch<2, 3, true, true, false, true>(a, B, stream);
#endif
    if (a.y_tm) {
        if (bias2 && !irm_aligned16(bias2)) return IRM_EINVAL;
        if (ks == 3) return gdfn_launch<3, 6, true, true, true>(a, B, stream);
        if (ks == 2) return ct <= 3 ? gdfn_launch<2, 3, true, true, true>(a, B, stream) : gdfn_launch<2, 4, true, true, true>(a, B, stream);
        if (ks == 1) return gdfn_launch<1, 2, true, true, true>(a, B, stream);
        return IRM_EINVAL;
    }
    if (ks == 3) return gdfn_launch<3, 6, true, true>(a, B, stream);
    if (ks == 2) return ct <= 3 ? gdfn_launch<2, 3, true, true>(a, B, stream) : gdfn_launch<2, 4, true, true>(a, B, stream);
    if (ks == 1) return gdfn_launch<1, 2, true, true>(a, B, stream);
    return IRM_EINVAL;
}

static int qkv_dw_fused(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode,
                        float eps, float inv_s1, int B, int C, int M, int H, int W, int tm, hipStream_t stream) {
    if (!rec || !x || !y || x == y || B <= 0 || C <= 0 || M <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (C > 96 || (W & 3) || (long)M * H * W >= (1L << 30)) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(rec)) return IRM_EINVAL;
    FusedArgs a;
    a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.rec = rec; a.w2 = nullptr; a.bias2 = nullptr;
    a.C = C; a.H = H; a.W = W; a.S = (M + 31) / 32; a.M = M; a.ln_mode = ln_mode; a.eps = eps; a.inv_s1 = inv_s1; a.inv_s2 = 0.f;
    a.V = nullptr; a.v_bs = 0; a.mf = nullptr; a.mf_bs = 0; a.bias_o = nullptr; a.tm = tm & 1; a.x_tm = (tm >> 1) & 1; a.v_tm = (tm >> 2) & 1; a.y_tm = 0;
#ifdef FB_STAMP
    a.dbg = getenv("FB_DBG_PTR") ? (unsigned long long*)strtoull(getenv("FB_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    a.tiles_x = 0; a.tiles = 0;
#ifndef FB_NO_XVCL
    if (a.x_tm) {            // the blocks inside a stage: no planar load path in the instantiation
        if ((C + 31) / 32 == 3) return gdfn_launch<3, 1, false, false, false, true>(a, B, stream);
        if ((C + 31) / 32 == 2) return gdfn_launch<2, 1, false, false, false, true>(a, B, stream);
    }
#endif
    switch ((C + 31) / 32) {
        case 3: return gdfn_launch<3, 1, false>(a, B, stream);
        case 2: return gdfn_launch<2, 1, false>(a, B, stream);
        case 1: return gdfn_launch<1, 1, false>(a, B, stream);
    }
    return IRM_EINVAL;
}

extern "C" int irm_qkv_dw_fused_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode,
                                          float eps, float inv_s1, int B, int C, int M, int H, int W, hipStream_t stream) {
    return qkv_dw_fused(rec, x, x_bs, y, y_bs, ln_mode, eps, inv_s1, B, C, M, H, W, 0, stream);
}

// q, k (output channels [0, 2C)) tile-major inside y's q, k part, v planar as before (header).
extern "C" int irm_qkv_dw_fused_tm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode,
                                             float eps, float inv_s1, int B, int C, int H, int W, int x_tm, int v_tm,
                                             hipStream_t stream) {
    if (C <= 0 || (C & 15) || H <= 0 || W <= 0 || (H & 7) || (W & 31)) return IRM_EINVAL;
    return qkv_dw_fused(rec, x, x_bs, y, y_bs, ln_mode, eps, inv_s1, B, C, 3 * C, H, W, 1 | (x_tm ? 2 : 0) | (v_tm ? 4 : 0), stream);
}
