// The front half of MDTA for the C <= 96 levels with a CHANNEL-MAJOR hidden image (round 3, late):
//
//   q, k, v = dw3x3(W LN(x) + b)       (norm1 + Attention.qkv + qkv_dwconv, restormer.py:25-70, 105-106, 116)
//
// Same work items, operands and fp32 emulation as lnpw_dw_fused_kernel<!GATE> (fused_block.hip): an 8 x 32 tile with its
// 1-pixel halo, the normalised input split once into fp16 hi/lo and resident in registers, stages of 32 output channels.
// What differs is the layout of the 32-channel h image between the 1x1 conv and the depth-wise stencil.  The steady
// iteration of the branch kernels is a SUM of matrix, vector and LDS time, and with the PIXEL on the
// lane 44 of the 64 ds_read_b128 per wave and stage are the stencil's: every lane reads its 3 columns x 4 rows although its
// neighbours read two of the same columns, plus 20 broadcast reads of the taps.  Here
//   * the 1x1 conv runs with the pixels as the MFMA ROW index (the resident input is the A operand, the weights B): a lane
//     receives 4 consecutive pixels of ONE hidden channel and parks them with one ds_write_b128 in a channel-major image
//     [32 channels][10 rows x 36 pixel slots];
//   * the stencil has the CHANNEL on the lane: wave w owns output row w, lane (i, g) channel i of a 16-channel tile and the
//     8 pixels 8 g .. 8 g + 7: per halo row two ds_read_b128 and one ds_read_b64 give its 10-pixel segment (9 reads per tile
//     instead of 24 + taps 20), the 9 taps + bias of its channel are 10 scalars;
//   * the 8 results of a lane are 8 consecutive pixels of one channel: q, k go to their tile-major [tile][2C][256] block with
//     TWO 16-byte stores (before: 16 four-byte stores per stage), v to its channel-last block.
// Channel rows sit 1728 bytes apart plus 16 o(i) bytes, o = 0,1,4,5,8,9,12,13 for the rows {0-3, 12-15} and again for {4-11}
// of a 16-channel tile: the (non-contiguous) 16-lane groups of ds_read_b128 - rows {0-3, 12-15} at pixel group g with rows
// {4-11} at g + 1 - then touch 16 different bank quads.
// Whole tiles only (H % 8 == 0, W % 32 == 0), C % 16 == 0, M = 3 C; q, k tile-major, v channel-last or planar.
#include "irm_common.h"
#include <utility>

typedef _Float16 qc_h8 __attribute__((ext_vector_type(8)));
typedef float qc_v2 __attribute__((ext_vector_type(2)));

#define QC_TH 8
#define QC_TW 32
#define QC_PITCH 36                              // pixel slots per halo row (34 used)
#define QC_SLOTS ((QC_TH + 2) * QC_PITCH)        // 360
#define QC_PT 23                                 // 16-slot MFMA tiles (368 slots)
#define QC_CS 1728                               // bytes between channel rows
#define QC_IMG (32 * QC_CS)                      // bytes per image

struct QcArgs {
    const float* X; long x_bs;
    float* Y; long y_bs;                         // [B][3C][H][W]: q, k tile-major inside [0, 2C), v inside [2C, 3C)
    const float* rec;                            // records of irm_qkv_dw_fused_f16x3_f32 (include/irm_hip.h)
    int C, H, W, S, M;
    int ln_mode; float eps, inv_s1;
    int x_tm, v_tm;
    int tiles_x, tiles, items, gpx;
    // GRAM (C = 48, one head): q, k are not written; per chunk of QC_NCH consecutive tiles of an image the workgroup emits one
    // partial record of irm_mdta_gram_* (G[48][48], |q|^2[48], |k|^2[48]) into part[(image * tiles / QC_NCH + chunk)]
    const float* gscale;                         // [2C] power-of-two operand scales of q, k (_hip.gram_scales)
    float* part;
};
#define QC_NCH 4

typedef __attribute__((address_space(3))) char qc_lc;
__device__ __forceinline__ unsigned qc_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
template <typename T>
__device__ __forceinline__ T qc_ld(const qc_lc* base, unsigned voff, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(base + voff + imm);
}
template <typename T>
__device__ __forceinline__ void qc_st(qc_lc* base, unsigned voff, int imm, T v) {
    *reinterpret_cast<__attribute__((address_space(3))) T*>(base + voff + imm) = v;
}
template <int NP>
__device__ __forceinline__ void qc_dma(const float* src, float* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NP + 7) / 8; ++i) {
        const int pc = wave + 8 * i;
        if (pc < NP)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    }
}
template <class F, int... Is>
__device__ __forceinline__ void qc_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void qc_for(F&& f) { qc_for_impl(f, std::make_integer_sequence<int, N>{}); }
// byte offset of channel row c (0 .. 31) inside an image
__device__ __forceinline__ unsigned qc_row(int c) {
    const int i = c & 15;
    const int rk = i < 4 ? i : i < 12 ? i - 4 : i - 8;                 // rank inside {0-3, 12-15} or inside {4-11}
    const int o = (rk >> 1) * 4 + (rk & 1);                            // 0, 1, 4, 5, 8, 9, 12, 13
    // the row's bank-quad phase is (c CS / 16 + o) mod 16; CS / 16 = 108 = 12 (mod 16): take that out so that the phase is o
    const int fix = (16 - ((c * (QC_CS / 16)) & 15)) & 15;
    return (unsigned)(c * QC_CS + 16 * ((o + fix) & 15));
}

// GRAM (KS == 2, C = 48, one head of 48 channels): the stencil outputs of q and k ARE fragments of the Gram MFMA (channel on the
// lane, the wave's 32 pixels of row w as the k index), so the wave keeps its three q tiles as fp16 hi/lo operands (24 registers),
// and every k tile it produces goes - scaled by the channel's power of two, split, never stored - into 3 x 3 MFMAs onto the
// wave's own 9 accumulator tiles; squared norms on the vector pipe.  The accumulators run over the QC_NCH tiles of a chunk
// (a fixed set of consecutive tiles of ONE image, whatever the batch: results do not depend on the batch size); at its end
// the 8 waves' partials meet in LDS in wave order and one record leaves for mdta_reduce / mdta_finalize.  q, k never reach HBM
// (2/3 of this kernel's stores, all of the Gram pass's reads) - VERDICT r2 item 1, for the level where the registers allow it:
// 36 accumulator + 24 operand registers here; c = 96 needs 144 + 48 (two heads of 48: 72 + 48) beside 72 of resident input.
template <int KS, bool GRAM = false>
__global__ __launch_bounds__(512, 2) void qkv_cm_kernel(QcArgs a) {
    IRM_KERNEL_ENTRY();
    static_assert(!GRAM || KS == 2, "GRAM: C = 48");
    constexpr int W1F = KS * 1024, RECF = W1F + 512, RECP = KS * 4 + 2;
    constexpr int SLOT_B = RECF * 4, CF_OFF = W1F * 4;
    constexpr int OSC_OFF = 3 * SLOT_B, MSK_OFF = OSC_OFF + 368 * 4, PL_OFF = MSK_OFF + 368 * 4;   // three record slots
    static_assert(PL_OFF % 16 == 0 && PL_OFF + 2 * QC_IMG <= 160 * 1024, "LDS");
    static_assert(15 * 16 + 368 * 4 <= QC_CS, "row + phase shift inside the row stride");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    qc_lc* lds = (qc_lc*)smem;
    float* slots = smem;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long plane = (long)a.H * a.W;
    const int S = a.S;
    const int per = GRAM ? (a.items / QC_NCH + 7) >> 3 : (a.items + 7) >> 3;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    auto item_of = [&](int round) {
        if constexpr (GRAM) {       // units of QC_NCH consecutive tiles: a workgroup walks whole chunks (per counts chunks here)
            const int i = (round / QC_NCH) * a.gpx + pos;
            return i < per ? (xcd * per + i) * QC_NCH + round % QC_NCH : a.items;
        }
        const int i = round * a.gpx + pos; return i < per ? xcd * per + i : a.items;
    };

    int round = 0;
    int item = item_of(0);
    if (item >= a.items) return;
    float xr[3][KS][8];                             // raw input of the item (free again after the LayerNorm phase: the
                                                    // next item's input is requested into it three iterations before the end)
    f32x4 gacc[GRAM ? 3 : 1][GRAM ? 3 : 1];         // GRAM: [q tile][k tile], over the tiles of the chunk
    float gnq[3] = {0.f, 0.f, 0.f}, gnk[3] = {0.f, 0.f, 0.f};
    if constexpr (GRAM) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) gacc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, g = lane >> 4, r = lane & 15;
        const int b = item / a.tiles, tile = item - b * a.tiles;
        const int ty0 = (tile / a.tiles_x) * QC_TH, tx0 = (tile % a.tiles_x) * QC_TW;
        float* Y = a.Y + (long)b * a.y_bs;
        const int nitem = item_of(round + 1);
        const unsigned vw = qc_opaque((unsigned)(lane * 16));

        // ------------------------------------------------------------ input: lane (r, g) -> pixel slot 16 (wave + 8 j) + r
        bool inside[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 16 * (wave + 8 * j) + r, ph = p / QC_PITCH, pc = p - ph * QC_PITCH;
            const int gy0 = ty0 - 1 + ph, gx0 = tx0 - 1 + pc;
            inside[j] = p < QC_SLOTS && pc < QC_TW + 2 && gy0 >= 0 && gy0 < a.H && gx0 >= 0 && gx0 < a.W;
        }
        auto load_x = [&](int item) {
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int r = t2 & 15, g = (t2 & 63) >> 4;
            const int b = item / a.tiles, tile = item - b * a.tiles;
            const int ty0 = (tile / a.tiles_x) * QC_TH, tx0 = (tile % a.tiles_x) * QC_TW;
            const float* X = a.X + (long)b * a.x_bs;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 16 * (wave + 8 * j) + r, ph = p / QC_PITCH, pc = p - ph * QC_PITCH;
            const int gy0 = ty0 - 1 + ph, gx0 = tx0 - 1 + pc;
            const int gy = min(max(gy0, 0), a.H - 1), gx = min(max(gx0, 0), a.W - 1);
            if (a.x_tm) {
                const float* xp = X + (unsigned)((((gy >> 3) * a.tiles_x + (gx >> 5)) * 256 + (gy & 7) * QC_TW + (gx & 31)) * a.C);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c0 = 32 * ks + 8 * min(g, max((a.C - 32 * ks - 8) / 8, 0));
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(xp + c0), hi4 = *reinterpret_cast<const f32x4*>(xp + c0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xr[j][ks][e] = lo4[e]; xr[j][ks][4 + e] = hi4[e]; }
                }
            } else {
                const unsigned pix = (unsigned)(gy * a.W + gx);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned off = pix + (unsigned)(8 * min(g, max((a.C - 32 * ks - 8) / 8, 0)) * plane);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[j][ks][e] = (X + (long)(32 * ks + e) * plane)[off];
                }
            }
        }
        };
        // KS == 3 (C = 96): the 72 input registers of the next item do not fit beside the stencil's read-ahead (40 registers):
        // measured, the read-ahead is worth more (C 96, 12 x 512^2 in isolation: 1.52 ms with it, 1.60 with the early request
        // instead, 2.25 with both and 155 spilled registers; two thirds of the request early: 1.87) - the input is requested
        // at the start of the item there
        // ... there it goes out behind the LAST 1x1 conv (iteration S - 2), when the operand registers are free, as in the
        // pixel-on-lane kernel
        constexpr bool PREFETCH = KS <= 2 && !GRAM;
        if (round == 0) load_x(item);
        // the previous item's last barrier has passed: every LDS region is free.  Record k lives in slot k % 3.
        qc_dma<RECP>(a.rec, slots, wave, lane);
        qc_dma<RECP>(a.rec + RECF, slots + RECF, wave, lane);
        if (S >= 2) qc_dma<RECP>(a.rec + 2 * RECF, slots + 2 * RECF, wave, lane);

        // ------------------------------------------------------------ LayerNorm + fp16 split (as fused_block.hip)
        qc_h8 xh[3][KS], xl[3][KS];
        // (FULL: C == 32 KS, no channel masks - the same expressions as lnpw_dw_fused_kernel's specialised copies, so that the
        // compiler contracts the same multiply-adds and the two kernels stay bit-identical)
        auto ln_phase = [&](auto FULL_, auto WBK_) {
            constexpr bool FULL = decltype(FULL_)::value;
            constexpr int WBK = decltype(WBK_)::value;
            const float invC = 1.0f / (float)a.C;
            const bool wb = WBK < 0 ? a.ln_mode == IRM_LN_WITHBIAS : WBK == 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float v[KS][8];
                float s = 0.f;
                int kl[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { kl[ks] = FULL ? 8 : a.C - 32 * ks - 8 * g; if (!FULL) asm volatile("" : "+v"(kl[ks])); }
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
                const float var = q * invC;
                const float ms = wb ? var : fmaf(mean, mean, var);
                const float rs = ms > 0.f ? 16.0f / sqrtf(ms) : 0.f;
                const float oscj = a.inv_s1 * sqrtf(ms / (var + a.eps));
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        unsigned hh, ll;
                        irm_split2(__fmul_rn(v[ks][e], rs), __fmul_rn(v[ks][e + 1], rs), hh, ll);
                        reinterpret_cast<unsigned*>(&xh[j][ks])[e / 2] = hh;
                        reinterpret_cast<unsigned*>(&xl[j][ks])[e / 2] = ll;
                    }
                // per-pixel factors of the accumulators (h = acc osc + bias msk; zero outside the image: the reference
                // zero-pads h) for the lanes that will hold this pixel's h values: through LDS
                if (g == 0 && wave + 8 * j < QC_PT) {
                    const int p = 16 * (wave + 8 * j) + r;
                    qc_st<float>(lds, (unsigned)(OSC_OFF + p * 4), 0, inside[j] ? oscj : 0.f);
                    qc_st<float>(lds, (unsigned)(MSK_OFF + p * 4), 0, inside[j] ? 1.f : 0.f);
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 osc4[3], msk4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned po = (unsigned)((16 * min(wave + 8 * j, QC_PT - 1) + 4 * g) * 4);
            osc4[j] = qc_ld<f32x4>(lds, po, OSC_OFF);
            msk4[j] = qc_ld<f32x4>(lds, po, MSK_OFF);
        }

        // one stage of the 1x1 conv: 32 channels (2 tiles) for this wave's 3 pixel tiles -> image img
        const unsigned vrow0 = qc_opaque(qc_row(r)), vrow1 = qc_opaque(qc_row(16 + r));
        auto gemm1h = [&](int slot, int img, int hct) {
            {
                f32x4 acc[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const qc_h8 bh = qc_ld<qc_h8>(lds, vw, slot * SLOT_B + ((hct * KS + ks) * 2) * 1024);
                    const qc_h8 bl = qc_ld<qc_h8>(lds, vw, slot * SLOT_B + ((hct * KS + ks) * 2 + 1) * 1024);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        // (the product order of lnpw_dw_fused_kernel: W_lo x_hi, W_hi x_lo, W_hi x_hi - bit-identical sums)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[j][ks], bl, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[j][ks], bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[j][ks], bh, acc[j], 0, 0, 0);
                    }
                }
                const float b1 = qc_ld<float>(lds, (unsigned)(r * 4), slot * SLOT_B + CF_OFF + 320 * 4 + hct * 64);
                const unsigned vr = hct ? vrow1 : vrow0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (wave + 8 * j < QC_PT) {
                        f32x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = fmaf(acc[j][e], osc4[j][e], b1 * msk4[j][e]);
                        qc_st<f32x4>(lds, vr + (unsigned)((16 * (wave + 8 * j) + 4 * g) * 4), PL_OFF + img * QC_IMG, h);
                    }
                }
            }
        };
        auto gemm1 = [&](int slot, int img) { gemm1h(slot, img, 0); gemm1h(slot, img, 1); };
        // the stencil of a stage: row `wave`, both channel tiles; results to q, k (tile-major) or v
        // the reads of one (16-channel tile, row) unit: issued BEFORE the 1x1 conv of the same channel tile of the next stage, so
        // that their LDS latency passes under its MFMAs; used by the FMAs behind it
        f32x4 sp0[3], sp1[3];
        qc_v2 sp2[3];
        float stap[10];
        auto st_read = [&](int slot, int img, int hct) {
            const unsigned vr = (hct ? vrow1 : vrow0) + (unsigned)((wave * QC_PITCH + 8 * g) * 4);
            const unsigned vt = (unsigned)((16 * hct + r) * 4);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                sp0[dy] = qc_ld<f32x4>(lds, vr, PL_OFF + img * QC_IMG + dy * QC_PITCH * 4);
                sp1[dy] = qc_ld<f32x4>(lds, vr, PL_OFF + img * QC_IMG + dy * QC_PITCH * 4 + 16);
                sp2[dy] = qc_ld<qc_v2>(lds, vr, PL_OFF + img * QC_IMG + dy * QC_PITCH * 4 + 32);
            }
#pragma unroll
            for (int t = 0; t < 10; ++t) stap[t] = qc_ld<float>(lds, vt, slot * SLOT_B + CF_OFF + t * 128);
        };
        qc_h8 gqh[GRAM ? 3 : 1], gql[GRAM ? 3 : 1];      // GRAM: the wave's q tiles as MFMA operands (this tile's 32 pixels of row w)
        auto st_comp = [&](int st, int hct, auto UC) {
            {
                constexpr int U = decltype(UC)::value;             // GRAM: 16-channel tile index 2 st + hct at compile time
                const int ch = 32 * st + 16 * hct + r;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = stap[9];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float P[10] = {sp0[dy][0], sp0[dy][1], sp0[dy][2], sp0[dy][3], sp1[dy][0], sp1[dy][1], sp1[dy][2], sp1[dy][3],
                                         sp2[dy][0], sp2[dy][1]};
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float t = stap[dy * 3 + dx];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaf(t, P[e + dx], o[e]);
                    }
                }
                if constexpr (GRAM && U >= 0 && U < 6) {
                    // q (U < 3) or k tile: scaled by the channel's power of two (exact), squared norm in fp32, fp16 hi/lo split;
                    // a k tile goes straight into the 3 x 3 MFMAs against the resident q tiles (the product order of
                    // mdta_gram_f16x3_kernel: q_lo k_hi, q_hi k_lo, q_hi k_hi)
                    const float sc = a.gscale[16 * U + r];         // (C = 48: the k scales start at index 48 = 16 * 3)
                    float nn = 0.f, xs[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { nn = fmaf(o[e], o[e], nn); xs[e] = o[e] * sc; }
                    qc_h8 hi, lo;
                    irm_split8(xs, hi, lo);
                    if constexpr (U < 3) {
                        gnq[U] += nn; gqh[U] = hi; gql[U] = lo;
                    } else {
                        gnk[U - 3] += nn;
#pragma unroll
                        for (int x = 0; x < 3; ++x) {
                            f32x4& t = gacc[x][U - 3];
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(gql[x], hi, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(gqh[x], lo, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(gqh[x], hi, t, 0, 0, 0);
                        }
                    }
                    return;
                }
                if (ch >= a.M) return;
                if (ch < 2 * a.C) {
                    float* yt = Y + (long)tile * (2L * a.C * 256) + ch * 256 + wave * QC_TW + 8 * g;
                    *reinterpret_cast<f32x4*>(yt) = (f32x4){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(yt + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                } else if (a.v_tm) {
                    float* yt = Y + 2L * a.C * plane + (long)tile * (a.C * 256L) + (wave * QC_TW + 8 * g) * a.C + (ch - 2 * a.C);
#pragma unroll
                    for (int e = 0; e < 8; ++e) yt[e * a.C] = o[e];
                } else {
                    float* yt = Y + (long)ch * plane + (long)(ty0 + wave) * a.W + tx0 + 8 * g;
                    *reinterpret_cast<f32x4*>(yt) = (f32x4){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(yt + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                }
            }
        };

        gemm1(0, 0);                                                     // stage 0 (record 0) -> image 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // iteration it: the 1x1 conv of stage it + 1 (weights: record it + 1) into image (it + 1) & 1, the stencil of stage it
        // (taps: record it + 1) from image it & 1; record it + 3 is requested into the slot record it left.  The LAST request
        // (record S) goes out in iteration S - 3: behind it the wave's vector-memory queue holds nothing it has to wait for
        // any more, so the next item's input is requested right there and stays in flight for the last ~2.5 iterations
        // (the pixel-on-lane kernel can request it only in its last iteration: it has no registers free before).
        constexpr std::integral_constant<int, -1> NOU{};
        if constexpr (GRAM) {
            // C = 48: S = 5 stages, unrolled (the role of a unit - q, k or v tile - is a compile-time property here); the next
            // item's input is requested in the last iteration (the Gram operands and accumulators take the registers the early
            // request would need)
            qc_for<5>([&](auto ITC) {
                constexpr int it = decltype(ITC)::value;
                constexpr int s1c = (it + 1) % 3, s3c = it % 3;
                if constexpr (it + 1 < 5) {
                    if constexpr (it + 3 <= 5) qc_dma<RECP>(a.rec + (long)(it + 3) * RECF, slots + s3c * RECF, wave, lane);
                    st_read(s1c, it & 1, 0); gemm1h(s1c, (it + 1) & 1, 0); st_comp(it, 0, std::integral_constant<int, 2 * it>{});
                    st_read(s1c, it & 1, 1); gemm1h(s1c, (it + 1) & 1, 1); st_comp(it, 1, std::integral_constant<int, 2 * it + 1>{});
                    if constexpr (it + 3 <= 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    load_x(min(nitem, a.items - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    st_read(s1c, it & 1, 0); st_comp(it, 0, std::integral_constant<int, 8>{});
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            });
        } else {
        const int it_pf = max(S - 3, 0);
        int s1 = 1, s3 = 0;                                              // slots of record it + 1 / it + 3
        for (int it = 0; it + 1 < S; ++it) {
            const bool dma = it + 3 <= S;
            if (dma) qc_dma<RECP>(a.rec + (long)(it + 3) * RECF, slots + s3 * RECF, wave, lane);
#ifdef QC_NO_PIPE
            gemm1h(s1, (it + 1) & 1, 0); gemm1h(s1, (it + 1) & 1, 1);
            st_read(s1, it & 1, 0); st_comp(it, 0, NOU);
            st_read(s1, it & 1, 1); st_comp(it, 1, NOU);
#else
            st_read(s1, it & 1, 0); gemm1h(s1, (it + 1) & 1, 0); st_comp(it, 0, NOU);
            st_read(s1, it & 1, 1); gemm1h(s1, (it + 1) & 1, 1); st_comp(it, 1, NOU);
#endif
            if (it <= it_pf) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the stage's stores included)
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (PREFETCH && it == it_pf) { load_x(min(nitem, a.items - 1)); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_s_barrier();
            s1 = s1 == 2 ? 0 : s1 + 1;
            s3 = s3 == 2 ? 0 : s3 + 1;
        }
        {
            // last iteration (peeled: no 1x1 conv, so the operand registers are dead here and - where the early request did not
            // fit - take the next item's input, as in the pixel-on-lane kernel)
            const int it = S - 1;
            if (!PREFETCH || S < 2) {
                load_x(min(nitem, a.items - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
            st_read(s1, it & 1, 0); st_comp(it, 0, NOU);
            st_read(s1, it & 1, 1); st_comp(it, 1, NOU);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        }      // !GRAM
        if constexpr (GRAM) {
            if (round % QC_NCH == QC_NCH - 1) {
                // ---- end of the chunk: the 8 waves' partial records meet in LDS (the image area is free) in wave order
                constexpr int REC = 48 * 48 + 96;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    gnq[t] += __shfl_xor(gnq[t], 16); gnq[t] += __shfl_xor(gnq[t], 32);
                    gnk[t] += __shfl_xor(gnk[t], 16); gnk[t] += __shfl_xor(gnk[t], 32);
                }
                float* mine = reinterpret_cast<float*>(smem) + PL_OFF / 4 + wave * REC;
                float isk[3];
#pragma unroll
                for (int y = 0; y < 3; ++y) isk[y] = 1.0f / a.gscale[48 + 16 * y + r];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const f32x4 sq4 = *reinterpret_cast<const f32x4*>(a.gscale + 16 * x + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float isq = 1.0f / sq4[e];           // (powers of two: exact)
#pragma unroll
                        for (int y = 0; y < 3; ++y) mine[(16 * x + 4 * g + e) * 48 + 16 * y + r] = gacc[x][y][e] * (isq * isk[y]);
                    }
                    if (g == 0) { mine[48 * 48 + 16 * x + r] = gnq[x]; mine[48 * 48 + 48 + 16 * x + r] = gnk[x]; }
                }
                __syncthreads();
                const float* all = reinterpret_cast<const float*>(smem) + PL_OFF / 4;
                float* out = a.part + ((long)b * (a.tiles / QC_NCH) + tile / QC_NCH) * REC;
                for (int e = threadIdx.x; e < REC; e += 512) {
                    float sum = all[e];
#pragma unroll
                    for (int w = 1; w < 8; ++w) sum += all[w * REC + e];
                    out[e] = sum;
                }
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    gnq[x] = 0.f; gnk[x] = 0.f;
#pragma unroll
                    for (int y = 0; y < 3; ++y) gacc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (nitem >= a.items) break;
        item = nitem;
        ++round;
    }
}

template <int KS, bool GRAM = false>
static int qc_launch(QcArgs a, int B, hipStream_t stream) {
    const size_t lds = (size_t)3 * (KS * 1024 + 512) * 4 + 2 * 368 * 4 + 2 * QC_IMG;
    IRM_ALLOW_BIG_LDS((&qkv_cm_kernel<KS, GRAM>));
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return IRM_ELAUNCH;
    a.tiles_x = a.W / QC_TW;
    a.tiles = a.tiles_x * (a.H / QC_TH);
    a.items = B * a.tiles;
    const int per = GRAM ? (a.items / QC_NCH + 7) >> 3 : (a.items + 7) >> 3;
    a.gpx = (n + 7) / 8;
    if (a.gpx > per) a.gpx = per;
    hipLaunchKernelGGL((qkv_cm_kernel<KS, GRAM>), dim3(a.gpx * 8), dim3(512), lds, stream, a);
    return irm_launch_status();
}

// v = the last C channels of dw3x3(W LN(x) + b) into y's v part, and the Gram partial records of q, k (never written) into
// part [B][H W / (256 QC_NCH)][48 * 48 + 96]; C == 48 (one head), (H / 8) (W / 32) % QC_NCH == 0 (header).
extern "C" int irm_qkv_gram_cm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, const float* gscale,
                                         float* part, int ln_mode, float eps, float inv_s1, int B, int C, int H, int W, int x_tm,
                                         int v_tm, hipStream_t stream) {
    if (!rec || !x || !y || !gscale || !part || x == y || B <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (C != 48 || (H & 7) || (W & 31) || ((H / 8) * (W / 32)) % QC_NCH || (long)3 * C * H * W >= (1L << 30)) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(rec) || !irm_aligned16(gscale)) return IRM_EINVAL;
    QcArgs a;
    a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.rec = rec; a.C = C; a.H = H; a.W = W; a.M = 3 * C; a.S = (3 * C + 31) / 32;
    a.ln_mode = ln_mode; a.eps = eps; a.inv_s1 = inv_s1; a.x_tm = x_tm ? 1 : 0; a.v_tm = v_tm ? 1 : 0;
    a.tiles_x = 0; a.tiles = 0; a.items = 0; a.gpx = 0; a.gscale = gscale; a.part = part;
    return qc_launch<2, true>(a, B, stream);
}

// Same contract as irm_qkv_dw_fused_tm_f16x3_f32 (q, k tile-major; v channel-last when v_tm, else planar); header.
extern "C" int irm_qkv_dw_cm_f16x3_f32(const float* rec, const float* x, long x_bs, float* y, long y_bs, int ln_mode, float eps,
                                       float inv_s1, int B, int C, int H, int W, int x_tm, int v_tm, hipStream_t stream) {
    if (!rec || !x || !y || x == y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return IRM_EINVAL;
    if (C > 96 || (C & 15) || (H & 7) || (W & 31) || (long)3 * C * H * W >= (1L << 30)) return IRM_EINVAL;
    if (ln_mode != IRM_LN_WITHBIAS && ln_mode != IRM_LN_BIASFREE) return IRM_EINVAL;
    if ((x_bs & 3) || (y_bs & 3) || !irm_aligned16(x) || !irm_aligned16(y) || !irm_aligned16(rec)) return IRM_EINVAL;
    QcArgs a;
    a.X = x; a.x_bs = x_bs; a.Y = y; a.y_bs = y_bs; a.rec = rec; a.C = C; a.H = H; a.W = W; a.M = 3 * C; a.S = (3 * C + 31) / 32;
    a.ln_mode = ln_mode; a.eps = eps; a.inv_s1 = inv_s1; a.x_tm = x_tm ? 1 : 0; a.v_tm = v_tm ? 1 : 0;
    a.tiles_x = 0; a.tiles = 0; a.items = 0; a.gpx = 0; a.gscale = nullptr; a.part = nullptr;
    switch ((C + 31) / 32) {
        case 3: return qc_launch<3>(a, B, stream);
        case 2: return qc_launch<2>(a, B, stream);
        case 1: return qc_launch<1>(a, B, stream);
    }
    return IRM_EINVAL;
}
