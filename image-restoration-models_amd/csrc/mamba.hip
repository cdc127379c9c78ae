// MaIR / LoSh2D kernels (src/mair/realDenoising/basicsr/models/archs/mairunet_arch.py:226-282):
// the selective-scan SSM recurrence that the reference gets from the third-party mamba_ssm CUDA wheel
// (`selective_scan_fn`, call site mairunet_arch.py:252-258), with the four nested-S scan orders
// (shift_scanf_util.py:206-244), the dt projection (:243), the ShuffleAttn gate (:21-60, :273), the
// direction sum (:274-275), out_norm and the SiLU(z) gate (:277-278) fused around it.
//
// Layout: the scan works channel-LAST (tokens [L][D]) so that at one time step the 64 lanes of a wave
// (= 64 channels d) read one contiguous 256-byte piece; the per-step projections dt_raw / B / C of a
// (direction, pixel) are one contiguous row [4*(R+2N)] read through the scalar cache.  The gather by the
// scan order and the inverse scatter are index arithmetic inside the kernel (u is read at pixel
// ids[k][t], y is written back to pixel ids[k][t]) - the 4x expanded tensors of the reference never exist.
//
// Parallelism over L: chunked scan.  Phase A scans every chunk from h = 0 and records its end state and
// the chunk's total dt; phase B (tiny) carries the state across chunks, h_in[c+1] = exp(A*sum_dt[c]) h_in[c]
// + h_end[c]; phase C rescans every chunk from its true initial state and emits y.  A work unit is one
// wave: (batch, direction, 64-channel block, chunk); it keeps h[N], A[N] and the dt weights in registers.
#include "irm_common.h"

// ---------------------------------------------------------------------------
// [B][R][C] -> [B][C][R] through a padded 32x32 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long in_bs,
                                                        float* __restrict__ out, long out_bs, int R, int C) {
    IRM_KERNEL_ENTRY();
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const float* src = in + (long)b * in_bs;
    float* dst = out + (long)b * out_bs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        if (r < R && c < C) tile[ty + 8 * j][tx] = src[(long)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (r < R && c < C) dst[(long)c * R + r] = tile[tx][ty + 8 * j];
    }
}

extern "C" int irm_transpose_f32(const float* in, long in_bs, float* out, long out_bs, int B, int R, int C,
                                 hipStream_t stream) {
    if (!in || !out || B <= 0 || R <= 0 || C <= 0 || B > 65535) return IRM_EINVAL;
    dim3 grid((C + 31) / 32, (R + 31) / 32, B);
    if (grid.y > 65535) return IRM_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, in_bs, out, out_bs, R, C);
    return irm_launch_status();
}

// ---------------------------------------------------------------------------
struct ScanArgs {
    const float* xT;       // [B][L][D]    u, channel last
    const float* pT;       // [B][L][4*J]  per pixel: for each direction k: dt_raw[R], B[N], C[N]   (J = R + 2N)
    const int* ids;        // [4][L]       scan order: time t of direction k visits pixel ids[k][t]
    const float* dtw;      // [4][D][R]
    const float* dtb;      // [4][D]
    const float* A;        // [4*D][N]     = -exp(A_logs)
    const float* Dskip;    // [4*D]
    float* yT;             // [B][4][L][D] (phase C)
    float* state;          // [2][B][4][DB][nchunk][N][64]  [0]: chunk end states (phase A), [1]: chunk initial states (phase B)
    long state_half;       // elements of one half
    float* sdt;            // [B][4][DB][nchunk][64]     sum of dt over the chunk
    float* ysum;           // [B][4][DB][nchunk][64]     sum of y over the chunk (phase C)
    int L, D, DB, chunk, nchunk;
};

// softplus with the hardware exp2/log2 (abs. error ~1e-7 on dt, far inside the 1e-3 output budget)
// softplus(x) = max(x, 0) + log1p(exp(-|x|)), branch free.  log1p by Kahan's correction (log(w) * e / (w - 1), w = 1 + e):
// dt spans 1e-3 .. 1e-1 in trained Mamba weights, where log(1 + e) alone loses the low bits of e (6e-5 relative at 1e-3).
__device__ __forceinline__ float irm_softplus(float x) {
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
    const float w = 1.0f + e, dd = w - 1.0f;
    const float l = __builtin_amdgcn_logf(w) * 0.69314718055994531f * (e * __builtin_amdgcn_rcpf(dd));
    return fmaxf(x, 0.0f) + (dd == 0.0f ? e : l);
}

typedef float sc_v2 __attribute__((ext_vector_type(2)));
typedef float sc_v4 __attribute__((ext_vector_type(4)));

// The per-step row [dt_raw R | B N | C N] of a (direction, pixel) is wave-uniform.  It is fetched with ONE coalesced
// vector load per 64 elements (lane j holds element j) two batches of TU steps ahead - vector loads return in
// order, so the prefetch depth is free of the one-wait-for-everything rule of scalar loads -, parked in a small
// LDS ring private to the wave (no barriers: one wave, in-order LDS), and read back with same-address (broadcast)
// ds_read_b128: four row elements per instruction arrive in VGPRs of every lane and feed packed fp32 instructions
// directly - no v_readlane, no SGPR budget.  States go in pairs: per pair 2 exp2 + 4 packed operations
// (v_pk_mul_f32 / v_pk_fma_f32).  Pixel ids travel as one vector load per batch (lane i = step i), three
// batches ahead; u one load per step, two batches ahead.
template <int N, int R, bool EMIT>
__global__ __launch_bounds__(64, (N <= 8 ? 4 : N == 16 ? 3 : 2)) void scan_chunk_kernel(ScanArgs a) {
    IRM_KERNEL_ENTRY();   // (waves per SIMD: caps the
    constexpr int J = R + 2 * N, JV = (J + 63) / 64;                                                    // scheduler's read hoisting)
    constexpr int TU = 8;                                     // time steps per batch
    constexpr int RP = (R + 3) & ~3;                          // LDS row: [dt_raw, padded to 16 bytes | B | C]
    constexpr int JS = RP + 2 * N;
    static_assert(N % 4 == 0, "states are read four at a time");
    __shared__ __attribute__((aligned(16))) float ring[2][TU][JS];
    const int lane = threadIdx.x;
    const int c = blockIdx.x, kdb = blockIdx.y, b = blockIdx.z;
    const int k = kdb / a.DB, db = kdb % a.DB;
    const int d = db * 64 + lane;
    const bool on = d < a.D;
    const int dc = on ? d : a.D - 1;                          // idle lanes shadow a valid channel

    sc_v2 Ac[N / 2], h[N / 2];
    float wdt[R];
#pragma unroll
    for (int n = 0; n < N; ++n) Ac[n / 2][n % 2] = a.A[((long)k * a.D + dc) * N + n] * 1.44269504088896341f;   // exp(x) = exp2(x log2 e)
#pragma unroll
    for (int r = 0; r < R; ++r) wdt[r] = a.dtw[((long)k * a.D + dc) * R + r];
    const float bias = a.dtb[k * a.D + dc];
    const float dsk = a.Dskip[k * a.D + dc];

    const long unit = (((long)b * 4 + k) * a.DB + db) * a.nchunk + c;
    float* st = a.state + (EMIT ? a.state_half : 0) + unit * N * 64;
#pragma unroll
    for (int n = 0; n < N; ++n) h[n / 2][n % 2] = EMIT ? st[n * 64 + lane] : 0.0f;

    const int* ids = a.ids + (long)k * a.L;
    const float* xT = a.xT + (long)b * a.L * a.D + dc;
    const float* pT = a.pT + (long)b * a.L * 4 * J + k * J;
    float* yT = EMIT ? a.yT + ((long)b * 4 + k) * a.L * a.D + d : nullptr;
    float sum_dt = 0.0f, sum_y = 0.0f;
    const int t0 = c * a.chunk, t1 = min(t0 + a.chunk, a.L);

    // element e = 64 jv + lane of a row -> its LDS position
    int epos[JV], eidx[JV];
#pragma unroll
    for (int jv = 0; jv < JV; ++jv) {
        const int e = jv * 64 + lane;
        eidx[jv] = min(e, J - 1);
        epos[jv] = e < R ? e : (e < J ? e + (RP - R) : -1);
    }
    auto load_ids = [&](int t) { return ids[min(t + (lane & (TU - 1)), a.L - 1)]; };
    auto load_batch = [&](float (&rw)[TU][JV], float (&u)[TU], int idv) {
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int p = __builtin_amdgcn_readlane(idv, i);
            u[i] = xT[(long)p * a.D];
#pragma unroll
            for (int jv = 0; jv < JV; ++jv) rw[i][jv] = pT[(long)p * (4 * J) + eidx[jv]];
        }
    };
    auto park = [&](const float (&rw)[TU][JV], int slot) {
#pragma unroll
        for (int i = 0; i < TU; ++i)
#pragma unroll
            for (int jv = 0; jv < JV; ++jv)
                if (epos[jv] >= 0) ring[slot][i][epos[jv]] = rw[i][jv];
    };

    int id0 = load_ids(t0), id1 = load_ids(t0 + TU), id2 = load_ids(t0 + 2 * TU);
    float u0[TU], u1[TU], u2[TU], r1[TU][JV], r2[TU][JV];
    load_batch(r1, u0, id0);
    park(r1, 0);
    load_batch(r1, u1, id1);
    int slot = 0;
    for (int t = t0; t < t1; t += TU, slot ^= 1) {
        const int id3 = load_ids(t + 3 * TU);
        load_batch(r2, u2, id2);
        park(r1, slot ^ 1);                                   // batch t + TU (requested one iteration ago)
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            // steps past the chunk end run with dt = 0 (decay 1, input 0: the state is untouched) and store nothing:
            // the batch stays one straight-line block, free of per-step branches
            const bool live = t + i < t1;
            const float* row = &ring[slot][i][0];
            float dt = bias;
#pragma unroll
            for (int r = 0; r < RP; r += 4) {
                const sc_v4 q = *reinterpret_cast<const sc_v4*>(row + r);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (r + e < R) dt = fmaf(wdt[r + e], q[e], dt);
            }
            dt = irm_softplus(dt) * (live ? 1.0f : 0.0f);
            const float du = dt * u0[i];
            sc_v2 y2 = {dsk * u0[i], 0.f};
#pragma unroll
            for (int n = 0; n < N; n += 4) {
                const sc_v4 Bq = *reinterpret_cast<const sc_v4*>(row + RP + n);
                const sc_v4 Cq = *reinterpret_cast<const sc_v4*>(row + RP + N + n);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int m = (n + 2 * hh) / 2;
                    const sc_v2 x = (sc_v2){dt, dt} * Ac[m];
                    const sc_v2 e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    const sc_v2 Bv = {Bq[2 * hh], Bq[2 * hh + 1]}, Cv = {Cq[2 * hh], Cq[2 * hh + 1]};
                    h[m] = e * h[m] + (sc_v2){du, du} * Bv;
                    if (EMIT) y2 = h[m] * Cv + y2;
                }
            }
            sum_dt += dt;
            if (EMIT) {
                const float y = live ? y2.x + y2.y : 0.0f;
                const int p = __builtin_amdgcn_readlane(id0, i);
                if (on && live) yT[(long)p * a.D] = y;
                sum_y += y;
            }
            // without this the scheduler hoists the LDS reads of all 8 steps to the top (388 VGPRs at N = 32)
            if (i & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            u0[i] = u1[i]; u1[i] = u2[i];
#pragma unroll
            for (int jv = 0; jv < JV; ++jv) r1[i][jv] = r2[i][jv];
        }
        id0 = id1; id1 = id2; id2 = id3;
    }
    if (EMIT) {
        a.ysum[unit * 64 + lane] = on ? sum_y : 0.0f;
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) st[n * 64 + lane] = h[n / 2][n % 2];
        a.sdt[unit * 64 + lane] = sum_dt;
    }
}

// phase B: carry the state across the chunks of one (batch, direction, channel block):
// h_in[0] = 0, h_in[c+1] = exp(A * sum_dt[c]) * h_in[c] + h_end[c].  A serial walk over all chunks is a chain of
// several hundred dependent steps on a handful of waves (it used to cost as much as a scan phase), so the chunks
// are cut into 16 groups, one wave each: the wave composes its group (decay = exp(A * sum of the group's dt), state
// from zero), the 16 composites meet in LDS, every wave folds the groups before it (<= 15 LDS steps) and walks its
// group again to emit the initial states.  A workgroup owns 8 states of 64 channels.
template <int N>
__global__ __launch_bounds__(1024) void scan_carry_kernel(ScanArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int NB = N < 8 ? N : 8, G = 16;
    __shared__ float comp[G][NB + 1][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int kdb = blockIdx.x, b = blockIdx.y, n0 = blockIdx.z * NB;
    const int k = kdb / a.DB, db = kdb % a.DB;
    const int dc = min(db * 64 + lane, a.D - 1);
    float Ac[NB], h[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) { Ac[n] = a.A[((long)k * a.D + dc) * N + n0 + n]; h[n] = 0.0f; }
    const long base = (((long)b * 4 + k) * a.DB + db) * a.nchunk;
    const float* __restrict__ hend = a.state + base * N * 64 + n0 * 64 + lane;
    float* __restrict__ hin = a.state + a.state_half + base * N * 64 + n0 * 64 + lane;
    const float* __restrict__ sdt = a.sdt + base * 64 + lane;
    const int cpg = (a.nchunk + G - 1) / G;
    const int c0 = min(g * cpg, a.nchunk), c1 = min(c0 + cpg, a.nchunk);
    float ssum = 0.0f;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float s = sdt[(long)c * 64];
        float e[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) e[n] = hend[((long)c * N + n) * 64];
#pragma unroll
        for (int n = 0; n < NB; ++n) h[n] = fmaf(__expf(Ac[n] * s), h[n], e[n]);
        ssum += s;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) comp[g][n][lane] = h[n];
    comp[g][NB][lane] = ssum;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NB; ++n) h[n] = 0.0f;
    for (int j = 0; j < g; ++j) {
        const float s = comp[j][NB][lane];
#pragma unroll
        for (int n = 0; n < NB; ++n) h[n] = fmaf(__expf(Ac[n] * s), h[n], comp[j][n][lane]);
    }
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float s = sdt[(long)c * 64];
        float e[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) e[n] = hend[((long)c * N + n) * 64];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            hin[((long)c * N + n) * 64] = h[n];
            h[n] = fmaf(__expf(Ac[n] * s), h[n], e[n]);
        }
    }
}

// per (batch, direction, channel block): ysum[chunk 0] <- sum over chunks (fixed order: 16 interleaved partial
// sums, four loads in flight each, combined in wave order), so the gate reads one value per channel
__global__ __launch_bounds__(1024) void ysum_reduce_kernel(float* __restrict__ ysum, int nchunk) {
    IRM_KERNEL_ENTRY();
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* p = ysum + (long)blockIdx.x * nchunk * 64 + lane;
    float s = 0.0f;
    int c = w;
    for (; c + 48 < nchunk; c += 64) {
        const float v0 = p[(long)c * 64], v1 = p[(long)(c + 16) * 64], v2 = p[(long)(c + 32) * 64], v3 = p[(long)(c + 48) * 64];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; c < nchunk; c += 16) s += p[(long)c * 64];
    part[w][lane] = s;
    __syncthreads();
    if (w == 0) {
        float t = part[0][lane];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += part[j][lane];
        p[0] = t;
    }
}

template <int N, int R>
static int scan_launch(const ScanArgs& a, int B, hipStream_t stream) {
    dim3 g1(a.nchunk, 4 * a.DB, B), g2(4 * a.DB, B, N < 8 ? 1 : N / 8);
    hipLaunchKernelGGL((scan_chunk_kernel<N, R, false>), g1, dim3(64), 0, stream, a);
    hipLaunchKernelGGL((scan_carry_kernel<N>), g2, dim3(1024), 0, stream, a);
    hipLaunchKernelGGL((scan_chunk_kernel<N, R, true>), g1, dim3(64), 0, stream, a);
    return irm_launch_status();
}

extern "C" int irm_selective_scan_f32(const float* xT, const float* pT, const int* ids, const float* dtw,
                                      const float* dtb, const float* A, const float* Dskip, float* yT,
                                      float* state, float* sdt, float* ysum, int B, int L, int D, int N, int R,
                                      int chunk, hipStream_t stream) {
    if (!xT || !pT || !ids || !dtw || !dtb || !A || !Dskip || !yT || !state || !sdt || !ysum) return IRM_EINVAL;
    if (B <= 0 || L <= 0 || D <= 0 || chunk <= 0 || B > 65535) return IRM_EINVAL;
    ScanArgs a{xT, pT, ids, dtw, dtb, A, Dskip, yT, state, 0, sdt, ysum, L, D, (D + 63) / 64, chunk,
               (L + chunk - 1) / chunk};
    a.state_half = (long)B * 4 * a.DB * a.nchunk * N * 64;
    if (4 * a.DB > 65535) return IRM_EINVAL;
    if (N == 4 && R == 3) return scan_launch<4, 3>(a, B, stream);
    if (N == 8 && R == 6) return scan_launch<8, 6>(a, B, stream);
    if (N == 16 && R == 12) return scan_launch<16, 12>(a, B, stream);
    if (N == 32 && R == 24) return scan_launch<32, 24>(a, B, stream);
    return IRM_EINVAL;                                  // (d_state, dt_rank) pairs of MaIRUNet / MaIR
}

// ---------------------------------------------------------------------------
// ShuffleAttn gate: g[b][k'][d] = sigmoid(bias[4d+k'] + sum_k W[4d+k'][k] * mean_HW(y[k][d]))
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ ysum, const float* __restrict__ gw,
                                                   const float* __restrict__ gb, float* __restrict__ gate, int D,
                                                   int DB, int nchunk, float inv_L) {
    IRM_KERNEL_ENTRY();
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;           // e = kq * D + d
    if (e >= 4 * D) return;
    const int kq = e / D, d = e % D;
    const int db = d >> 6, lane = d & 63;
    float acc = gb[4 * d + kq];
    for (int k = 0; k < 4; ++k) {
        const float s = ysum[((((long)b * 4 + k) * DB + db) * nchunk) * 64 + lane];   // reduced into chunk slot 0
        acc = fmaf(gw[(4 * d + kq) * 4 + k], s * inv_L, acc);
    }
    gate[(long)b * 4 * D + e] = 1.0f / (1.0f + expf(-acc));
}

// combine: v[p][d] = sum_k y[k][p][d] * g[k][d]; LayerNorm over d (out_norm); * silu(z[d][p]); planar output.
// One workgroup = 4 PW pixels; a wave normalises PW pixels (lanes over d), the tile is transposed through LDS.
// PW follows the plane size only (8 for large planes; 2 for the 64x64 / 32x32 levels of a single image, which
// would otherwise run on 32 - 128 workgroups).
struct CombArgs {
    const float* yT;      // [B][4][L][D]
    const float* gate;    // [B][4][D]
    const float* nw;      // [D] out_norm weight
    const float* nb;      // [D] out_norm bias
    const float* z;       // planar [B][D][L] (batch stride z_bs)
    long z_bs;
    float* out;           // planar [B][D][L]
    long out_bs;
    int L, D;
    float eps;
};

template <int DV, int PW>      // DV = ceil(D / 64) values per lane
__global__ __launch_bounds__(256) void combine_kernel(CombArgs a) {
    IRM_KERNEL_ENTRY();
    constexpr int PX = 4 * PW, TS = PX + 1;
    extern __shared__ float tile[];                          // [D][PX + 1]
    const int b = blockIdx.y, p0 = blockIdx.x * PX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* g = a.gate + (long)b * 4 * a.D;
    float gk[4][DV], nw[DV], nb[DV];
#pragma unroll
    for (int i = 0; i < DV; ++i) {
        const int d = min(i * 64 + lane, a.D - 1);
        nw[i] = a.nw[d]; nb[i] = a.nb[d];
#pragma unroll
        for (int k = 0; k < 4; ++k) gk[k][i] = g[k * a.D + d];
    }
    // 8 pixels per wave, all loads and both butterfly reductions of the 8 pixels interleaved (ILP)
    float v[PW][DV], s[PW], sq[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        const int p = min(p0 + wave * PW + q, a.L - 1);
        s[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            const int d = i * 64 + lane;
            float t = 0.0f;
            if (d < a.D) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    t = fmaf(a.yT[(((long)b * 4 + k) * a.L + p) * a.D + d], gk[k][i], t);
            }
            v[q][i] = t;
            s[q] += t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < PW; ++q) s[q] += __shfl_xor(s[q], o);
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        s[q] /= (float)a.D;                                  // mean
        sq[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            const float dlt = (i * 64 + lane < a.D) ? v[q][i] - s[q] : 0.0f;
            sq[q] += dlt * dlt;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < PW; ++q) sq[q] += __shfl_xor(sq[q], o);
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        const int px = wave * PW + q;
        const float rstd = 1.0f / sqrtf(sq[q] / (float)a.D + a.eps);
        if (p0 + px < a.L) {
#pragma unroll
            for (int i = 0; i < DV; ++i) {
                const int d = i * 64 + lane;
                if (d < a.D) tile[d * TS + px] = (v[q][i] - s[q]) * rstd * nw[i] + nb[i];
            }
        }
    }
    __syncthreads();
    const float* z = a.z + (long)b * a.z_bs;
    float* out = a.out + (long)b * a.out_bs;
    const int px = threadIdx.x % PX;
    if (p0 + px < a.L) {
        for (int d = threadIdx.x / PX; d < a.D; d += 256 / PX) {
            const float zz = z[(long)d * a.L + p0 + px];
            out[(long)d * a.L + p0 + px] = tile[d * TS + px] * (zz / (1.0f + expf(-zz)));
        }
    }
}

extern "C" int irm_losh_combine_f32(float* ysum, const float* gw, const float* gb, float* gate,
                                    const float* yT, const float* nw, const float* nb, const float* z, long z_bs,
                                    float* out, long out_bs, int B, int L, int D, int nchunk, float eps,
                                    hipStream_t stream) {
    if (!ysum || !gw || !gb || !gate || !yT || !nw || !nb || !z || !out) return IRM_EINVAL;
    if (B <= 0 || L <= 0 || D <= 0 || nchunk <= 0 || B > 65535 || D > 1024) return IRM_EINVAL;
    const int DB = (D + 63) / 64;
    hipLaunchKernelGGL(ysum_reduce_kernel, dim3(B * 4 * DB), dim3(1024), 0, stream, ysum, nchunk);
    hipLaunchKernelGGL(gate_kernel, dim3((4 * D + 255) / 256, B), dim3(256), 0, stream, ysum, gw, gb, gate, D, DB,
                       nchunk, 1.0f / (float)L);
    CombArgs a{yT, gate, nw, nb, z, z_bs, out, out_bs, L, D, eps};
    const int pw = L >= 16384 ? 8 : 2;
    const size_t lds = (size_t)D * (4 * pw + 1) * sizeof(float);
    dim3 grid((L + 4 * pw - 1) / (4 * pw), B);
#define IRM_COMB(DVV)                                                                                              \
    do {                                                                                                           \
        if (pw == 8) {                                                                                             \
            IRM_ALLOW_BIG_LDS((&combine_kernel<DVV, 8>));                                                          \
            hipLaunchKernelGGL((combine_kernel<DVV, 8>), grid, dim3(256), lds, stream, a);                         \
        } else {                                                                                                   \
            IRM_ALLOW_BIG_LDS((&combine_kernel<DVV, 2>));                                                          \
            hipLaunchKernelGGL((combine_kernel<DVV, 2>), grid, dim3(256), lds, stream, a);                         \
        }                                                                                                          \
    } while (0)
    if (DB <= 2) IRM_COMB(2);
    else if (DB <= 3) IRM_COMB(3);
    else if (DB <= 6) IRM_COMB(6);
    else if (DB <= 12) IRM_COMB(12);
    else IRM_COMB(16);
#undef IRM_COMB
    return irm_launch_status();
}
