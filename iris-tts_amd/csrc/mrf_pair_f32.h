// mrf_pair_f32.h -- one ResBlock conv PAIR of all MRF branches in one launch, exact fp32 (gfx950), C = 32 / 64.
//
// Reference semantics: one iteration of ResBlock.forward's loop (src/iris/hifigan_pretrained.py:64-71)
//     xt = Conv1d_{k, dil d}(LeakyReLU(x));  y = Conv1d_{k, dil 1}(LeakyReLU(xt)) + x
// for the branches k = 3 / 7 / 11 of a stage (hifigan_pretrained.py:130-136).  The numbers are bit for bit those of the
// two separate launches of mrf_conv_mfma_f32_kernel: every element is the same fp32 fmaf chain on
// v_mfma_f32_32x32x2_f32 (taps ascending, groups of 8 channels ascending, channels 0,4,1,5,2,6,3,7 inside a group),
// xt = (acc + bias1) + 0, y = (acc + bias2) + x -- the fp32 sibling of mrf_pair_bf16.h.
//
// Why: the two narrow stages (C = 64 / 32) run the shortest phases of the persistent kernel (a k = 3 phase at C = 32 is
// 6,000 cycles between two barriers, an LDS refill and an epilogue) and are its laggards (124 / 111 TFLOP/s against
// 133 at C = 128); and on short inputs every launch costs a ramp and a tail whatever it computes.  Fused, xt never
// leaves the CU: a block stages ONE window, runs conv1, keeps LeakyReLU(xt) in the same LDS region, runs conv2 from
// there -- one launch, one window, one epilogue per pair instead of two of each.  The price: the last k-1 rows of
// conv2's tile are computed and thrown away (they would need xt rows the block does not have), 1.6-7.8 % at 128 rows.
// Measured against the separate launches (release builds, batch 1): the stage's MRF time falls by 17-28 % (C = 32) and
// 0-16 % (C = 64) at 100-700 frames, by 7 % / 1 % at 1000 frames, 5 % / 1 % at batch 32 x 500.
//
// Job order: ALL jobs of the heaviest branch (k = 11) first, then k = 7, then k = 3.  Blocks are not persistent -- a block
// is one (tile, branch) job of up to 40 us -- so the tail of a launch is as long as the jobs dispatched last: with the
// k = 3 jobs last it is a quarter of what a tile-major order leaves (that order was 10-30 % slower whenever a launch
// was more than about one round of the chip).
//
// Work split: a 256-thread block owns M = WT*MT*32 rows of xt and all C channels (WC*32 == C) of one branch of one
// batch item; blocks advance by T_OUT = M - (k-1) rows.  Same LDS image as the other fp32 kernels (row stride C+4
// floats = 4*odd: conflict-free b128 fragment reads and xt writes), same packed weights, weight fragments four groups
// ahead in a register ring, the epilogue's 16-byte pieces straight from the D layout.  Never in place: a block's
// window overlaps the rows its neighbours write (the caller alternates between a branch's two workspace buffers).
#pragma once
#include "mrf_conv_mfma_f32.h"

namespace iris {

struct PairProblemF32 {
    const float* x;     // [B, L, C]: input of the pair and its residual
    const f32x4* w1;    // packed weights of convs1[m] (pack_conv1d_weights)
    const f32x4* w2;    // ... of convs2[m]
    const float* b1;    // [C]
    const float* b2;
    float* y;           // [B, L, C]
    int ks;             // taps of both convs
    int dil;            // dilation of conv1 (conv2: 1)
};

struct PairLaunchF32 {
    PairProblemF32 p[kMaxGroup];
    int B, L, C;
    float slope;
    int nz;               // branches
    int Gp, n_ct;         // packed-weight geometry (packed_groups / packed_cotiles of C)
    int n_jobs;           // tiles x branches (tiles = ceil(L / smallest T_OUT))
    int jobs_per_xcd;     // ceil(n_jobs / 8)
    int z_major;          // job order: branch-major (heaviest first) instead of tile-major
};

// The MFMA loop of one conv over the LDS window: NG = KS * GPC groups of 8 channels; weight fragment n + DB is
// requested while group n computes (the ring holds groups 0 .. DB-1 on entry); activation fragments one group ahead.
template <int KS, int MT, int GPC, int DB>
__device__ __forceinline__ void pair_f32_mma(f32x16 (&acc)[MT], f32x4 (&bw)[DB + 1], const float* aptr, int dilS, int S,
                                             __amdgpu_buffer_rsrc_t wr, unsigned wvoff, unsigned wbytes_group, unsigned tap_bytes) {
    constexpr int NG = KS * GPC;
    auto a_ptr = [&](int n) { return aptr + (n / GPC) * dilS + 8 * (n % GPC); };
    f32x4 av[2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(a_ptr(0) + m * 32 * S);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        if (n + DB < NG)
            bw[(n + DB) % (DB + 1)] = buf_load4(wr, wvoff, (unsigned)((n + DB) / GPC) * tap_bytes + (unsigned)((n + DB) % GPC) * wbytes_group);
        if (n + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MT; ++m) av[(n + 1) & 1][m] = *reinterpret_cast<const f32x4*>(a_ptr(n + 1) + m * 32 * S);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[n % (DB + 1)][e], av[n & 1][m][e], acc[m], 0, 0, 0);
        // one request slotted behind each MFMA (masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read), as in mrf_conv_mfma_f32.h
        {
            int ds_left = (n + 1 < NG) ? MT : 0, vm_left = (n + DB < NG) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 4 * MT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int WT, int WC, int MT, int C, int MINB>
__global__ void __launch_bounds__(256, MINB) mrf_pair_f32_kernel(const PairLaunchF32 a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(WT * WC == 4 && WC * 32 == C, "a block owns all C channels");
    constexpr int S = C + 4;
    constexpr int QPR = C / 4;                                            // 16-byte quads per window row (8 or 16)
    constexpr int GPC = C / 8;
    constexpr int M = WT * MT * 32;
    constexpr int DB = 4;                                                 // weight ring depth (groups); DB <= GPC
    constexpr int NQ = ((M + kMrfSpanMax) * QPR + 255) / 256;           // staged quads per thread: all in flight at once
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;

    // job -> (tile, branch).  z_major: all jobs of the heaviest branch first (the tail of the launch is then made of the
    // short k = 3 blocks); else tile-major with contiguous job ranges per XCD (neighbouring tiles share halo rows through
    // that XCD's L2)
    int tile, zr;
    if (a.z_major) {
        const int job = (int)blockIdx.x;
        if (job >= a.n_jobs) return;
        const int tiles = a.n_jobs / a.nz;
        zr = job / tiles; tile = job - zr * tiles;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int job = xcd * a.jobs_per_xcd + slot;
        if (job >= a.n_jobs) return;
        tile = job / a.nz; zr = job - tile * a.nz;
    }
    const int z = a.nz - 1 - zr;                      // heaviest branch first
    PairProblemF32 p = a.p[0];
    if (z == 1) p = a.p[1];
    if (z == 2) p = a.p[2];
    if (z == 3) p = a.p[3];
    const int ks = p.ks, dil = p.dil;
    const int h2 = (ks - 1) / 2, h1 = dil * (ks - 1) / 2;
    const int T_OUT = M - (ks - 1);
    const int o0 = tile * T_OUT;
    const int L = a.L;
    if (o0 >= L) return;                              // (tiles are counted for the smallest T_OUT of the launch)
    const int b = blockIdx.y;
    const float slope = a.slope;

    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 4u;
    const size_t item = (size_t)b * L * C;
    const unsigned wbytes_group = (unsigned)a.n_ct * 64u * 16u;
    const unsigned tap_bytes = (unsigned)a.Gp * wbytes_group;
    const unsigned wvoff = (unsigned)(wc * 64 + lane) * 16u;              // this wave's 32-wide channel tile = wc
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + item, tensor_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + item, tensor_bytes);
    const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(p.w1, (unsigned)ks * tap_bytes);
    const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(p.w2, (unsigned)ks * tap_bytes);
    const float* aptr = lds + (wt * MT * 32 + lo) * S + 4 * hi;
    const int co4 = wc * 32 + 4 * hi;                                     // this lane's channels: co4 + 8g + {0..3}

    // ---- 1. requests, oldest first (vmcnt retires in order): bias1, conv1's first weight fragments (L2), the window (HBM) ----
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(p.b1 + co4 + 8 * g);
    f32x4 bw[DB + 1];
#pragma unroll
    for (int d = 0; d < DB; ++d) bw[d] = buf_load4(wr1, wvoff, (unsigned)d * wbytes_group);
    {
        const int in_row0 = o0 - h2 - h1, R = M + (ks - 1) * dil, total = R * QPR;
        f32x4 st[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / QPR, q = idx & (QPR - 1);
            const int row = in_row0 + r;
            const bool ok = idx < total && row >= 0 && row < L;
            st[u] = buf_load4(xr, ok ? (unsigned)(row * C + 4 * q) * 4u : kOobOffset, 0);
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / QPR, q = idx & (QPR - 1);
            f32x4 v = st[u];                      // LeakyReLU(x) = max(x, slope x) for 0 <= slope <= 1 (checked by the host)
            v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
            v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
            if (idx < total) *reinterpret_cast<f32x4*>(lds + r * S + 4 * q) = v;
        }
    }
    __syncthreads();

    f32x16 acc[MT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    };
    // (a macro, not a lambda: the three unrolled loops must be inlined into the kernel)
#define IRIS_PAIR_F32_CONV(WR, DILS)                                                                                        \
    do {                                                                                                                    \
        if (ks == 3)       pair_f32_mma<3, MT, GPC, DB>(acc, bw, aptr, (DILS), S, (WR), wvoff, wbytes_group, tap_bytes);    \
        else if (ks == 7)  pair_f32_mma<7, MT, GPC, DB>(acc, bw, aptr, (DILS), S, (WR), wvoff, wbytes_group, tap_bytes);    \
        else               pair_f32_mma<11, MT, GPC, DB>(acc, bw, aptr, (DILS), S, (WR), wvoff, wbytes_group, tap_bytes);   \
    } while (0)
    // ---- 2. conv1 ------------------------------------------------------------------------------------------------
    zero_acc();
    IRIS_PAIR_F32_CONV(wr1, dil * S);
    // conv2's first fragments, its bias and the residual pieces travel during step 3
#pragma unroll
    for (int d = 0; d < DB; ++d) bw[d] = buf_load4(wr2, wvoff, (unsigned)d * wbytes_group);
    f32x4 bias2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias2[g] = *reinterpret_cast<const f32x4*>(p.b2 + co4 + 8 * g);
    unsigned ovoff[MT];                                                   // piece (m, g = 0) of this lane, or out of range
    f32x4 resv[MT * 4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int im = (wt * MT + m) * 32 + lo;
        const int o = o0 + im;
        ovoff[m] = (im < T_OUT && o < L) ? (unsigned)(o * C + co4) * 4u : kOobOffset;
#pragma unroll
        for (int g = 0; g < 4; ++g) resv[m * 4 + g] = buf_load4(xr, ovoff[m], (unsigned)(8 * g) * 4u);
    }
    __syncthreads();                                                      // every wave is done with the x window
    // ---- 3. xt -> LDS: LeakyReLU((acc + bias1) + 0), zero outside [0, L) (conv2's zero padding) -----------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row_l = (wt * MT + m) * 32 + lo;
        const int row_g = o0 - h2 + row_l;
        const bool inside = row_g >= 0 && row_g < L;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (acc[m][4 * g + e] + bias4[g][e]) + 0.f;   // what the separate launch stores (residual-free: + 0)
                v[e] = inside ? fmaxf(t, t * slope) : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + row_l * S + co4 + 8 * g) = v;
        }
    }
    __syncthreads();
    // ---- 4. conv2 (dilation 1; rows M .. M+k-2 of the window hold stale values: they only reach outputs >= T_OUT) -------
    zero_acc();
    IRIS_PAIR_F32_CONV(wr2, S);
#undef IRIS_PAIR_F32_CONV
    // ---- 5. epilogue: (acc + bias2) + x, 16-byte pieces straight from the D layout -----------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[m][4 * g + e] = (acc[m][4 * g + e] + bias2[g][e]) + resv[m * 4 + g][e];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 outv[MT * 4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            outv[m * 4 + g] = f32x4{acc[m][4 * g + 0], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < MT * 4; ++idx) buf_store4(outv[idx], yr, ovoff[idx / 4], (unsigned)(8 * (idx % 4)) * 4u);
    asm volatile("s_nop 1");           // explicit wait states behind the dwordx4 store group (see mrf_conv_mfma_f32.h)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < MT * 4; ++idx) asm volatile("" :: "v"(outv[idx]));
}

// ---- launch ----------------------------------------------------------------------------------------
#ifndef IRIS_PAIR_F32_DEFAULT
#define IRIS_PAIR_F32_DEFAULT 1        // (A/B builds: -DIRIS_PAIR_F32_DEFAULT=0 keeps the separate launches)
#endif
#ifndef IRIS_PAIR_F32_SHORT_FACTOR
#define IRIS_PAIR_F32_SHORT_FACTOR 2   // C = 64: 64-row tiles while 128-row jobs number fewer than this many per CU
#endif
struct PairTileF32 { int WT, WC, MT, MINB, M; };

// Tile height (release builds, profiles/r02zd_*, r02zh_*): 128 rows -- four blocks per CU at C = 32, three at C = 64; 64 rows at
// C = 64 when 128 would leave CUs without work (short inputs: the launch time is then one block's serial time, which is
// proportional to the tile height).  Taller tiles (256 / 192 rows: fewer rows thrown away) made no difference on large
// problems (+-1 %) and are not instantiated.
struct PairPlanF32 { PairTileF32 tile; long long tiles; };

inline PairPlanF32 pair_f32_plan(const PairLaunchF32& a, int nz) {
    PairPlanF32 pl;
    int kmax = 1;
    for (int j = 0; j < nz; ++j) if (a.p[j].ks > kmax) kmax = a.p[j].ks;
    const int n_cu = device_cu_count();
    auto tiles_of = [&](int M) { const int t_out = M - (kmax - 1); return (long long)((a.L + t_out - 1) / t_out); };
    if (a.C == 32) {
        pl.tile = PairTileF32{4, 1, 1, 4, 128};
        pl.tiles = tiles_of(128);
    } else {
        pl.tile = PairTileF32{2, 2, 2, 3, 128};
        pl.tiles = tiles_of(128);
        if (pl.tiles * nz * a.B < (long long)IRIS_PAIR_F32_SHORT_FACTOR * n_cu) { pl.tile = PairTileF32{2, 2, 1, 4, 64}; pl.tiles = tiles_of(64); }
    }
    return pl;
}

// True when the pair launch `a` (nz branches, a.C channels) can take the fused kernel.
inline bool pair_f32_applicable(const PairLaunchF32& a, int nz) {
    if (nz < 1 || nz > 4 || (a.C != 32 && a.C != 64)) return false;
    if (!(a.slope >= 0.f && a.slope <= 1.f)) return false;                        // LeakyReLU is evaluated as max(v, slope*v)
    if ((double)a.L * a.C * 4.0 >= 2147483648.0 || a.B > 65535) return false;
    for (int j = 0; j < nz; ++j) {
        const int ks = a.p[j].ks, d = a.p[j].dil;
        if (ks != 3 && ks != 7 && ks != 11) return false;                          // the MFMA loops are unrolled for the V1 MRF
        if (d < 1 || (ks - 1) * d > kMrfSpanMax) return false;
        if (packed_conv1d_floats(a.C, a.C, ks) * 4u >= 0x7fffffffull) return false;
    }
    return IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR", IRIS_PAIR_F32_DEFAULT) != 0;
}

inline hipError_t launch_pair_f32(PairLaunchF32& a, int nz, hipStream_t stream) {
    if (a.C != 32 && a.C != 64) return hipErrorInvalidValue;
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nz; ++i)
            if (a.p[j].x == a.p[i].y) return hipErrorInvalidValue;               // never in place
    a.nz = nz;
    a.Gp = packed_groups(a.C);
    a.n_ct = packed_cotiles(a.C);
    int span = 0;
    for (int j = 0; j < nz; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
    }
    const PairPlanF32 pl = pair_f32_plan(a, nz);
    const PairTileF32 t = pl.tile;
    const long long n_jobs = pl.tiles * nz;
    if (n_jobs > 0x3fffffffLL) return hipErrorInvalidValue;
    a.n_jobs = (int)n_jobs;
    a.jobs_per_xcd = (int)((n_jobs + 7) / 8);
    a.z_major = IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_ZMAJOR", 1);     // (tile-major: 10-20 % slower once a launch is more than one round)
    const size_t lds_bytes = (size_t)(t.M + span) * (a.C + 4) * sizeof(float);
    dim3 grid((unsigned)(a.jobs_per_xcd * 8), (unsigned)a.B, 1u), block(256);
#define IRIS_PAIR_F32_CASE(WT_, WC_, MT_, C_, MINB_)                                                         \
    if (a.C == C_ && t.WT == WT_ && t.MT == MT_) {                                                           \
        auto kfn = mrf_pair_f32_kernel<WT_, WC_, MT_, C_, MINB_>;                                            \
        { const hipError_t e__ = ::iris::launch_kernel_named("mrf_pair_f32_kernel", kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
        return hipSuccess;                                                                                   \
    }
    IRIS_PAIR_F32_CASE(4, 1, 1, 32, 4)
    IRIS_PAIR_F32_CASE(2, 2, 2, 64, 3)
    IRIS_PAIR_F32_CASE(2, 2, 1, 64, 4)
#undef IRIS_PAIR_F32_CASE
    return hipErrorInvalidValue;
}

}  // namespace iris
