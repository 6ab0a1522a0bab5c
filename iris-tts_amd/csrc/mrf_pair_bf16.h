// mrf_pair_bf16.h -- one ResBlock conv PAIR of all MRF branches in one launch, bf16 storage (gfx950).
//
// Reference semantics: one iteration of ResBlock.forward's loop (src/iris/hifigan_pretrained.py:64-71)
//     xt = Conv1d_{k, dil d}(LeakyReLU(x));  y = Conv1d_{k, dil 1}(LeakyReLU(xt)) + x
// for the branches k = 3 / 7 / 11 of a stage (hifigan_pretrained.py:130-136), with the rounding points of the
// bf16-storage variant (conv_mfma_bf16.h; its CPU restatement is the test suite's generator_forward_bf16): LeakyReLU(x) -> bf16,
// xt = bf16(acc + bias), LeakyReLU(xt) -> bf16, y = bf16(acc + bias + x).  The numbers are bit for bit those of the
// two separate launches of conv_mfma_bf16_kernel (same MFMA order, same roundings) -- what changes is the traffic:
// xt never leaves the CU.  Unfused, a pair costs five tensor passes over HBM (conv1: read x, write xt; conv2: read
// xt, read x, write y); fused it costs two (read x window, write y): the residual pieces of the block's rows are requested
// right behind the window's LDS write, while the window load's lines are still in the L2 (round 4; requested in the epilogue,
// tens of microseconds later, they came from HBM a second time: 2.0x the compulsory reads).  At bf16 the C = 32 / 64 stages are
// HBM-bound (86 / 172 FLOP/B against a machine balance of ~312), so bytes are what their time is made of.
//
// Work split.  A 256-thread block owns M = WT*MT*32 consecutive rows of xt and ALL C channels (WC*NT*32 == C), for
// one branch of one batch item:
//   1. stage x rows [o0 - h2 - h1, o0 - h2 - h1 + M + (k-1)d) into LDS as bf16(LeakyReLU(x))   (h1 = d(k-1)/2, h2 = (k-1)/2)
//   2. conv1 on the matrix cores: xt rows [o0 - h2, o0 - h2 + M), fp32 accumulators
//   3. xt -> bf16(LeakyReLU(bf16(acc + bias1))), rows outside [0, L) forced to zero (conv2's zero padding), written
//      into the SAME LDS region (the x window is dead by then)
//   4. conv2 on the matrix cores from that window: rows [o0, o0 + M); only the first T_OUT = M - (k-1) are valid
//      (the last k-1 would need xt rows this block does not have) -- blocks advance by T_OUT rows
//   5. epilogue as in conv_mfma_bf16.h (per-wave LDS transpose, 16-byte stores; the residual pieces are in registers since step 1).
// The (k-1)/M rows of conv2 that are computed and thrown away are the price of keeping xt on chip: 0.5-2.6 % at
// M = 384, 1-5 % at M = 192-256.
//
// Block -> job mapping: job = (tile, branch) with the branch fastest (the branches of the first pair of a stage all
// read the same x); jobs are dealt to the XCDs in contiguous ranges (blockIdx.x % 8 = XCD), so the halo rows that
// neighbouring tiles share and the windows that sibling branches share are re-read from that XCD's L2.
#pragma once
#include <stdio.h>
#include "conv_mfma_bf16.h"
#ifndef IRIS_B16_RES_EARLY
#define IRIS_B16_RES_EARLY 1     // fused pairs: residual pieces of all m-tiles requested behind the window's LDS write (A/B builds: 0 = in the epilogue)
#endif

#ifndef IRIS_B16_RING_TAIL_GUARD
#define IRIS_B16_RING_TAIL_GUARD 1       // (A/B builds: 0 = the ring's last trip runs all its groups, zero fragments included)
#endif
#ifndef IRIS_B16_PAIR_SUM_DEFAULT
#define IRIS_B16_PAIR_SUM_DEFAULT 1      // the stage's last pair on the summing kernel (A/B builds: 0)
#endif
#ifndef IRIS_B16_SUM32_MT
#define IRIS_B16_SUM32_MT 3              // summing kernel, C = 32: 4 x MT x 32 rows per block
#endif
#ifndef IRIS_B16_SUM64_MT
#define IRIS_B16_SUM64_MT 3              // C = 64: 2 x MT x 32 rows
#endif
#ifndef IRIS_B16_SUM32_MINB
#define IRIS_B16_SUM32_MINB 2            // blocks per CU the summing kernel is compiled for
#endif
#ifndef IRIS_B16_SUM64_MINB
#define IRIS_B16_SUM64_MINB 2
#endif

namespace iris {
namespace b16 {

struct PairProblem {
    const uint16_t* x;    // bf16 [B, L, C]: input of the pair and its residual
    const void* w1;       // packed bf16 fragments of convs1[m] (pack_conv1d_bf16)
    const void* w2;       // ... of convs2[m]
    const float* b1;      // fp32 [C]
    const float* b2;
    uint16_t* y;          // bf16 [B, L, C]
    int ks;               // taps of both convs
    int dil;              // dilation of conv1 (conv2: 1)
};

struct PairLaunch {
    PairProblem p[kMaxGroup];
    int B, L, C;
    float slope;
    int nz;               // branches
    int Qp, n_ct;         // packed-weight geometry (packed_qsteps / packed_cotiles of C)
    int n_jobs;           // tiles x branches (tiles = ceil(L / smallest T_OUT))
    int jobs_per_xcd;     // ceil(n_jobs / 8)
    int bias_off;         // persistent kernel: byte offset of the bias table in LDS
    void* sum_y;          // summing kernel: the ONE output, [B, L, C]: bf16(LeakyReLU(mean_j y_j)) -- the operand the next
                          // ConvTranspose1d stages -- or, sum_f32 != 0, the fp32 mean itself (conv_post's input)
    int sum_f32;
    float inv_n;          // fp32(1 / nz)
    int ablate;           // diagnostics only: 1 no staging loads, 2 no MFMA loops, 4 no stores, 8 no residual loads
    unsigned long long* dbg;  // diagnostics only (stamp builds): per-segment cycle sums, else nullptr
};

constexpr int kPairSpanMax = 50;   // (k-1)*d of the widest supported conv1: k = 11, d = 5

// LeakyReLU for 0 <= slope <= 1 (pair_applicable checks it): max(v, slope*v), two VALU ops instead of three
__device__ __forceinline__ float lrelu_max(float v, float slope) { return fmaxf(v, v * slope); }

// Weight-fragment ring of the MFMA loop (see mma_chunk in conv_mfma_bf16.h), split so that the first D groups can be
// requested long before the loop starts: vmcnt retires in order, so requests issued ahead of a phase's HBM loads (or
// ahead of a barrier) have landed by the time the loop wants them.
template <int NT, int CIC, int D>
__device__ __forceinline__ void ring_request(u32x4 (&wv)[D][NT], __amdgpu_buffer_rsrc_t wr, unsigned wvoff,
                                             unsigned q_bytes, unsigned tap_bytes) {
    constexpr int QPC = CIC / 16;
    constexpr int QL = QPC == 8 ? 3 : (QPC == 4 ? 2 : (QPC == 2 ? 1 : 0));
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wv[i][nt] = buf_load4(wr, wvoff + (unsigned)nt * 1024u, (unsigned)(i >> QL) * tap_bytes + (unsigned)(i & (QPC - 1)) * q_bytes);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MT, int NT, int CIC, int D>
__device__ __forceinline__ void ring_mma_loop(f32x16 (&acc)[MT][NT], u32x4 (&wv)[D][NT], const char* a_lane, int dil_bytes,
                                              __amdgpu_buffer_rsrc_t wr, unsigned wvoff, unsigned q_bytes,
                                              unsigned tap_bytes, int ks) {
    constexpr int QPC = CIC / 16;
    constexpr int QL = QPC == 8 ? 3 : (QPC == 4 ? 2 : (QPC == 2 ? 1 : 0));
    constexpr int SB = CIC * 2 + 16;
    const int NG = ks * QPC;
    auto w_soff = [&](int n) -> unsigned { return (unsigned)(n >> QL) * tap_bytes + (unsigned)(n & (QPC - 1)) * q_bytes; };
    auto load_a = [&](u32x4 (&av)[MT], int n) {
        int tap = n >> QL;
        tap = tap < ks ? tap : ks - 1;
        const char* ap = a_lane + tap * dil_bytes + (n & (QPC - 1)) * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const u32x4*>(ap + m * 32 * SB);
    };
    u32x4 av[2][MT];
    load_a(av[0], 0);
    auto group = [&](int i, int n) {
        load_a(av[(i + 1) & 1], n + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, wv[i][nt]), __builtin_bit_cast(bf16x8, av[i & 1][m]), acc[m][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wv[i][nt] = buf_load4(wr, wvoff + (unsigned)nt * 1024u, w_soff(n + D));   // past the last tap: out of range, zeros
    };
    // C = 32 (QPC = 2 groups per tap, D = 4, an odd number of taps): the ring's last trip is half empty.  Run whole, its second
    // half multiplies zero weight fragments (6 / 14 / 22 groups rounded up to 8 / 16 / 24: a seventh of a conv's MFMAs); it is
    // issued as a tail of QPC groups instead.
    constexpr bool kTail = IRIS_B16_RING_TAIL_GUARD && (QPC % D) != 0;
    const int NG_main = kTail ? (NG / D) * D : NG;
    int n0 = 0;
    for (; n0 < NG_main; n0 += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) group(i, n0 + i);
    }
    if constexpr (kTail) {
        if (n0 < NG) {
#pragma unroll
            for (int i = 0; i < QPC; ++i) group(i, n0 + i);
        }
    }
}

template <int WT, int WC, int MT, int NT, int C, int MINB>
__global__ void __launch_bounds__(256, MINB) mrf_pair_bf16_kernel(const PairLaunch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_pair[];
    char* lds = lds_pair;
    static_assert(WT * WC == 4 && WC * NT * 32 == C, "a block owns all C channels");
    constexpr int SB = C * 2 + 16;
    constexpr int M = WT * MT * 32;
    constexpr int PPR = C / 8;                                           // 16-byte pieces per window row
    constexpr int NQ = ((M + kPairSpanMax) * PPR + 255) / 256;          // staged pieces per thread: ALL in flight at once
    constexpr int D = 4;                                                 // weight ring depth (groups)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;
#ifdef IRIS_PAIR_STAMPS
    // diagnostic build only: per-wave cycle totals of [0] requests + window wait + LDS write, [1] barrier, [2] conv1 loop,
    // [3] barrier + xt write + barrier, [4] conv2 loop, [5] barrier + epilogue, [6] whole block; [7] waves counted
    unsigned long long seg_t[8];
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
#define PAIR_STAMP(i) seg_t[i] = stamp()
#else
#define PAIR_STAMP(i)
#endif
    PAIR_STAMP(0);

    // job -> (tile, branch): contiguous job ranges per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int job = xcd * a.jobs_per_xcd + slot;
    if (job >= a.n_jobs) return;
    const int tile = job / a.nz, zr = job - tile * a.nz;
    const int z = a.nz - 1 - zr;                      // heaviest branch first
    PairProblem p = a.p[0];
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
    const int ct0 = wc * NT;                          // this wave's first 32-wide channel tile

    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 2u;
    const size_t item = (size_t)b * L * C;
    const unsigned q_bytes = (unsigned)a.n_ct * 1024u;
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;
    const unsigned wvoff = (unsigned)ct0 * 1024u + (unsigned)lane * 16u;
    const char* a_lane = lds + (wt * MT * 32 + lo) * SB + hi * 16;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + item, tensor_bytes);
    const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(p.w1, (unsigned)ks * tap_bytes);
    const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(p.w2, (unsigned)ks * tap_bytes);

    // The accumulators START at the bias (the MFMA's C input): no bias registers to keep, no add in the epilogues.
    // conv_mfma_bf16_kernel does the same, so fused and separate launches still produce identical bits.
    f32x16 acc[MT][NT];
    auto load_bias = [&](f32x4 (&bias4)[NT][4], const float* bptr) {   // (nt, g) -> channels (ct0+nt)*32 + 8g + 4hi + {0..3}
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[nt][g] = *reinterpret_cast<const f32x4*>(bptr + (ct0 + nt) * 32 + 8 * g + 4 * hi);
    };
    auto init_acc = [&](const f32x4 (&bias4)[NT][4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nt][r] = bias4[nt][r >> 2][r & 3];
    };

    // epilogue geometry (IRIS_B16_RES_EARLY: the residual pieces of ALL m-tiles are requested right behind the window, while its lines
    // are still in the L2 -- the epilogue's re-read of x otherwise comes from HBM again: 2.0x the compulsory reads, profiles/r03zz_bf16_c3_hbm_traffic.json)
    constexpr int RS = NT * 32 * 4 + 16;               // scratch row stride (bytes) = 16 * odd
    constexpr int PPRO = NT * 4;                       // 16-byte bf16 pieces per row of this wave's channel span
    constexpr int NP = 2 * NT;                         // pieces per lane and m-tile
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + item, tensor_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.x + item, (a.ablate & 8) ? 0u : tensor_bytes);
    unsigned pvoff[MT][NP];
    int pscr[NP];
    auto geometry = [&](int lane_) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = j * 64 + lane_;
            const int row_l = q / PPRO, pc = q - row_l * PPRO;
            pscr[j] = row_l * RS + pc * 32;
            const int co = ct0 * 32 + 8 * pc;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int im = (wt * MT + m) * 32 + row_l;
                const int o = o0 + im;
                pvoff[m][j] = (im < T_OUT && o < L) ? (unsigned)(o * C + co) * 2u : kOob;
            }
        }
    };
    constexpr int RBUF = IRIS_B16_RES_EARLY ? MT : 2;
    u32x4 resv[RBUF][NP];
    if constexpr (IRIS_B16_RES_EARLY) geometry(lane);   // (for the early requests; computed again in front of the epilogue, see there)
    // ---- 1. requests, oldest first: bias1, conv1's first weight fragments (L2), then the whole x window (HBM) ----------
    f32x4 bias4[NT][4];
    load_bias(bias4, p.b1);
    u32x4 wv[D][NT];
    ring_request<NT, C, D>(wv, wr1, wvoff, q_bytes, tap_bytes);
    {
        const int in_row0 = o0 - h2 - h1, R = M + (ks - 1) * dil, total = R * PPR;
        u32x4 v[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            const int r = idx / PPR, pc = idx & (PPR - 1);
            const int row = in_row0 + r;
            const bool ok = idx < total && row >= 0 && row < L && !(a.ablate & 1);
            v[u] = buf_load4(xr, ok ? (unsigned)(row * C + 8 * pc) * 2u : kOob, 0);
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            const int r = idx / PPR, pc = idx & (PPR - 1);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned w = v[u][e];
                o[e] = pack_bf2(lrelu_max(bf_lo(w), a.slope), lrelu_max(bf_hi(w), a.slope));
            }
            if (idx < total) *reinterpret_cast<u32x4*>(lds + r * SB + pc * 16) = o;
        }
    }
    if constexpr (IRIS_B16_RES_EARLY) {
        // the residual pieces of every m-tile, behind the window's LDS write: their lines are still in the L2 (the window load brought
        // them in a few microseconds ago), and the barrier below covers their latency
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NP; ++j) resv[m][j] = buf_load4(rr, pvoff[m][j], 0);
    }
    PAIR_STAMP(1);
    __syncthreads();
    PAIR_STAMP(2);
    // ---- 2. conv1 ---------------------------------------------------------------------------------------
    init_acc(bias4);
    if (!(a.ablate & 2)) ring_mma_loop<MT, NT, C, D>(acc, wv, a_lane, dil * SB, wr1, wvoff, q_bytes, tap_bytes, ks);
    ring_request<NT, C, D>(wv, wr2, wvoff, q_bytes, tap_bytes);        // conv2's first fragments travel during steps 3
    load_bias(bias4, p.b2);
    PAIR_STAMP(3);
    __syncthreads();                                                    // every wave is done with the x window
    // ---- 3. xt -> LDS: bf16(LeakyReLU(bf16(acc))) (acc already holds the bias), zero outside [0, L) --------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row_l = (wt * MT + m) * 32 + lo;
        const int row_g = o0 - h2 + row_l;
        const unsigned keep = (row_g >= 0 && row_g < L) ? 0xffffffffu : 0u;   // (a mask, not a branch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const unsigned t = pack_bf2(acc[m][nt][4 * g + 2 * e2], acc[m][nt][4 * g + 2 * e2 + 1]);   // the stored xt
                    o[e2] = pack_bf2(lrelu_max(bf_lo(t), a.slope), lrelu_max(bf_hi(t), a.slope)) & keep;       // conv2's operand
                }
                *reinterpret_cast<u32x2*>(lds + row_l * SB + ((ct0 + nt) * 32 + 8 * g + 4 * hi) * 2) = o;
            }
    }
    {
        // (IRIS_B16_RES_EARLY: the offsets are computed a second time, from a lane index the compiler cannot see through, so that
        //  they do not occupy eight registers across both MFMA loops)
        int lane2 = lane;
        if constexpr (IRIS_B16_RES_EARLY) asm volatile("" : "+v"(lane2));
        geometry(lane2);
    }
    if constexpr (!IRIS_B16_RES_EARLY) {             // the residual of the first m-tile: requested now, back long before the epilogue
#pragma unroll
        for (int j = 0; j < NP; ++j) resv[0][j] = buf_load4(rr, pvoff[0][j], 0);
    }
    __syncthreads();
    PAIR_STAMP(4);
    // ---- 4. conv2 (dilation 1; rows M .. M+k-2 of the window hold stale bytes: they only reach outputs >= T_OUT) ----
    init_acc(bias4);
    if (!(a.ablate & 2)) ring_mma_loop<MT, NT, C, D>(acc, wv, a_lane, SB, wr2, wvoff, q_bytes, tap_bytes, ks);
    PAIR_STAMP(5);
    __syncthreads();                                   // the epilogue scratch aliases the window
    // ---- 5. epilogue: + x, one rounding to bf16.  Each wave turns its 32-row m-tiles through a private LDS scratch so that
    // residual loads and stores are 16 bytes per lane and whole 64-byte (NT = 1) / 128-byte row segments per 4 / 8 lanes.
    // (Measured against an LDS-free epilogue that builds 16-byte pieces with v_permlane32_swap: its stores are 32-byte
    // segments per row, and it lost 3.5 % at C = 128 and 7 % at C = 64 -- profiles/r02_notes.md.)
    char* scr = lds + wave * (32 * RS);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!IRIS_B16_RES_EARLY && m + 1 < MT) {
#pragma unroll
            for (int j = 0; j < NP; ++j) resv[(m + 1) & 1][j] = buf_load4(rr, pvoff[m + 1][j], 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[m][nt][4 * g + e];
                *reinterpret_cast<f32x4*>(scr + lo * RS + (nt * 32 + 8 * g + 4 * hi) * 4) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4 outp[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(scr + pscr[j]);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(scr + pscr[j] + 16);
            const u32x4 rv = resv[IRIS_B16_RES_EARLY ? m : (m & 1)][j];
            outp[j][0] = pack_bf2(lo4[0] + bf_lo(rv[0]), lo4[1] + bf_hi(rv[0]));
            outp[j][1] = pack_bf2(lo4[2] + bf_lo(rv[1]), lo4[3] + bf_hi(rv[1]));
            outp[j][2] = pack_bf2(hi4[0] + bf_lo(rv[2]), hi4[1] + bf_hi(rv[2]));
            outp[j][3] = pack_bf2(hi4[2] + bf_lo(rv[3]), hi4[3] + bf_hi(rv[3]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)((a.ablate & 4) ? kOob : pvoff[m][j]), 0, 0);
        asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 stores (see mrf_conv_mfma_f32.h)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef IRIS_PAIR_STAMPS
    PAIR_STAMP(6);
    if (lane == 0 && a.dbg) {
        for (int i = 0; i < 6; ++i) atomicAdd(a.dbg + i, seg_t[i + 1] - seg_t[i]);
        atomicAdd(a.dbg + 6, seg_t[6] - seg_t[0]);
        atomicAdd(a.dbg + 7, 1ull);
    }
#endif
#undef PAIR_STAMP
}

// ---- the stage's LAST pair, summing: a block runs the pair of ALL THREE branches on its rows and stores only the MRF mean ----
// hifigan_pretrained.py:131-137: xs = rb0(x); xs += rb1(x); xs += rb2(x); x = xs / 3, followed by LeakyReLU at the head of the
// next layer.  With separate (tile, branch) blocks the three y_j go to HBM and the next ConvTranspose1d (or conv_post) reads
// all three -- at bf16 those layers are pure traffic (configs[2]: 0.97 + 0.17 ms of a 10.7 ms step).  Here a block keeps
// sum_j float(y_j) in registers (y_j rounded to bf16 exactly as the separate kernel stores it, added in the order 0, 1, 2)
// and stores  bf16(LeakyReLU(((y0 + y1) + y2 + 0) * inv_n))  -- bit for bit the operand stage_window() builds from the three
// tensors (conv_mfma_bf16.h) -- so the consumer reads ONE tensor and applies no activation; for the last stage it stores the
// fp32 mean instead, which conv_post takes as it takes the fp32 path's (conv_post.h, n_in = 1).  Writes drop from three
// tensors to one, the consumer's reads from three to one.
// All branches use the rows [o0, o0 + T_OUT) with T_OUT = M - (k_max - 1): the smaller kernels throw away a few more rows.
// The price: accumulators AND the sum live at once (2 x MT x NT x 16 registers), so tiles are smaller or blocks fewer than in
// the kernel above, and a block is three pairs long.
template <int WT, int WC, int MT, int NT, int C, int MINB>
__global__ void __launch_bounds__(256, MINB) mrf_pair_bf16_sum_kernel(const PairLaunch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_pair[];
    char* lds = lds_pair;
    static_assert(WT * WC == 4 && WC * NT * 32 == C, "a block owns all C channels");
    constexpr int SB = C * 2 + 16;
    constexpr int M = WT * MT * 32;
    constexpr int PPR = C / 8;
    constexpr int NQ = ((M + kPairSpanMax) * PPR + 255) / 256;
    constexpr int D = 4;
    constexpr int NZ = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * a.jobs_per_xcd + slot;             // (n_jobs = tiles here)
    if (tile >= a.n_jobs) return;
    int kmax = a.p[0].ks;
    if (a.p[1].ks > kmax) kmax = a.p[1].ks;
    if (a.p[2].ks > kmax) kmax = a.p[2].ks;
    const int T_OUT = M - (kmax - 1);
    const int o0 = tile * T_OUT;
    const int L = a.L;
    if (o0 >= L) return;
    const int b = blockIdx.y;
    const int ct0 = wc * NT;
    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 2u;
    const size_t item = (size_t)b * L * C;
    const unsigned q_bytes = (unsigned)a.n_ct * 1024u;
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;
    const unsigned wvoff = (unsigned)ct0 * 1024u + (unsigned)lane * 16u;
    const char* a_lane = lds + (wt * MT * 32 + lo) * SB + hi * 16;

    // epilogue geometry (the same for the three branches)
    constexpr int RS = NT * 32 * 4 + 16;
    constexpr int PPRO = NT * 4;
    constexpr int NP = 2 * NT;
    unsigned pvoff[MT][NP];
    int pscr[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = j * 64 + lane;
        const int row_l = q / PPRO, pc = q - row_l * PPRO;
        pscr[j] = row_l * RS + pc * 32;
        const int co = ct0 * 32 + 8 * pc;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int im = (wt * MT + m) * 32 + row_l;
            const int o = o0 + im;
            pvoff[m][j] = (im < T_OUT && o < L) ? (unsigned)(o * C + co) * 2u : kOob;
        }
    }
    float sum[MT][NP][8];
    f32x16 acc[MT][NT];
    auto load_bias = [&](f32x4 (&bias4)[NT][4], const float* bptr) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[nt][g] = *reinterpret_cast<const f32x4*>(bptr + (ct0 + nt) * 32 + 8 * g + 4 * hi);
    };
    auto init_acc = [&](const f32x4 (&bias4)[NT][4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nt][r] = bias4[nt][r >> 2][r & 3];
    };
    char* scr = lds + wave * (32 * RS);

#pragma unroll
    for (int z = 0; z < NZ; ++z) {
        // (a distinct marker per branch keeps hipcc from merging the three bodies' heads across the unrolled loop)
        asm volatile("; bf16 summing pair, branch %0" :: "n"(z) : "memory");
        const PairProblem& p = a.p[z];                     // constant index: stays in the kernel-argument segment
        const int ks = p.ks, dil = p.dil;
        const int h2 = (ks - 1) / 2, h1 = dil * (ks - 1) / 2;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + item, tensor_bytes);
        const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(p.w1, (unsigned)ks * tap_bytes);
        const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(p.w2, (unsigned)ks * tap_bytes);
        if (z > 0) __syncthreads();                        // the previous branch's epilogue scratch aliases the window
        // ---- 1. bias1, conv1's first weight fragments, the x window ----
        f32x4 bias4[NT][4];
        load_bias(bias4, p.b1);
        u32x4 wv[D][NT];
        ring_request<NT, C, D>(wv, wr1, wvoff, q_bytes, tap_bytes);
        {
            const int in_row0 = o0 - h2 - h1, R = M + (ks - 1) * dil, total = R * PPR;
            u32x4 v[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int idx = u * 256 + (int)threadIdx.x;
                const int r = idx / PPR, pc = idx & (PPR - 1);
                const int row = in_row0 + r;
                const bool ok = idx < total && row >= 0 && row < L;
                v[u] = buf_load4(xr, ok ? (unsigned)(row * C + 8 * pc) * 2u : kOob, 0);
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int idx = u * 256 + (int)threadIdx.x;
                const int r = idx / PPR, pc = idx & (PPR - 1);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned w = v[u][e];
                    o[e] = pack_bf2(lrelu_max(bf_lo(w), a.slope), lrelu_max(bf_hi(w), a.slope));
                }
                if (idx < total) *reinterpret_cast<u32x4*>(lds + r * SB + pc * 16) = o;
            }
        }
        __syncthreads();
        // ---- 2. conv1 ----
        init_acc(bias4);
        ring_mma_loop<MT, NT, C, D>(acc, wv, a_lane, dil * SB, wr1, wvoff, q_bytes, tap_bytes, ks);
        ring_request<NT, C, D>(wv, wr2, wvoff, q_bytes, tap_bytes);
        load_bias(bias4, p.b2);
        __syncthreads();
        // ---- 3. xt -> LDS ----
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row_l = (wt * MT + m) * 32 + lo;
            const int row_g = o0 - h2 + row_l;
            const unsigned keep = (row_g >= 0 && row_g < L) ? 0xffffffffu : 0u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o;
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const unsigned t = pack_bf2(acc[m][nt][4 * g + 2 * e2], acc[m][nt][4 * g + 2 * e2 + 1]);
                        o[e2] = pack_bf2(lrelu_max(bf_lo(t), a.slope), lrelu_max(bf_hi(t), a.slope)) & keep;
                    }
                    *reinterpret_cast<u32x2*>(lds + row_l * SB + ((ct0 + nt) * 32 + 8 * g + 4 * hi) * 2) = o;
                }
        }
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.x + item, tensor_bytes);
        constexpr int RB = (MT * NT > 3) ? 1 : 2;         // residual buffers: one when registers are short (accumulators + sum)
        u32x4 resv[RB][NP];
        if (RB == 2) {
#pragma unroll
            for (int j = 0; j < NP; ++j) resv[0][j] = buf_load4(rr, pvoff[0][j], 0);
        }
        __syncthreads();
        // ---- 4. conv2 ----
        init_acc(bias4);
        ring_mma_loop<MT, NT, C, D>(acc, wv, a_lane, SB, wr2, wvoff, q_bytes, tap_bytes, ks);
        __syncthreads();
        // ---- 5. epilogue: y_j = bf16(acc + x_j) as the separate kernel stores it; sum += float(y_j) ----
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (RB == 2) {
                if (m + 1 < MT) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) resv[(m + 1) % RB][j] = buf_load4(rr, pvoff[m + 1][j], 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NP; ++j) resv[0][j] = buf_load4(rr, pvoff[m][j], 0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[m][nt][4 * g + e];
                    *reinterpret_cast<f32x4*>(scr + lo * RS + (nt * 32 + 8 * g + 4 * hi) * 4) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(scr + pscr[j]);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(scr + pscr[j] + 16);
                const u32x4 rv = resv[m % RB][j];
                unsigned yp[4];
                yp[0] = pack_bf2(lo4[0] + bf_lo(rv[0]), lo4[1] + bf_hi(rv[0]));
                yp[1] = pack_bf2(lo4[2] + bf_lo(rv[1]), lo4[3] + bf_hi(rv[1]));
                yp[2] = pack_bf2(hi4[0] + bf_lo(rv[2]), hi4[1] + bf_hi(rv[2]));
                yp[3] = pack_bf2(hi4[2] + bf_lo(rv[3]), hi4[3] + bf_hi(rv[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (z == 0) { sum[m][j][2 * e] = bf_lo(yp[e]); sum[m][j][2 * e + 1] = bf_hi(yp[e]); }
                    else        { sum[m][j][2 * e] = sum[m][j][2 * e] + bf_lo(yp[e]); sum[m][j][2 * e + 1] = sum[m][j][2 * e + 1] + bf_hi(yp[e]); }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- the mean.  "+ 0.f": the consumer's three-tensor staging adds a fourth, absent input (reads as 0) ----
    if (a.sum_f32) {
        const __amdgpu_buffer_rsrc_t yr = make_rsrc((float*)a.sum_y + item, tensor_bytes * 2u);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                u32x4 o0v, o1v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0v[e] = __builtin_bit_cast(unsigned, (sum[m][j][e] + 0.f) * a.inv_n);
                    o1v[e] = __builtin_bit_cast(unsigned, (sum[m][j][4 + e] + 0.f) * a.inv_n);
                }
                const unsigned vo = pvoff[m][j] == kOob ? kOob : pvoff[m][j] * 2u;
                __builtin_amdgcn_raw_buffer_store_b128(o0v, yr, (int)vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1v, yr, (int)vo, 16, 0);
            }
    } else {
        const __amdgpu_buffer_rsrc_t yr = make_rsrc((uint16_t*)a.sum_y + item, tensor_bytes);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf2(lrelu1((sum[m][j][2 * e] + 0.f) * a.inv_n, a.slope), lrelu1((sum[m][j][2 * e + 1] + 0.f) * a.inv_n, a.slope));
                __builtin_amdgcn_raw_buffer_store_b128(o, yr, (int)pvoff[m][j], 0, 0);
            }
    }
}

// ---- launch ----------------------------------------------------------------------------------------
struct PairTile { int WT, WC, MT, NT, MINB, M; };

inline bool pair_tile_for(int C, PairTile* t) {
    const int v64 = IRIS_DIAG_ENV("IRIS_B16_PAIR64", 0);
    if (C == 32) {                                                     // (512 rows at three blocks per CU: no difference)
        *t = IRIS_DIAG_ENV("IRIS_B16_PAIR32", 0) == 1 ? PairTile{4, 1, 3, 1, 3, 384} : PairTile{4, 1, 3, 1, 4, 384};   // (A/B: three blocks per CU)
        return true;
    }
    if (C == 64) {
        // 256 rows as 2 x 2 waves of 128 x 32 at three blocks per CU: 2 % ahead of 192 rows at four (2.09 vs 2.13 ms)
        if (v64 == 1) { *t = PairTile{4, 1, 2, 2, 3, 256}; return true; }
        if (v64 == 2) { *t = PairTile{2, 2, 3, 1, 4, 192}; return true; }
        if (v64 == 3) { *t = PairTile{2, 2, 4, 1, 2, 256}; return true; }   // (A/B: the default tile at two blocks per CU)
        *t = PairTile{2, 2, 4, 1, 3, 256};
        return true;
    }
    // C = 128: 192 rows at two blocks per CU beat 128 rows at three (stage 1 of configs[2]: 3.75 vs 3.95 ms): a weight
    // fragment (1 KB per wave, from L2) feeds three MFMAs per channel tile instead of two, and 10 of 192 conv2 rows are
    // thrown away instead of 10 of 128
    if (C == 128 && IRIS_DIAG_ENV("IRIS_B16_PAIR128", 1) == 2) { *t = PairTile{2, 2, 2, 2, 3, 128}; return true; }
    if (C == 128 && IRIS_DIAG_ENV("IRIS_B16_PAIR128", 1)) { *t = PairTile{2, 2, 3, 2, 2, 192}; return true; }
    return false;
}

// True when the pair launch `a` (nz branches, a.C channels) can take the fused kernel.
inline bool pair_applicable(const PairLaunch& a, int nz) {
    PairTile t;
    if (nz < 1 || nz > kMaxGroup || !pair_tile_for(a.C, &t)) return false;
    if (!(a.slope >= 0.f && a.slope <= 1.f)) return false;                        // LeakyReLU is evaluated as max(v, slope*v)
    if ((double)a.L * a.C * 2.0 >= 2147483648.0) return false;
    for (int j = 0; j < nz; ++j) {
        const int ks = a.p[j].ks, d = a.p[j].dil;
        if (ks < 1 || !(ks & 1) || d < 1) return false;
        if (ks - 1 >= t.M / 2) return false;                                     // keeps T_OUT >= M / 2
        if ((ks - 1) * d > kPairSpanMax) return false;                           // the kernel's staging registers are sized for this
        if ((size_t)(t.M + (ks - 1) * d) * (a.C * 2 + 16) > 66 * 1024) return false; // window must leave room for two blocks per CU
    }
    return IRIS_DIAG_ENV("IRIS_B16_PAIR", 1) != 0;
}

inline hipError_t launch_pair_bf16(PairLaunch& a, int nz, hipStream_t stream) {
    PairTile t;
    if (!pair_tile_for(a.C, &t)) return hipErrorInvalidValue;
    // never in place: a block's input window overlaps the rows its neighbours write
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nz; ++i)
            if (a.p[j].x == a.p[i].y) return hipErrorInvalidValue;
    a.nz = nz;
    a.Qp = packed_qsteps(a.C);
    a.n_ct = packed_cotiles(a.C);
    a.ablate = IRIS_DIAG_ENV("IRIS_B16_ABLATE", 0);
    a.dbg = nullptr;
#ifdef IRIS_PAIR_STAMPS
    static unsigned long long* dbg_dev = nullptr;
    if (!dbg_dev && hipMalloc(&dbg_dev, 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(dbg_dev, 0, 8 * sizeof(unsigned long long), stream);
    a.dbg = dbg_dev;
    struct StampReport {
        unsigned long long* d; hipStream_t s; int C, L;
        ~StampReport() {
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const double n = (double)h[7], tot = (double)h[6];
            if (n > 0) fprintf(stderr, "[pair stamps] C=%d L=%d waves=%.0f cyc/wave=%.0f | window %.3f bar %.3f conv1 %.3f bar+xt+bar %.3f conv2 %.3f bar+epilogue %.3f\n",
                               C, L, n, tot / n, h[0] / tot, h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot);
        }
    } report{dbg_dev, stream, a.C, a.L};
#endif
    int span = 0, kmax = 1;
    for (int j = 0; j < nz; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
        if (a.p[j].ks > kmax) kmax = a.p[j].ks;
    }
    const int t_out_min = t.M - (kmax - 1);
    const long long tiles = (a.L + t_out_min - 1) / t_out_min;
    const long long n_jobs = tiles * nz;
    if (n_jobs > 0x3fffffffLL || a.B > 65535) return hipErrorInvalidValue;
    a.n_jobs = (int)n_jobs;
    a.jobs_per_xcd = (int)((n_jobs + 7) / 8);
    const size_t window_bytes = (size_t)(t.M + span) * (a.C * 2 + 16);
    const size_t scratch_bytes = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);      // epilogue transpose, aliases the window
    const size_t lds_bytes = window_bytes > scratch_bytes ? window_bytes : scratch_bytes;
    dim3 grid((unsigned)(a.jobs_per_xcd * 8), (unsigned)a.B, 1u), block(256);
#define IRIS_PAIR_CASE(WT_, WC_, MT_, NT_, C_, MINB_)                                                        \
    if (a.C == C_ && t.WT == WT_ && t.WC == WC_ && t.MT == MT_ && t.NT == NT_ && t.MINB == MINB_) {        \
        auto kfn = mrf_pair_bf16_kernel<WT_, WC_, MT_, NT_, C_, MINB_>;                                      \
        { const hipError_t e__ = ::iris::launch_kernel_named("mrf_pair_bf16_kernel", kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
        return hipSuccess;                                                                                   \
    }
    IRIS_PAIR_CASE(4, 1, 3, 1, 32, 4)
    IRIS_PAIR_CASE(4, 1, 3, 1, 32, 3)
    IRIS_PAIR_CASE(2, 2, 3, 1, 64, 4)
    IRIS_PAIR_CASE(4, 1, 2, 2, 64, 3)
    IRIS_PAIR_CASE(2, 2, 4, 1, 64, 3)
    IRIS_PAIR_CASE(2, 2, 4, 1, 64, 2)
    IRIS_PAIR_CASE(2, 2, 2, 2, 128, 3)
    IRIS_PAIR_CASE(2, 2, 3, 2, 128, 2)
#undef IRIS_PAIR_CASE
    return hipErrorInvalidValue;
}

// ---- the summing kernel's tiles and launch ----------------------------------------------------------------
// Accumulators and the sum are live together: 2 x MT x NT x 16 registers, so two blocks per CU.
inline bool pair_sum_tile_for(int C, PairTile* t) {
    if (C == 32)  { *t = PairTile{4, 1, IRIS_B16_SUM32_MT, 1, IRIS_B16_SUM32_MINB, 4 * IRIS_B16_SUM32_MT * 32}; return true; }
    if (C == 64)  { *t = PairTile{2, 2, IRIS_B16_SUM64_MT, 1, IRIS_B16_SUM64_MINB, 2 * IRIS_B16_SUM64_MT * 32}; return true; }
    // (C = 128 -- 128 rows x 128 channels: 64 + 64 accumulator and sum registers next to a 32-register weight ring -- spills ~95
    //  registers at two blocks per CU and is left to the three-tensor path)
    return false;
}

// True when the stage's last pair can run on the summing kernel: three branches of equal kernel size per pair, shapes the
// pair kernel takes, and -- for the fp32 mean -- an item below 2^31 bytes at four bytes per element.
inline bool pair_sum_applicable(const PairLaunch& a, int nz, bool f32_out) {
    PairTile t;
    if (nz != 3 || !pair_sum_tile_for(a.C, &t)) return false;
    if (!(a.slope >= 0.f && a.slope <= 1.f)) return false;
    if ((double)a.L * a.C * (f32_out ? 4.0 : 2.0) >= 2147483648.0) return false;
    for (int j = 0; j < nz; ++j) {
        const int ks = a.p[j].ks, d = a.p[j].dil;
        if (ks < 1 || !(ks & 1) || d < 1) return false;
        if (ks - 1 >= t.M / 2) return false;
        if ((ks - 1) * d > kPairSpanMax) return false;
        if ((size_t)(t.M + (ks - 1) * d) * (a.C * 2 + 16) > 66 * 1024) return false;
    }
    return IRIS_DIAG_ENV("IRIS_B16_PAIR_SUM", IRIS_B16_PAIR_SUM_DEFAULT) != 0;
}

inline hipError_t launch_pair_bf16_sum(PairLaunch& a, void* sum_y, bool f32_out, hipStream_t stream) {
    PairTile t;
    if (!pair_sum_tile_for(a.C, &t) || !sum_y) return hipErrorInvalidValue;
    for (int j = 0; j < 3; ++j)
        if ((const void*)a.p[j].x == sum_y) return hipErrorInvalidValue;     // never in place (and the fp32 mean spans two buffers: the caller checks)
    a.nz = 3;
    a.Qp = packed_qsteps(a.C);
    a.n_ct = packed_cotiles(a.C);
    a.ablate = 0;
    a.dbg = nullptr;
    a.sum_y = sum_y; a.sum_f32 = f32_out ? 1 : 0; a.inv_n = 1.0f / 3.0f;
    int span = 0, kmax = 1;
    for (int j = 0; j < 3; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
        if (a.p[j].ks > kmax) kmax = a.p[j].ks;
    }
    const int t_out = t.M - (kmax - 1);
    const long long tiles = (a.L + t_out - 1) / t_out;
    if (tiles > 0x3fffffffLL || a.B > 65535) return hipErrorInvalidValue;
    a.n_jobs = (int)tiles;
    a.jobs_per_xcd = (int)((tiles + 7) / 8);
    const size_t window_bytes = (size_t)(t.M + span) * (a.C * 2 + 16);
    const size_t scratch_bytes = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);
    const size_t lds_bytes = window_bytes > scratch_bytes ? window_bytes : scratch_bytes;
    dim3 grid((unsigned)(a.jobs_per_xcd * 8), (unsigned)a.B, 1u), block(256);
#define IRIS_PAIR_SUM_CASE(WT_, WC_, MT_, NT_, C_, MINB_)                                                    \
    if (a.C == C_ && t.WT == WT_ && t.WC == WC_ && t.MT == MT_ && t.NT == NT_ && t.MINB == MINB_) {        \
        auto kfn = mrf_pair_bf16_sum_kernel<WT_, WC_, MT_, NT_, C_, MINB_>;                                  \
        { const hipError_t e__ = ::iris::launch_kernel_named("mrf_pair_bf16_sum_kernel", kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
        return hipSuccess;                                                                                   \
    }
    IRIS_PAIR_SUM_CASE(4, 1, IRIS_B16_SUM32_MT, 1, 32, IRIS_B16_SUM32_MINB)
    IRIS_PAIR_SUM_CASE(2, 2, IRIS_B16_SUM64_MT, 1, 64, IRIS_B16_SUM64_MINB)
#undef IRIS_PAIR_SUM_CASE
    return hipErrorInvalidValue;
}

}  // namespace b16
}  // namespace iris
