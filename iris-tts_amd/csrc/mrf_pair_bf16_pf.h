// mrf_pair_bf16_pf.h -- the fused bf16 conv pair (mrf_pair_bf16.h) as PERSISTENT blocks that fetch the next job's
// window while the current job still computes.
//
// Reference semantics, rounding points and the per-job work split are those of mrf_pair_bf16_kernel (one iteration of
// ResBlock.forward's loop, src/iris/hifigan_pretrained.py:64-71, for the k = 3 / 7 / 11 branches of a stage,
// :130-136); every output element is the same MFMA chain, so the bits are those of the two separate launches.
//
// Why (profiles/r02_notes.md): a non-persistent block spends 24 % of its life waiting for its
// window, and with four such blocks per CU the chip has on average ONE window (~34 KB) in flight per CU -- by
// Little's law that is the 2.9 TB/s the C = 32 launches were measured at.  Here a block walks jobs i, i + G, ... of
// its XCD's contiguous job range and requests job i+1's window into registers during the LAST D groups of job i's
// conv2, so the loads fly during the epilogue and are written to LDS after it.
//
// vmcnt retires in order: a wait for a weight fragment also waits for every older load.  That fixes where the
// long-latency requests may sit -- only where no younger load is waited for before the window itself is needed:
//   * the weight ring of a conv is D groups deep, so its last D groups request nothing for that conv; those slots take
//     the first D fragment groups of the NEXT conv instead (conv1 -> this job's conv2, conv2 -> the next job's conv1);
//   * the same D groups of conv2 carry, oldest first, the residual pieces of the epilogue's m-tiles 1.. (m-tile 0 is
//     requested before conv2) and then the next window; the epilogue waits only for the residuals, which are older.
// The MFMA loops are unrolled per kernel size (3 / 7 / 11), which keeps every vmcnt the compiler derives exact and
// removes the groups the generic loop runs past the last tap (k*C/16 is not a multiple of D at C = 32: +14 % MFMAs).
//
// RESULT (round 3, profiles/r03_notes.md): bit-identical and SLOWER than the non-persistent kernel -- configs[2] stage 2
// (C = 64) 2.40-2.66 ms against 2.11, stage 3 (C = 32) 1.68 against 1.50.  The window held in registers (28-40 VGPRs)
// plus all residual pieces in flight cost a resident block per CU (C = 32: 4 -> 3; C = 64: 256-row tiles 3 -> 2, or
// 128-row tiles whose weight stream per MFMA doubles), and with 32-cycle MFMAs a wave's loop runs far below the pipe
// rate on its own: the lost wave per SIMD costs more than the hidden window wait returns.  Kept for the diagnostic
// build only (IRIS_B16_PAIR_PF=1); the release library does not contain it.
#pragma once
#include <type_traits>
#include "device_info.h"
#include "mrf_pair_bf16.h"

namespace iris {
namespace b16 {

// One conv over the LDS window: NG = KS * CIC/16 groups of MT*NT MFMAs.  Ring slot n % D holds group n's weight
// fragments on entry (groups 0..D-1) and is refilled behind group n's MFMAs with group n + D, or -- in the last D
// groups, when NEXT -- with group n % D of the conv that runs next (descriptor wr_next), so that it finds slot s = group s.  tail(t), t = 0..D-1, is
// called behind the refill of the last D groups.
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

template <int KS, int MT, int NT, int CIC, int D, bool NEXT, class Tail>
__device__ __forceinline__ void pf_mma(f32x16 (&acc)[MT][NT], u32x4 (&wv)[D][NT], const char* a_lane, int dil_bytes,
                                       __amdgpu_buffer_rsrc_t wr, __amdgpu_buffer_rsrc_t wr_next, unsigned wvoff,
                                       unsigned q_bytes, unsigned tap_bytes, Tail tail) {
    constexpr int QPC = CIC / 16;
    constexpr int NG = KS * QPC;
    constexpr int SB = CIC * 2 + 16;
    static_assert(NG >= D, "the ring must not be deeper than the conv");
    auto w_soff = [&](int g) -> unsigned { return (unsigned)(g / QPC) * tap_bytes + (unsigned)(g % QPC) * q_bytes; };
    u32x4 av[2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const u32x4*>(a_lane + m * 32 * SB);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        if (n + 1 < NG) {
            const char* ap = a_lane + ((n + 1) / QPC) * dil_bytes + ((n + 1) % QPC) * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) av[(n + 1) & 1][m] = *reinterpret_cast<const u32x4*>(ap + m * 32 * SB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, wv[n % D][nt]), __builtin_bit_cast(bf16x8, av[n & 1][m]), acc[m][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (n + D < NG) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[n % D][nt] = buf_load4(wr, wvoff + (unsigned)nt * 1024u, w_soff(n + D));
        } else {
            if constexpr (NEXT) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wv[n % D][nt] = buf_load4(wr_next, wvoff + (unsigned)nt * 1024u, w_soff(n % D));   // slot s holds group s on entry
            }
            tail(n + D - NG);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int WT, int WC, int MT, int NT, int C, int MINB>
__global__ void __launch_bounds__(256, MINB) mrf_pair_bf16_pf_kernel(const PairLaunch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_pair_pf[];
    char* lds = lds_pair_pf;
    static_assert(WT * WC == 4 && WC * NT * 32 == C, "a block owns all C channels");
    constexpr int SB = C * 2 + 16;
    constexpr int M = WT * MT * 32;
    constexpr int PPR = C / 8;                                           // 16-byte pieces per window row
    constexpr int RPI = 256 / PPR;                                       // window rows covered by one piece per thread
    constexpr int NQ = (M + kPairSpanMax + RPI - 1) / RPI;               // staged pieces per thread
    constexpr int D = 4;                                                 // weight ring depth (groups)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int L = a.L;
    const int ct0 = wc * NT;                                             // this wave's first 32-wide channel tile
    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 2u;
    const unsigned q_bytes = (unsigned)a.n_ct * 1024u;
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;
    const unsigned wvoff = (unsigned)ct0 * 1024u + (unsigned)lane * 16u;
    const char* a_lane = lds + (wt * MT * 32 + lo) * SB + hi * 16;
    const float slope = a.slope;

    // window staging: thread -> (row r_lane + u*RPI, piece pc); the byte offset of piece u inside the batch item's tensor
    // is vbase + u*row_step (rows < 0 wrap to >= 2^31, rows >= L exceed num_records: both read 0 -- the zero padding)
    const int r_lane = tid / PPR, pc_lane = tid - r_lane * PPR;
    constexpr unsigned row_step = (unsigned)(RPI * C) * 2u;
    char* const lds_wr = lds + r_lane * SB + pc_lane * 16;

    // jobs of this block: indices slot, slot + G, ... inside its XCD's contiguous range; job = ((b * tiles) + tile) * nz + branch
    const int xcd = blockIdx.x & 7, G = (int)(gridDim.x >> 3);
    const int tiles = a.n_jobs / a.nz;                                   // per batch item
    const long long total_jobs = (long long)a.n_jobs * a.B;
    struct Job { PairProblem p; int b, o0, z; };
    auto advance = [&](int& jx, Job& J) -> bool {                        // first valid job at jx, jx + G, ...; block-uniform
        for (; jx < a.jobs_per_xcd; jx += G) {
            const long long job = (long long)xcd * a.jobs_per_xcd + jx;
            if (job >= total_jobs) return false;
            const int t2 = (int)(job / a.nz), zr = (int)(job - (long long)t2 * a.nz);
            const int bb = t2 / tiles, tile = t2 - bb * tiles;
            const int z = a.nz - 1 - zr;                                 // heaviest branch first
            PairProblem p = a.p[0];
            if (z == 1) p = a.p[1];
            if (z == 2) p = a.p[2];
            if (z == 3) p = a.p[3];
            // (the job state is block-uniform by construction; saying so right here keeps the branch below and every buffer
            //  descriptor built from the job in SGPRs -- hipcc routes the branch select through a private table, whose loads
            //  it treats as per-lane values: exec-masked control flow and a waterfall loop around each buffer access)
            const int ks_u = __builtin_amdgcn_readfirstlane(p.ks);
            const int o0 = tile * (M - (ks_u - 1));
            if (o0 >= L) continue;                                       // (tiles are counted for the smallest T_OUT of the launch)
            J.p.x = uniform_ptr(p.x); J.p.w1 = uniform_ptr(p.w1); J.p.w2 = uniform_ptr(p.w2);
            J.p.y = uniform_ptr(p.y);
            J.p.ks = ks_u; J.p.dil = __builtin_amdgcn_readfirstlane(p.dil);
            J.b = bb; J.o0 = o0; J.z = z;
            return true;
        }
        return false;
    };
    auto window_vbase = [&](const Job& J) -> unsigned {
        const int in_row0 = J.o0 - (J.p.ks - 1) / 2 - J.p.dil * (J.p.ks - 1) / 2;
        return (unsigned)((in_row0 + r_lane) * C + 8 * pc_lane) * 2u;
    };

    u32x4 st[NQ];                                                        // the window in flight
    auto window_request_one = [&](int u, __amdgpu_buffer_rsrc_t xr, unsigned vbase, int R) {
        // rows past this branch's window are not requested: they would be real rows, i.e. HBM reads nobody uses
        st[u] = buf_load4(xr, r_lane + u * RPI < R ? vbase + (unsigned)u * row_step : kOob, 0);
    };
    auto window_write = [&](int R) {                                     // bf16(LeakyReLU(x)) on the way in
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (r_lane + u * RPI < R) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned w = st[u][e];
                    o[e] = pack_bf2(lrelu_max(bf_lo(w), slope), lrelu_max(bf_hi(w), slope));
                }
                *reinterpret_cast<u32x4*>(lds_wr + u * RPI * SB) = o;
            }
    };

    // The biases of every branch (both convs) live in LDS behind the window for the whole life of the block: 2*C floats per
    // branch, read back as the accumulators' start values (no bias registers, no bias request per job).
    float* const lds_bias = reinterpret_cast<float*>(lds + a.bias_off);
#pragma unroll
    for (int z = 0; z < kMaxGroup; ++z)
        if (z < a.nz && tid < C / 2) {
            const float* src = tid < C / 4 ? a.p[z].b1 + 4 * tid : a.p[z].b2 + 4 * (tid - C / 4);
            *reinterpret_cast<f32x4*>(lds_bias + z * 2 * C + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
        }
    f32x16 acc[MT][NT];
    auto init_acc = [&](int z, int conv) {                               // the accumulators START at the bias
        const float* bp = lds_bias + (z * 2 + conv) * C + ct0 * 32 + 4 * hi;   // (nt, g) -> channels (ct0+nt)*32 + 8g + 4hi + {0..3}
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + nt * 32 + 8 * g);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][nt][4 * g + e] = b4[e];
            }
    };
    u32x4 wv[D][NT];

    // epilogue geometry (job-independent part)
    constexpr int RS = NT * 32 * 4 + 16;               // scratch row stride (bytes) = 16 * odd
    constexpr int PPRO = NT * 4;                       // 16-byte bf16 pieces per row of this wave's channel span
    constexpr int NP = 2 * NT;                         // pieces per lane and m-tile
    int pscr[NP], prow[NP], pco[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = j * 64 + lane;
        prow[j] = q / PPRO;
        const int pc = q - prow[j] * PPRO;
        pscr[j] = prow[j] * RS + pc * 32;
        pco[j] = ct0 * 32 + 8 * pc;
    }
    char* scr = lds + wave * (32 * RS);

    int jx = (int)(blockIdx.x >> 3);
    Job J;
    if (!advance(jx, J)) return;
    // ---- prologue: the only window wait a block exposes ------------------------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(J.p.x + (size_t)J.b * L * C, tensor_bytes);
        const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(J.p.w1, (unsigned)J.p.ks * tap_bytes);
        ring_request<NT, C, D>(wv, wr1, wvoff, q_bytes, tap_bytes);
        const unsigned vb = window_vbase(J);
        const int R = M + (J.p.ks - 1) * J.p.dil;
#pragma unroll
        for (int u = 0; u < NQ; ++u) window_request_one(u, xr, vb, R);
        window_write(R);
    }
    __syncthreads();

    for (;;) {
        int jxn = jx + G;
        Job Jn = J;                                            // (stays J when there is no further job: never dereferenced then)
        const bool more = advance(jxn, Jn);
        const int ks = J.p.ks, dil = J.p.dil;
        const int h2 = (ks - 1) / 2;
        const int T_OUT = M - (ks - 1);
        const int o0 = J.o0;
        const size_t item = (size_t)J.b * L * C;
        const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(J.p.w1, (unsigned)ks * tap_bytes);
        const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(J.p.w2, (unsigned)ks * tap_bytes);
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(J.p.y + item, tensor_bytes);
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(J.p.x + item, tensor_bytes);
        // the job that follows (zero-length descriptors when there is none: such loads return 0 without touching memory)
        const __amdgpu_buffer_rsrc_t xrn = make_rsrc(Jn.p.x + (size_t)Jn.b * L * C, more ? tensor_bytes : 0u);
        const __amdgpu_buffer_rsrc_t wr1n = make_rsrc(Jn.p.w1, more ? (unsigned)Jn.p.ks * tap_bytes : 0u);
        const unsigned vbn = window_vbase(Jn);
        const int Rn = M + (Jn.p.ks - 1) * Jn.p.dil;

        unsigned pvoff[MT][NP];
        u32x4 resv[MT][NP];
        constexpr int NRES = (MT - 1) * NP;            // residual pieces requested in conv2's tail (m-tiles 1..)
        constexpr int NEXTRA = NRES + NQ;
        constexpr int PER = (NEXTRA + D - 1) / D;      // extra requests per tail group, oldest first: residuals, then the window

        auto iteration = [&](auto ks_tag) {
            constexpr int KS = decltype(ks_tag)::value;
            // (a distinct marker per kernel size: hipcc otherwise hoists the three paths' common head -- bias reads, the first
            //  activation fragments -- above the dispatch and spills it to scratch across the branch)
            asm volatile("; conv pair, %0 taps" :: "n"(KS) : "memory");
            // ---- conv1; its tail fetches conv2's first fragments ---------------------------------------------------------
            init_acc(J.z, 0);
            pf_mma<KS, MT, NT, C, D, true>(acc, wv, a_lane, dil * SB, wr1, wr2, wvoff, q_bytes, tap_bytes, [](int) {});
            __syncthreads();                                              // every wave is done with the x window
            // ---- xt -> LDS: bf16(LeakyReLU(bf16(acc))) (acc already holds the bias), zero outside [0, L) ----------------
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
                            o[e2] = pack_bf2(lrelu_max(bf_lo(t), slope), lrelu_max(bf_hi(t), slope)) & keep;           // conv2's operand
                        }
                        *reinterpret_cast<u32x2*>(lds + row_l * SB + ((ct0 + nt) * 32 + 8 * g + 4 * hi) * 2) = o;
                    }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int im = (wt * MT + m) * 32 + prow[j];
                    const int o = o0 + im;
                    pvoff[m][j] = (im < T_OUT && o < L) ? (unsigned)(o * C + pco[j]) * 2u : kOob;
                }
#pragma unroll
            for (int j = 0; j < NP; ++j) resv[0][j] = buf_load4(rr, pvoff[0][j], 0);
            __syncthreads();
            // ---- conv2 (dilation 1); its tail requests, oldest first: residuals of m-tiles 1.., then the NEXT window ------------
            init_acc(J.z, 1);
            pf_mma<KS, MT, NT, C, D, false>(acc, wv, a_lane, SB, wr2, wr2, wvoff, q_bytes, tap_bytes, [&](int t) {
#pragma unroll
                for (int e = 0; e < NEXTRA; ++e) {
                    if (e / PER != t) continue;
                    if (e < NRES) { const int m = 1 + e / NP, j = e % NP; resv[m][j] = buf_load4(rr, pvoff[m][j], 0); }
                    else          window_request_one(e - NRES, xrn, vbn, Rn);
                }
            });
            __syncthreads();                                   // the epilogue scratch aliases the window
            // ---- epilogue: + x, one rounding to bf16 (per-wave LDS transpose, as in mrf_pair_bf16_kernel) ---------------
#pragma unroll
            for (int m = 0; m < MT; ++m) {
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
                    const u32x4 rv = resv[m][j];
                    outp[j][0] = pack_bf2(lo4[0] + bf_lo(rv[0]), lo4[1] + bf_hi(rv[0]));
                    outp[j][1] = pack_bf2(lo4[2] + bf_lo(rv[1]), lo4[3] + bf_hi(rv[1]));
                    outp[j][2] = pack_bf2(hi4[0] + bf_lo(rv[2]), hi4[1] + bf_hi(rv[2]));
                    outp[j][3] = pack_bf2(hi4[2] + bf_lo(rv[3]), hi4[3] + bf_hi(rv[3]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)pvoff[m][j], 0, 0);
                asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 stores (see mrf_conv_mfma_f32.h)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (ks == 3)      iteration(std::integral_constant<int, 3>{});
        else if (ks == 7) iteration(std::integral_constant<int, 7>{});
        else              iteration(std::integral_constant<int, 11>{});
        if (!more) break;
        __syncthreads();                                       // every wave is done with its scratch
        // the next job's first conv1 fragments (L2): younger than the window, so the write below does not wait for them,
        // and back by the time the barrier behind it has passed
        ring_request<NT, C, D>(wv, wr1n, wvoff, q_bytes, tap_bytes);
        window_write(Rn);
        __syncthreads();
        J = Jn;
        jx = jxn;
    }
}

// ---- launch ----------------------------------------------------------------------------------------
#ifndef IRIS_B16_PAIR_PF_DEFAULT
#define IRIS_B16_PAIR_PF_DEFAULT 0
#endif

// Tile of the persistent kernel for C channels (the staged window in registers costs one resident block per CU
// against the non-persistent kernel's tile).
inline bool pair_pf_tile_for(int C, PairTile* t) {
    if (C == 32) {
        const int v = IRIS_DIAG_ENV("IRIS_B16_PAIR_PF32", 1);        // 0: the non-persistent kernel
        if (v == 1) { *t = PairTile{4, 1, 3, 1, 3, 384}; return true; }
        if (v == 2) { *t = PairTile{4, 1, 2, 1, 4, 256}; return true; }
        return false;
    }
    if (C == 64) {
        const int v = IRIS_DIAG_ENV("IRIS_B16_PAIR_PF64", 1);
        if (v == 1) { *t = PairTile{2, 2, 2, 1, 3, 128}; return true; }
        if (v == 2) { *t = PairTile{2, 2, 4, 1, 2, 256}; return true; }
        if (v == 3) { *t = PairTile{2, 2, 3, 1, 3, 192}; return true; }
        if (v == 4) { *t = PairTile{2, 2, 2, 1, 4, 128}; return true; }
        return false;
    }
    return false;
}

inline bool pair_pf_applicable(const PairLaunch& a, int nz) {
    PairTile t;
    if (!IRIS_DIAG_ENV("IRIS_B16_PAIR_PF", IRIS_B16_PAIR_PF_DEFAULT)) return false;
    if (!pair_applicable(a, nz) || !pair_pf_tile_for(a.C, &t)) return false;
    for (int j = 0; j < nz; ++j) {
        const int ks = a.p[j].ks;
        if (ks != 3 && ks != 7 && ks != 11) return false;                        // the MFMA loops are unrolled for the V1 MRF
        if ((size_t)(t.M + (ks - 1) * a.p[j].dil) * (a.C * 2 + 16) > 64 * 1024) return false;
    }
    return true;
}

inline hipError_t launch_pair_bf16_pf(PairLaunch& a, int nz, hipStream_t stream) {
    PairTile t;
    if (!pair_pf_tile_for(a.C, &t)) return hipErrorInvalidValue;
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nz; ++i)
            if (a.p[j].x == a.p[i].y) return hipErrorInvalidValue;               // never in place
    a.nz = nz;
    a.Qp = packed_qsteps(a.C);
    a.n_ct = packed_cotiles(a.C);
    a.ablate = 0;
    a.dbg = nullptr;
    int span = 0, kmax = 1;
    for (int j = 0; j < nz; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
        if (a.p[j].ks > kmax) kmax = a.p[j].ks;
    }
    const int t_out_min = t.M - (kmax - 1);
    const long long tiles = (a.L + t_out_min - 1) / t_out_min;
    const long long n_jobs = tiles * nz;                                         // per batch item
    const long long total = n_jobs * a.B;
    if (n_jobs > 0x3fffffffLL || total > 0x3fffffffLL * 8) return hipErrorInvalidValue;
    a.n_jobs = (int)n_jobs;
    const long long per_xcd = (total + 7) / 8;
    if (per_xcd > 0x7fffffffLL) return hipErrorInvalidValue;
    a.jobs_per_xcd = (int)per_xcd;
    // persistent grid: at most MINB resident blocks per CU; per XCD a block count that is not a multiple of the
    // branch count, so that a block's jobs i, i + G, ... cycle through the branches (equal work per block)
    const long long slots_per_xcd = (long long)device_cu_count() * t.MINB / 8;
    long long G = per_xcd < slots_per_xcd ? per_xcd : slots_per_xcd;
    if (G < 1) G = 1;
    if (per_xcd > G && nz > 1 && G % nz == 0) G -= 1;
    if (G < 1) G = 1;
    const size_t window_bytes = (size_t)(t.M + span) * (a.C * 2 + 16);
    const size_t scratch_bytes = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);          // epilogue transpose, aliases the window
    const size_t region_bytes = ((window_bytes > scratch_bytes ? window_bytes : scratch_bytes) + 15) & ~(size_t)15;
    a.bias_off = (int)region_bytes;
    const size_t lds_bytes = region_bytes + (size_t)nz * 2 * a.C * sizeof(float);
    dim3 grid((unsigned)(G * 8), 1u, 1u), block(256);
#define IRIS_PAIR_PF_CASE(WT_, WC_, MT_, NT_, C_, MINB_)                                                     \
    if (a.C == C_ && t.WT == WT_ && t.WC == WC_ && t.MT == MT_ && t.NT == NT_ && t.MINB == MINB_) {        \
        auto kfn = mrf_pair_bf16_pf_kernel<WT_, WC_, MT_, NT_, C_, MINB_>;                                   \
        if (lds_bytes > 64 * 1024) return hipErrorInvalidValue;                                              \
        return ::iris::launch_kernel_named("mrf_pair_bf16_pf_kernel", kfn, grid, block, lds_bytes, stream, a);                                \
    }
    IRIS_PAIR_PF_CASE(4, 1, 3, 1, 32, 3)
    IRIS_PAIR_PF_CASE(4, 1, 2, 1, 32, 4)
    IRIS_PAIR_PF_CASE(2, 2, 2, 1, 64, 4)
    IRIS_PAIR_PF_CASE(2, 2, 2, 1, 64, 3)
    IRIS_PAIR_PF_CASE(2, 2, 3, 1, 64, 3)
    IRIS_PAIR_PF_CASE(2, 2, 4, 1, 64, 2)
#undef IRIS_PAIR_PF_CASE
    return hipErrorInvalidValue;
}

}  // namespace b16
}  // namespace iris
