// mrf_pair_bf16.h -- one ResBlock conv PAIR of all MRF branches in one launch, bf16 storage (gfx950).
//
// Reference semantics: one iteration of ResBlock.forward's loop (src/iris/hifigan_pretrained.py:64-71)
//     xt = Conv1d_{k, dil d}(LeakyReLU(x));  y = Conv1d_{k, dil 1}(LeakyReLU(xt)) + x
// for the branches k = 3 / 7 / 11 of a stage (hifigan_pretrained.py:130-136), with the rounding points of the
// bf16-storage variant (conv_mfma_bf16.h; oracle/hifigan_oracle.py:generator_forward_bf16): LeakyReLU(x) -> bf16,
// xt = bf16(acc + bias), LeakyReLU(xt) -> bf16, y = bf16(acc + bias + x).  The numbers are bit for bit those of the
// two separate launches of conv_mfma_bf16_kernel (same MFMA order, same roundings) -- what changes is the traffic:
// xt never leaves the CU.  Unfused, a pair costs five tensor passes over HBM (conv1: read x, write xt; conv2: read
// xt, read x, write y); fused it costs two (read x window, write y) plus the residual re-read, which the L2 / MALL
// serves because the same block fetched those rows a few microseconds earlier.  At bf16 the C = 32 / 64 stages are
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
//   5. epilogue as in conv_mfma_bf16.h (per-wave LDS transpose, 16-byte residual loads and stores).
// The (k-1)/M rows of conv2 that are computed and thrown away are the price of keeping xt on chip: 0.5-2.6 % at
// M = 384, 1-5 % at M = 192.
//
// Block -> job mapping: job = (tile, branch) with the branch fastest (the branches of the first pair of a stage all
// read the same x); jobs are dealt to the XCDs in contiguous ranges (blockIdx.x % 8 = XCD), so the halo rows that
// neighbouring tiles share and the windows that sibling branches share are re-read from that XCD's L2.
#pragma once
#include "conv_mfma_bf16.h"

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
    int ablate;           // diagnostics only: 1 no staging loads, 2 no MFMA loops, 4 no stores, 8 no residual loads
};

// Rows [in_row0, in_row0 + R) x all CIC channels of bf16(LeakyReLU(x)) -> LDS (row stride 2*CIC + 16 bytes).
template <int CIC>
__device__ __forceinline__ void pair_stage_rows(const uint16_t* x_item, unsigned tensor_bytes, int L, char* lds,
                                                int in_row0, int R, float slope, bool skip_loads) {
    constexpr int SB = CIC * 2 + 16;
    constexpr int PPR = CIC / 8;          // 16-byte pieces per row (power of two)
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(x_item, tensor_bytes);
    const int total = R * PPR;
    constexpr int U = 4;
    for (int base = 0; base < total; base += 256 * U) {
        u32x4 v[U];
        int ldso[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + (int)threadIdx.x;
            const int r = idx / PPR, pc = idx & (PPR - 1);
            const int row = in_row0 + r;
            const bool ok = idx < total && row >= 0 && row < L && !skip_loads;
            v[u] = buf_load4(r0, ok ? (unsigned)(row * CIC + 8 * pc) * 2u : kOob, 0);
            ldso[u] = idx < total ? r * SB + pc * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned w = v[u][e];
                o[e] = pack_bf2(lrelu1(bf_lo(w), slope), lrelu1(bf_hi(w), slope));
            }
            if (ldso[u] >= 0) *reinterpret_cast<u32x4*>(lds + ldso[u]) = o;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;

    // job -> (tile, branch): contiguous job ranges per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= a.jobs_per_xcd) return;
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

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;
    };

    // ---- 1. x window ----------------------------------------------------------------------------------
    pair_stage_rows<C>(p.x + item, tensor_bytes, L, lds, o0 - h2 - h1, M + (ks - 1) * dil, a.slope, (a.ablate & 1) != 0);
    // bias of conv1 for this lane's channels: (nt, g) -> channels (ct0+nt)*32 + 8g + 4hi + {0..3}
    f32x4 bias4[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[nt][g] = *reinterpret_cast<const f32x4*>(p.b1 + (ct0 + nt) * 32 + 8 * g + 4 * hi);
    __syncthreads();
    // ---- 2. conv1 ---------------------------------------------------------------------------------------
    zero_acc();
    if (!(a.ablate & 2))
        mma_chunk<MT, NT, C>(acc, a_lane, dil * SB, make_rsrc(p.w1, (unsigned)ks * tap_bytes), wvoff, q_bytes, tap_bytes, 0u, ks);
    __syncthreads();                                   // every wave is done with the x window
    // ---- 3. xt -> LDS: bf16(LeakyReLU(bf16(acc + bias1))), zero outside [0, L) --------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row_l = (wt * MT + m) * 32 + lo;
        const int row_g = o0 - h2 + row_l;
        const bool inside = row_g >= 0 && row_g < L;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const unsigned t = pack_bf2(acc[m][nt][4 * g + 2 * e2] + bias4[nt][g][2 * e2],
                                                acc[m][nt][4 * g + 2 * e2 + 1] + bias4[nt][g][2 * e2 + 1]);   // the stored xt
                    o[e2] = inside ? pack_bf2(lrelu1(bf_lo(t), a.slope), lrelu1(bf_hi(t), a.slope)) : 0u;      // conv2's operand
                }
                *reinterpret_cast<u32x2*>(lds + row_l * SB + ((ct0 + nt) * 32 + 8 * g + 4 * hi) * 2) = o;
            }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[nt][g] = *reinterpret_cast<const f32x4*>(p.b2 + (ct0 + nt) * 32 + 8 * g + 4 * hi);
    __syncthreads();
    // ---- 4. conv2 (dilation 1; rows M .. M+k-2 of the window hold stale bytes: they only reach outputs >= T_OUT) ----
    zero_acc();
    if (!(a.ablate & 2))
        mma_chunk<MT, NT, C>(acc, a_lane, SB, make_rsrc(p.w2, (unsigned)ks * tap_bytes), wvoff, q_bytes, tap_bytes, 0u, ks);
    __syncthreads();                                   // the epilogue scratch aliases the window
    // ---- 5. epilogue: + bias2 + x, one rounding to bf16, coalesced 16-byte stores ----------------------------
    constexpr int RS = NT * 32 * 4 + 16;               // scratch row stride (bytes) = 16 * odd
    constexpr int PPRO = NT * 4;                       // 16-byte bf16 pieces per row of this wave's channel span
    constexpr int NP = 2 * NT;                         // pieces per lane and m-tile
    char* scr = lds + wave * (32 * RS);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + item, tensor_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.x + item, (a.ablate & 8) ? 0u : tensor_bytes);
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
    u32x4 resv[MT][NP];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NP; ++j) resv[m][j] = buf_load4(rr, pvoff[m][j], 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[m][nt][4 * g + e] + bias4[nt][g][e];
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
            __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)((a.ablate & 4) ? kOob : pvoff[m][j]), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // (store data stays live until every store of the group has issued: see mrf_conv_mfma_f32.h)
#pragma unroll
        for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- launch ----------------------------------------------------------------------------------------
struct PairTile { int WT, WC, MT, NT, MINB, M; };

inline bool pair_tile_for(int C, PairTile* t) {
    const int v64 = IRIS_DIAG_ENV("IRIS_B16_PAIR64", 0);
    if (C == 32) { *t = PairTile{4, 1, 3, 1, 4, 384}; return true; }
    if (C == 64) {
        if (v64 == 1) { *t = PairTile{4, 1, 2, 2, 3, 256}; return true; }
        *t = PairTile{2, 2, 3, 1, 4, 192};
        return true;
    }
    return false;
}

// True when the pair launch `a` (nz branches, a.C channels) can take the fused kernel.
inline bool pair_applicable(const PairLaunch& a, int nz) {
    PairTile t;
    if (nz < 1 || nz > kMaxGroup || !pair_tile_for(a.C, &t)) return false;
    if ((double)a.L * a.C * 2.0 >= 2147483648.0) return false;
    for (int j = 0; j < nz; ++j) {
        const int ks = a.p[j].ks, d = a.p[j].dil;
        if (ks < 1 || !(ks & 1) || d < 1) return false;
        if (ks - 1 >= t.M / 2) return false;                                     // keeps T_OUT >= M / 2
        if ((size_t)(t.M + (ks - 1) * d) * (a.C * 2 + 16) > 64 * 1024) return false; // window must leave room for several blocks per CU
    }
    return IRIS_DIAG_ENV("IRIS_B16_PAIR", 1) != 0;
}

inline hipError_t launch_pair_bf16(PairLaunch& a, int nz, hipStream_t stream) {
    PairTile t;
    if (!pair_tile_for(a.C, &t)) return hipErrorInvalidValue;
    a.nz = nz;
    a.Qp = packed_qsteps(a.C);
    a.n_ct = packed_cotiles(a.C);
    a.ablate = IRIS_DIAG_ENV("IRIS_B16_ABLATE", 0);
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
    const size_t scratch_bytes = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);
    const size_t lds_bytes = window_bytes > scratch_bytes ? window_bytes : scratch_bytes;
    dim3 grid((unsigned)(a.jobs_per_xcd * 8), (unsigned)a.B, 1u), block(256);
#define IRIS_PAIR_CASE(WT_, WC_, MT_, NT_, C_, MINB_)                                                        \
    if (a.C == C_ && t.WT == WT_ && t.WC == WC_ && t.MT == MT_ && t.NT == NT_) {                            \
        auto kfn = mrf_pair_bf16_kernel<WT_, WC_, MT_, NT_, C_, MINB_>;                                      \
        if (lds_bytes > 64 * 1024) {                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
            if (e != hipSuccess) return e;                                                                   \
        }                                                                                                    \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, stream, a);                                          \
        return hipGetLastError();                                                                            \
    }
    IRIS_PAIR_CASE(4, 1, 3, 1, 32, 4)
    IRIS_PAIR_CASE(2, 2, 3, 1, 64, 4)
    IRIS_PAIR_CASE(4, 1, 2, 2, 64, 3)
#undef IRIS_PAIR_CASE
    return hipErrorInvalidValue;
}

}  // namespace b16
}  // namespace iris
