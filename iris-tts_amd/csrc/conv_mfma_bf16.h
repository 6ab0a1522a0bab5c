// conv_mfma_bf16.h -- bf16-storage variant of the generator's Conv1d / ConvTranspose1d kernel (gfx950).
//
// Same layers and dataflow as conv_mfma_f32.h (reference: HiFiGANModel.forward,
// src/iris/hifigan_pretrained.py:123-143; ResBlock :64-71), for BASELINE.json configs[2]
// ("batch=32, 80x500-frame mels, bf16"):
//   * activations live in HBM as bf16, channels-last [B, L, C]: half the bytes of the fp32 path;
//   * weights are bf16 (rounded to nearest-even from the folded fp32 weights on the host);
//   * products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate);
//   * bias, residual add, LeakyReLU and the MRF mean are evaluated in fp32 and rounded to bf16 once,
//     where a value is stored (epilogue) or becomes an MFMA operand (LDS staging).
// The reference has no bf16 path, so its tolerance is unpinned by the reference: tests compare against a CPU
// restatement with the same rounding points (tests/test_gpu_bf16.py) and report the distance to fp32.
//
// Roofline: at bf16 the MRF convs of the C = 32 / 64 stages are HBM-bound (86 / 172 FLOP/B against a
// machine balance of ~312), C = 128 sits at the ridge and C = 256 stays MFMA-bound.
//
// Work split.  A 256-thread block owns (WT*MT*32) time rows x (WC*NT*32) output channels.  The input
// window (tile + dilated-tap halo) of CIC channels is staged once into LDS as bf16 with the input
// activation applied; every tap re-reads it at a shifted row.  The MFMA is issued as D = W x X^T:
//   A operand (weights):     one 16-byte buffer load per lane and (tap, 16 channels, 32-wide C_out tile) from
//                            the host-packed fragment order, prefetched four groups ahead in a register ring;
//   B operand (activations): one ds_read_b128 per lane and (tap, 16 channels, 32 rows); LDS row stride
//                            2*CIC + 16 bytes = 16 * odd -> conflict-free b128 reads and writes;
//   D: lane = time step, registers 4g..4g+3 = 4 consecutive channels; the epilogue turns each 32-row tile through
//      a per-wave LDS scratch so that residual loads and bf16 stores are 16 bytes per lane and coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "diag_env.h"

namespace iris {
namespace b16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef IRIS_B16_XCD_FOLD
#define IRIS_B16_XCD_FOLD 1     // 0: XCD grouping per batch item only (the round-3 form; A/B builds)
#endif

#ifndef IRIS_B16_UPS32_MT
#define IRIS_B16_UPS32_MT 3     // m-tiles per wave of the ConvTranspose phases at C_out <= 32 (A/B builds)
#endif

enum InAct : int { IN_ACT_NONE = 0, IN_ACT_LRELU = 1, IN_ACT_MRF_LRELU = 2 };

constexpr int kMaxGroup = 4;

struct Problem {
    const void* x;        // bf16 [B, L_in, C_in]; fp32 [B, C_in, L_in] when Launch::x_f32_cf (the mel)
    const void* wp;       // packed bf16 weight fragments, see pack_conv1d_bf16
    const float* bias;    // [C_out] fp32
    const uint16_t* res;  // bf16 residual added in the epilogue [B, L_out, C_out], or nullptr
    uint16_t* y;          // bf16 output [B, L_out, C_out]
    int ks;               // taps
    int dil;              // rows between taps
    int pad_left;         // input row of tap 0 for output row-index i is i - pad_left
    int reserved;
};

struct Launch {
    Problem p[kMaxGroup];
    const uint16_t* xmrf[kMaxGroup];  // in_act == IN_ACT_MRF_LRELU: input = lrelu((xmrf[0]+...) * (1/n_mrf))
    int n_mrf;
    int B, L_in, L_out, C_in, C_out;
    int n_idx;            // output row-indices (L_out for a conv, L_in + taps - 1 for a ConvTranspose phase)
    int out_stride;       // output row o = i*out_stride + out_off (+ phase)
    int out_off;
    int z_is_phase;       // the z part of blockIdx.x is a ConvTranspose phase of problem 0 (else a problem)
    unsigned phase_wp_bytes;  // bytes between the packed weights of consecutive phases
    int in_act;
    int x_f32_cf;
    float slope;
    float inv_n_mrf;
    int nz;               // problems (or phases) interleaved along blockIdx.x
    int n_co_blk;         // blocks along C_out
    int Qp;               // padded number of 16-channel steps in the packed weights
    int n_ct;             // 32-wide C_out tiles in the packed weights
    int n_items;          // window items: time tiles (x nz when z is a problem index)
    int n_share;          // consecutive sub-blocks that stage the same window: C_out blocks (x phases for a ConvTranspose)
    int xcd_group;        // the sub-blocks of a window item share an XCD (see the kernel); the batch index is then part
                          // of the window item (item = b * n_items + tile), so that short items of a large batch qualify too
    int fold_b;           // grouped launches: the batch index is part of blockIdx.x (item = b * n_items + tile)
    int z_in_y;           // the problem index is blockIdx.y / B (heaviest problem dispatched first, over ALL tiles and batch items)
    int ablate;           // diagnostics only (env IRIS_B16_ABLATE): 1 no staging loads, 2 no MFMA loop, 4 no stores,
                          // 8 no residual loads, 16 every weight fragment from one (L1-resident) address.  Results are wrong.
};

// ---- small helpers -------------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lrelu1(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ u32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store2(u32x2 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
constexpr unsigned kOob = 0x80000000u;   // >= any num_records used here: loads return 0, stores are dropped

// ---- LDS staging ---------------------------------------------------------------------------------
// Rows [in_row0, in_row0 + R) x channels [c0, c0 + CIC) of the activated input -> LDS (bf16).
template <int CIC>
__device__ __forceinline__ void stage_window(const Launch& a, const Problem& p, char* lds, int b,
                                             int in_row0, int R, int c0) {
    constexpr int SB = CIC * 2 + 16;
    constexpr int PPR = CIC / 8;          // 16-byte pieces (8 channels) per row
    const int tid = threadIdx.x;
    const int L_in = a.L_in, C_in = a.C_in;
    if (!a.x_f32_cf) {
        const unsigned tensor_bytes = (unsigned)L_in * (unsigned)C_in * 2u;
        const size_t boff = (size_t)b * L_in * C_in;
        const bool mrf = a.in_act == IN_ACT_MRF_LRELU;
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc((mrf ? a.xmrf[0] : (const uint16_t*)p.x) + boff, tensor_bytes);
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc((mrf && a.n_mrf > 1 ? a.xmrf[1] : a.xmrf[0]) + boff, mrf && a.n_mrf > 1 ? tensor_bytes : 0u);
        const __amdgpu_buffer_rsrc_t r2 = make_rsrc((mrf && a.n_mrf > 2 ? a.xmrf[2] : a.xmrf[0]) + boff, mrf && a.n_mrf > 2 ? tensor_bytes : 0u);
        const __amdgpu_buffer_rsrc_t r3 = make_rsrc((mrf && a.n_mrf > 3 ? a.xmrf[3] : a.xmrf[0]) + boff, mrf && a.n_mrf > 3 ? tensor_bytes : 0u);
        const int total = R * PPR;
        const float slope = a.in_act == IN_ACT_NONE ? 1.f : a.slope;
        auto piece = [&](int idx, unsigned& voff, int& ldso) {
            const int r = idx / PPR, pc = idx - r * PPR;
            const int row = in_row0 + r, ci = c0 + 8 * pc;
            const bool ok = idx < total && row >= 0 && row < L_in && ci < C_in;
            voff = (ok && !(a.ablate & 1)) ? (unsigned)(row * C_in + ci) * 2u : kOob;
            ldso = idx < total ? r * SB + pc * 16 : -1;
        };
        if (!mrf) {
            constexpr int U = 4;          // 16-byte pieces in flight per thread
            for (int base = 0; base < total; base += 256 * U) {
                u32x4 v0[U];
                int ldso[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    unsigned voff;
                    piece(base + u * 256 + tid, voff, ldso[u]);
                    v0[u] = buf_load4(r0, voff, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = pack_bf2(lrelu1(bf_lo(v0[u][e]), slope), lrelu1(bf_hi(v0[u][e]), slope));
                    if (ldso[u] >= 0) *reinterpret_cast<u32x4*>(lds + ldso[u]) = o;
                }
            }
        } else {
            constexpr int U = 2;
            for (int base = 0; base < total; base += 256 * U) {
                u32x4 v0[U], v1[U], v2[U], v3[U];
                int ldso[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    unsigned voff;
                    piece(base + u * 256 + tid, voff, ldso[u]);
                    v0[u] = buf_load4(r0, voff, 0);
                    v1[u] = buf_load4(r1, voff, 0);
                    v2[u] = buf_load4(r2, voff, 0);
                    v3[u] = buf_load4(r3, voff, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = ((bf_lo(v0[u][e]) + bf_lo(v1[u][e])) + bf_lo(v2[u][e])) + bf_lo(v3[u][e]);
                        float hi = ((bf_hi(v0[u][e]) + bf_hi(v1[u][e])) + bf_hi(v2[u][e])) + bf_hi(v3[u][e]);
                        lo *= a.inv_n_mrf; hi *= a.inv_n_mrf;
                        o[e] = pack_bf2(lrelu1(lo, slope), lrelu1(hi, slope));
                    }
                    if (ldso[u] >= 0) *reinterpret_cast<u32x4*>(lds + ldso[u]) = o;
                }
            }
        }
    } else {
        // the mel: fp32, channels-first [B, C_in, L_in] (hifigan_pretrained.py:228); lanes run along time
        const float* x = (const float*)p.x;
        const int total = R * CIC;
        for (int idx = tid; idx < total; idx += 256) {
            const int c = idx / R, r = idx - c * R;
            const int row = in_row0 + r, ci = c0 + c;
            float v = 0.f;
            if (row >= 0 && row < L_in && ci < C_in) {
                v = x[((size_t)b * C_in + ci) * L_in + row];
                if (a.in_act == IN_ACT_LRELU) v = lrelu1(v, a.slope);
            }
            const unsigned pk = pack_bf2(v, 0.f);
            *reinterpret_cast<uint16_t*>(lds + r * SB + c * 2) = (uint16_t)(pk & 0xffffu);
        }
    }
}

// ---- MFMA main loop over one staged chunk ----------------------------------------------------------
// One "group" = 16 input channels of one tap = MT*NT MFMAs.  Weight fragments run four groups ahead of
// their use (register ring); groups past the last tap read beyond the descriptor and get zeros, so the
// loop needs no tail.  Activation fragments are read one group ahead.
template <int MT, int NT, int CIC>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MT][NT], const char* a_lane, int dil_bytes,
                                          __amdgpu_buffer_rsrc_t wr, unsigned wvoff, unsigned q_bytes,
                                          unsigned tap_bytes, unsigned q0_bytes, int ks) {
    constexpr int QPC = CIC / 16;
    constexpr int QL = QPC == 8 ? 3 : (QPC == 4 ? 2 : (QPC == 2 ? 1 : 0));
    static_assert(QPC == 8 || QPC == 4 || QPC == 2 || QPC == 1, "CIC must be 16, 32, 64 or 128");
    constexpr int SB = CIC * 2 + 16;
    // ring depth: a group of 8 MFMAs lasts 256 cycles, two of them cover an L2 hit (and ks*2 groups need no tail)
    constexpr int D = (QPC == 2 && MT * NT >= 8) ? 2 : 4;
    const int NG = ks * QPC;
    auto w_soff = [&](int n) -> unsigned {
        return (unsigned)(n >> QL) * tap_bytes + q0_bytes + (unsigned)(n & (QPC - 1)) * q_bytes;
    };
    auto load_a = [&](u32x4 (&av)[MT], int n) {
        int tap = n >> QL;
        tap = tap < ks ? tap : ks - 1;
        const char* ap = a_lane + tap * dil_bytes + (n & (QPC - 1)) * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const u32x4*>(ap + m * 32 * SB);
    };
    u32x4 wv[D][NT], av[2][MT];
    // (issue order pinned: vmcnt is in-order, so slot 0 must be the OLDEST request when the loop is entered --
    //  hipcc otherwise loads it last and the loop head waits for vmcnt(0) on every iteration)
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wv[i][nt] = buf_load4(wr, wvoff + (unsigned)nt * 1024u, w_soff(i));
        __builtin_amdgcn_sched_barrier(0);
    }
    load_a(av[0], 0);
    for (int n0 = 0; n0 < NG; n0 += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int n = n0 + i;
            load_a(av[(i + 1) & 1], n + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, wv[i][nt]), __builtin_bit_cast(bf16x8, av[i & 1][m]),
                        acc[m][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wv[i][nt] = buf_load4(wr, wvoff + (unsigned)nt * 1024u, w_soff(n + D));
        }
    }
}

// ---- the kernel -----------------------------------------------------------------------------------
template <int WT, int WC, int MT, int NT, int CIC, int MINB>
__global__ void __launch_bounds__(256, MINB) conv_mfma_bf16_kernel(const Launch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_b16[];
    char* lds = lds_b16;
    constexpr int SB = CIC * 2 + 16;
    constexpr int T_BLK = WT * MT * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;

    // z = MRF branch (heaviest first) or ConvTranspose phase; compile-time indices keep the kernarg in SGPRs
    // blockIdx.x -> (window item, sub-block).  Blocks that stage the SAME input window -- the C_out blocks of a
    // tile and, for a ConvTranspose, its u output phases -- are `n_share` consecutive sub-blocks of one window item.
    // With xcd_group they are given the same XCD (blockIdx.x % 8) and neighbouring slots, so the window is fetched
    // from HBM once and re-read from that XCD's L2 (measured without it: the phase blocks of an upsample launch
    // fetched the input u times, 1.5 GB instead of 0.2 GB per launch at configs[2]).
    // (Round 4: the batch index is folded into the item index of a grouped launch.  Before, grouping needed >= 64 tiles
    //  PER BATCH ITEM, which configs[2] -- 32 items of 32 tiles -- does not have: its first two upsamplers fetched their
    //  input 8x and the C = 256 MRF steps 2x, profiles/r04zz_bf16_c3_hbm_traffic.json.)
    int w_item, sub;
    int b = blockIdx.y;
    if (a.xcd_group) {
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
        sub = s % a.n_share;
        const int w_lin = (s / a.n_share) * 8 + xcd;
        if (w_lin >= a.n_items * (a.fold_b ? a.B : 1)) return;
        if (a.fold_b) { b = w_lin / a.n_items; w_item = w_lin - b * a.n_items; }
        else w_item = w_lin;
    } else {
        sub = blockIdx.x % a.n_share;
        w_item = blockIdx.x / a.n_share;
    }
    // Problems of different cost in one launch (the MRF branches, k = 3 / 7 / 11): the longest jobs go first over the
    // whole grid -- problem index in blockIdx.y, which is dispatched slowest -- so that the launch ends on short jobs.
    // (Measured: -2 % on the C = 256 MRF launches of configs[2].)
    int tile_co, tile_t, zr;
    if (a.z_is_phase)   { zr = sub / a.n_co_blk; tile_co = sub - zr * a.n_co_blk; tile_t = w_item; }
    else if (a.z_in_y && a.fold_b) { tile_co = sub; tile_t = w_item; zr = blockIdx.y; }
    else if (a.z_in_y)  { tile_co = sub; tile_t = w_item; zr = blockIdx.y / a.B; b = blockIdx.y - zr * a.B; }
    else                { tile_co = sub; zr = w_item % a.nz; tile_t = w_item / a.nz; }
    const int z = a.z_is_phase ? zr : a.nz - 1 - zr;
    const int pz = a.z_is_phase ? 0 : z;
    Problem p = a.p[0];
    if (pz == 1) p = a.p[1];
    if (pz == 2) p = a.p[2];
    if (pz == 3) p = a.p[3];
    const int out_off = a.out_off + (a.z_is_phase ? z : 0);

    const int i0 = tile_t * T_BLK;
    const int ks = p.ks;
    const int R = T_BLK + (ks - 1) * p.dil;
    const int in_row0 = i0 - p.pad_left;
    const int ct0 = (tile_co * WC + wc) * NT;        // this wave's first 32-wide C_out tile
    const bool wave_active = ct0 < a.n_ct;           // wave-uniform

    // the accumulators start at the bias (the MFMA's C input): (nt, r) -> channel (ct0+nt)*32 + 8*(r>>2) + 4*hi + (r&3)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = (ct0 + nt) * 32 + 8 * g + 4 * hi;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + (co < a.C_out ? co : 0));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][nt][4 * g + e] = b4[e];
        }

    const unsigned q_bytes = (a.ablate & 16) ? 0u : (unsigned)a.n_ct * 1024u;   // bytes per (tap, 16-channel step)
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;                        // (ablation 16: weights from one address)
    const char* wbase = (const char*)p.wp + (a.z_is_phase ? (size_t)z * a.phase_wp_bytes : 0);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wbase, (a.ablate & 16) ? 65536u : (unsigned)ks * tap_bytes);
    const unsigned wvoff = (unsigned)ct0 * 1024u + (unsigned)lane * 16u;
    const char* a_lane = lds + (wt * MT * 32 + lo) * SB + hi * 16;
    const int dil_bytes = p.dil * SB;

    for (int c0 = 0; c0 < a.C_in; c0 += CIC) {
        if (c0 > 0) __syncthreads();
        stage_window<CIC>(a, p, lds, b, in_row0, R, c0);
        __syncthreads();
        if (wave_active && !(a.ablate & 2))
            mma_chunk<MT, NT, CIC>(acc, a_lane, dil_bytes, wr, wvoff, q_bytes, tap_bytes,
                                   (unsigned)(c0 >> 4) * q_bytes, ks);
    }
    // Epilogue.  In the D = W x X^T layout a lane owns 4 channels of one row: storing from there would be 8-byte
    // pieces 2*C bytes apart (measured: a third of the kernel).  Each wave therefore turns its 32-row m-tiles
    // through a private LDS scratch (fp32, bias added, [row][channel]) and reads them back as 16-byte bf16
    // pieces that are contiguous in the channels-last row: residual loads and stores are fully coalesced.
    // (acc + bias) + residual is formed in fp32 and rounded to bf16 once.  The scratch aliases the input
    // window, so every wave must be done with the window first.
    __syncthreads();
    if (!wave_active) return;
    constexpr int RS = NT * 32 * 4 + 16;               // scratch row stride (bytes) = 16 * odd
    constexpr int PPRO = NT * 4;                       // 16-byte bf16 pieces per row of this wave's channel span
    constexpr int NP = 2 * NT;                         // pieces per lane and m-tile
    char* scr = lds + wave * (32 * RS);
    const unsigned out_bytes = (unsigned)a.L_out * (unsigned)a.C_out * 2u;
    const size_t ob = (size_t)b * a.L_out * a.C_out;
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + ob, out_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res + ob : p.y, (p.res && !(a.ablate & 8)) ? out_bytes : 0u);
    // this lane's pieces of an m-tile: q = j*64 + lane -> row q / PPRO, channels 8*(q % PPRO) .. +7 of the span
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
            const int im = i0 + (wt * MT + m) * 32 + row_l;
            const int o = im * a.out_stride + out_off;
            const bool ok = im < a.n_idx && o >= 0 && o < a.L_out && co < a.C_out;
            pvoff[m][j] = ok ? (unsigned)(o * a.C_out + co) * 2u : kOob;
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
            __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)((a.ablate & 4) ? kOob : pvoff[m][j]), 0, 0);
        asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 stores (see mrf_conv_mfma_f32.h)
        __builtin_amdgcn_sched_barrier(0);
        // the store data must stay live until every store of the group has issued (see the store-data note in
        // mrf_conv_mfma_f32.h: hipcc otherwise re-uses a store's data VGPRs right behind it)
#pragma unroll
        for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the scratch is rewritten by the next m-tile
    }
}

// ---- conv_post for bf16 inputs (hifigan_pretrained.py:139-141): fp32 sliding dot product --------------
struct PostLaunch {
    const uint16_t* x[kMaxGroup];  // n_in bf16 inputs [B, L, C]; input = lrelu((x[0]+...) * (1/n_in))
    int n_in;
    const float* w;                // [k][C] fp32
    const float* bias;             // [1]
    float* y;                      // [B, L] fp32
    int B, L, C, k;
    float slope, inv_n;
};

constexpr int kPostTile = 256;

__global__ void __launch_bounds__(256) conv_post_tanh_bf16_kernel(const PostLaunch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_b16[];
    float* lds = reinterpret_cast<float*>(lds_b16);
    const int C = a.C, k = a.k, pad = (k - 1) / 2;
    const int S = C | 1;
    const int PPR = C / 8;
    const int b = blockIdx.y, t0 = blockIdx.x * kPostTile;
    const int R = kPostTile + k - 1;
    const unsigned tensor_bytes = (unsigned)a.L * (unsigned)C * 2u;
    const size_t boff = (size_t)b * a.L * C;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(a.x[0] + boff, tensor_bytes);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc((a.n_in > 1 ? a.x[1] : a.x[0]) + boff, a.n_in > 1 ? tensor_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc((a.n_in > 2 ? a.x[2] : a.x[0]) + boff, a.n_in > 2 ? tensor_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r3 = make_rsrc((a.n_in > 3 ? a.x[3] : a.x[0]) + boff, a.n_in > 3 ? tensor_bytes : 0u);
    const int total = R * PPR;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int r = idx / PPR, pc = idx - r * PPR;
        const int row = t0 - pad + r;
        const unsigned voff = (row >= 0 && row < a.L) ? (unsigned)(row * C + 8 * pc) * 2u : kOob;
        const u32x4 v0 = buf_load4(r0, voff, 0), v1 = buf_load4(r1, voff, 0), v2 = buf_load4(r2, voff, 0),
                    v3 = buf_load4(r3, voff, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo = ((bf_lo(v0[e]) + bf_lo(v1[e])) + bf_lo(v2[e])) + bf_lo(v3[e]);
            float hi = ((bf_hi(v0[e]) + bf_hi(v1[e])) + bf_hi(v2[e])) + bf_hi(v3[e]);
            if (a.n_in > 1) { lo *= a.inv_n; hi *= a.inv_n; }
            lds[r * S + 8 * pc + 2 * e] = lrelu1(lo, a.slope);
            lds[r * S + 8 * pc + 2 * e + 1] = lrelu1(hi, a.slope);
        }
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= a.L) return;
    const float* __restrict__ w = a.w;
    float acc = a.bias[0];
    for (int kap = 0; kap < k; ++kap) {
        const float* row = lds + (threadIdx.x + kap) * S;
        for (int c = 0; c < C; ++c) acc = fmaf(row[c], w[kap * C + c], acc);
    }
    a.y[(size_t)b * a.L + t] = tanhf(acc);
}

inline hipError_t launch_conv_post_bf16(const PostLaunch& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)(kPostTile + a.k - 1) * (a.C | 1) * sizeof(float);
    if (lds_bytes > 160 * 1024 || (a.C & 7)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((a.L + kPostTile - 1) / kPostTile), (unsigned)a.B), block(256);
    { const hipError_t e__ = ::iris::launch_kernel(conv_post_tanh_bf16_kernel, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; }
    return hipSuccess;       
}

// ---- host side: weight packing ---------------------------------------------------------------------
inline uint16_t f32_to_bf16(float f) {   // round to nearest even; weights are finite
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// packed[(((kap*Qp + q)*n_ct + ct)*64 + lane)*8 + e]
//     = bf16(W[co = ct*32 + (lane&31)][ci = 16q + 8*(lane>>5) + e][kap])      (0 outside C_out / C_in)
// Qp = 16-channel steps padded to a multiple of 4 (64 channels) so that any chunking stays inside.
inline int packed_qsteps(int C_in) { return ((C_in + 63) / 64) * 4; }
inline int packed_cotiles(int C_out) { return (C_out + 31) / 32; }
inline size_t packed_conv1d_halfs(int C_in, int C_out, int ks) {
    return (size_t)ks * packed_qsteps(C_in) * packed_cotiles(C_out) * 64 * 8;
}
// w: reference Conv1d layout [C_out][C_in][ks] (hifigan_pretrained.py:50-57)
inline void pack_conv1d_bf16(const float* w, int C_in, int C_out, int ks, uint16_t* out) {
    const int Qp = packed_qsteps(C_in), n_ct = packed_cotiles(C_out);
    for (int kap = 0; kap < ks; ++kap)
        for (int q = 0; q < Qp; ++q)
            for (int ct = 0; ct < n_ct; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = ct * 32 + (lane & 31);
                        const int ci = 16 * q + 8 * (lane >> 5) + e;
                        float v = 0.f;
                        if (co < C_out && ci < C_in) v = w[((size_t)co * C_in + ci) * ks + kap];
                        out[((((size_t)kap * Qp + q) * n_ct + ct) * 64 + lane) * 8 + e] = f32_to_bf16(v);
                    }
}
// ConvTranspose1d weights [C_in][C_out][k] (hifigan_pretrained.py:101-107) as u phase convolutions with
// ceil(k/u) taps each; tap kap of phase ph holds w[:, :, ph + (taps-1-kap)*u] (see conv_mfma_f32.h).
inline int convt_taps(int k, int u) { return (k + u - 1) / u; }
inline size_t packed_convt_phase_halfs(int C_in, int C_out, int k, int u) {
    return packed_conv1d_halfs(C_in, C_out, convt_taps(k, u));
}
inline void pack_convt_bf16(const float* w, int C_in, int C_out, int k, int u, uint16_t* out) {
    const int taps = convt_taps(k, u);
    const int Qp = packed_qsteps(C_in), n_ct = packed_cotiles(C_out);
    const size_t phase_halfs = packed_convt_phase_halfs(C_in, C_out, k, u);
    for (int ph = 0; ph < u; ++ph)
        for (int kap = 0; kap < taps; ++kap) {
            const int kk = ph + (taps - 1 - kap) * u;
            for (int q = 0; q < Qp; ++q)
                for (int ct = 0; ct < n_ct; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = ct * 32 + (lane & 31);
                            const int ci = 16 * q + 8 * (lane >> 5) + e;
                            float v = 0.f;
                            if (co < C_out && ci < C_in && kk < k) v = w[((size_t)ci * C_out + co) * k + kk];
                            out[ph * phase_halfs + ((((size_t)kap * Qp + q) * n_ct + ct) * 64 + lane) * 8 + e] =
                                f32_to_bf16(v);
                        }
        }
}

// ---- launch ----------------------------------------------------------------------------------------
struct Tile { int WT, WC, MT, NT, CIC, MINB, T_BLK, CO_BLK; };

inline Tile pick_tile(int C_in, int C_out, bool phases = false) {
    const int v32 = IRIS_DIAG_ENV("IRIS_B16_TILE32", 0), v64 = IRIS_DIAG_ENV("IRIS_B16_TILE64", 0), v128 = IRIS_DIAG_ENV("IRIS_B16_TILE128", 0);
    Tile t;
    if (C_out <= 32) {
        t.WT = 4; t.WC = 1; t.NT = 1;
        // default: 384 rows at <= 128 VGPRs -> four blocks per CU (HBM-bound layers want requests in flight);
        // diagnostics: v32 = 8 -> 512 rows / 3 blocks, else bit0 -> MT=2, bits1.. -> min blocks 2/3/4
        if (v32 == 0)      { t.MT = phases ? IRIS_B16_UPS32_MT : 3; t.MINB = 4; }
        else if (v32 == 8) { t.MT = 4; t.MINB = 2; }
        else               { t.MT = (v32 & 1) ? 2 : 4; t.MINB = 2 + (v32 >> 1); }
    } else if (C_out <= 64) {
        // default: 192 rows x 64 channels as 2 x 2 waves of 96 x 32 at <= 128 VGPRs -> four blocks per CU (2-3 %
        // ahead of 256 x 64 at three blocks); diagnostics: v64 = 8 -> 256 x 64, else bit0 -> MT=1, bits1.. -> min blocks
        if (v64 == 0)      { t.WT = 2; t.WC = 2; t.MT = 3; t.NT = 1; t.MINB = 4; }
        else               { t.WT = 4; t.WC = 1; t.NT = 2; t.MT = (v64 & 1) ? 1 : 2; t.MINB = v64 == 8 ? 2 : 2 + (v64 >> 1); }
    } else {
        t.WT = 2; t.WC = 2; t.NT = 2;
        t.MT = 2;             // (96 rows per wave, MT = 3: -1.5 % on the C = 256 steps of configs[2], -15 % at 1 x 1000: profiles/r04_notes.md section 15)
        t.MINB = 2;
    }
    t.CIC = (C_in <= 32 && t.NT == 1 && t.WC == 1) ? 32 : 64;
    // wide layers: 128-channel chunks halve the staging round trips of a block (window 48 KB, three blocks per CU)
    if (C_out > 64 && !(v128 & 2) && C_in % 128 == 0) t.CIC = 128;
    t.T_BLK = t.WT * t.MT * 32;
    t.CO_BLK = t.WC * t.NT * 32;
    return t;
}

// Fills the derived fields of `a` and launches.  `nz` = problems or phases.
inline hipError_t launch_conv_bf16(Launch& a, int nz, hipStream_t stream) {
    const Tile t = pick_tile(a.C_in, a.C_out, a.z_is_phase != 0);
    a.n_co_blk = (a.C_out + t.CO_BLK - 1) / t.CO_BLK;
    a.Qp = packed_qsteps(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    a.nz = nz;
    a.inv_n_mrf = a.n_mrf > 0 ? 1.0f / (float)a.n_mrf : 1.0f;
    a.ablate = IRIS_DIAG_ENV("IRIS_B16_ABLATE", 0);
    int span = 0;
    const int np = a.z_is_phase ? 1 : nz;
    if (np > kMaxGroup) return hipErrorInvalidValue;
    for (int j = 0; j < np; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
    }
    // 32-bit byte offsets inside one batch item's tensor (buffer descriptors)
    if ((double)a.L_in * a.C_in * 2.0 >= 2147483648.0 || (double)a.L_out * a.C_out * 2.0 >= 2147483648.0)
        return hipErrorInvalidValue;
    if (!a.x_f32_cf && (a.C_in & 7)) return hipErrorInvalidValue;
    if (a.C_out & 3) return hipErrorInvalidValue;
    const size_t window_bytes = (size_t)(t.T_BLK + span) * (t.CIC * 2 + 16);
    const size_t scratch_bytes = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);      // epilogue transpose, aliases the window
    const size_t lds_bytes = window_bytes > scratch_bytes ? window_bytes : scratch_bytes;
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const int n_t = (a.n_idx + t.T_BLK - 1) / t.T_BLK;
    a.z_in_y = !a.z_is_phase && nz > 1 && (long long)a.B * nz <= 65535 && IRIS_DIAG_ENV("IRIS_B16_ZMAJOR", 1);
    a.n_items = (a.z_is_phase || a.z_in_y) ? n_t : n_t * nz;
    a.n_share = a.z_is_phase ? a.n_co_blk * nz : a.n_co_blk;
    const int xcd_env = IRIS_DIAG_ENV("IRIS_B16_XCDGROUP", 1);
    // grouped launches carry the batch index in blockIdx.x (item = b * n_items + tile; blockIdx.y is the z index alone
    // when z_in_y, else 1)
    a.fold_b = IRIS_B16_XCD_FOLD;
    const long long items_all = (long long)a.n_items * (a.fold_b ? a.B : 1);
    const long long gx_grouped = ((items_all + 7) / 8) * 8 * a.n_share;
    a.xcd_group = xcd_env && a.n_share > 1 && items_all >= 64 && gx_grouped < 0x7fffffffLL;   // (a few items would leave XCDs without work)
    const long long gx = a.xcd_group ? gx_grouped : (long long)a.n_items * a.n_share;
    if (gx > 0x7fffffffLL) return hipErrorInvalidValue;
    a.fold_b = a.fold_b && a.xcd_group;
    const unsigned gy = a.fold_b ? (unsigned)(a.z_in_y ? nz : 1) : (unsigned)(a.z_in_y ? a.B * nz : a.B);
    dim3 grid((unsigned)gx, gy, 1u), block(256);
#define IRIS_B16_LAUNCH(...)                                                                      \
    do {                                                                                          \
        auto kfn = __VA_ARGS__;                                                                   \
        { const hipError_t e__ = ::iris::launch_kernel_named(#__VA_ARGS__, kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
        return hipSuccess;                                                                        \
    } while (0)
#define IRIS_B16_CASE(WT_, WC_, MT_, NT_, CIC_, MINB_)                                            \
    if (t.WT == WT_ && t.WC == WC_ && t.MT == MT_ && t.NT == NT_ && t.CIC == CIC_ && t.MINB == MINB_) \
        IRIS_B16_LAUNCH(conv_mfma_bf16_kernel<WT_, WC_, MT_, NT_, CIC_, MINB_>)
    IRIS_B16_CASE(4, 1, 4, 1, 32, 2); IRIS_B16_CASE(4, 1, 4, 1, 32, 3); IRIS_B16_CASE(4, 1, 4, 1, 32, 4);
    IRIS_B16_CASE(4, 1, 3, 1, 32, 4); IRIS_B16_CASE(4, 1, 3, 1, 64, 4); IRIS_B16_CASE(4, 1, 1, 1, 64, 4);
    IRIS_B16_CASE(4, 1, 2, 1, 32, 2); IRIS_B16_CASE(4, 1, 2, 1, 32, 3); IRIS_B16_CASE(4, 1, 2, 1, 32, 4);
    IRIS_B16_CASE(4, 1, 4, 1, 64, 2); IRIS_B16_CASE(4, 1, 4, 1, 64, 3); IRIS_B16_CASE(4, 1, 4, 1, 64, 4);
    IRIS_B16_CASE(4, 1, 2, 1, 64, 2); IRIS_B16_CASE(4, 1, 2, 1, 64, 3); IRIS_B16_CASE(4, 1, 2, 1, 64, 4);
    IRIS_B16_CASE(4, 1, 2, 2, 64, 2); IRIS_B16_CASE(4, 1, 2, 2, 64, 3); IRIS_B16_CASE(4, 1, 2, 2, 64, 4);
    IRIS_B16_CASE(4, 1, 1, 2, 64, 2); IRIS_B16_CASE(4, 1, 1, 2, 64, 3); IRIS_B16_CASE(4, 1, 1, 2, 64, 4);
    IRIS_B16_CASE(2, 2, 2, 2, 64, 2); IRIS_B16_CASE(2, 2, 2, 2, 128, 2); IRIS_B16_CASE(2, 2, 3, 1, 64, 4);
#undef IRIS_B16_CASE
#undef IRIS_B16_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace b16
}  // namespace iris
