// conv_mfma_f32.h -- the Conv1d / ConvTranspose1d kernel of the HiFiGAN generator for gfx950.
//
// One kernel family covers 77 of the generator's 78 convolutions
// (reference: HiFiGANModel.forward, src/iris/hifigan_pretrained.py:123-143):
//   conv_pre           Conv1d(80->512,k7)                       hifigan_pretrained.py:92-94,124
//   ups[i]             LeakyReLU + ConvTranspose1d              hifigan_pretrained.py:97-109,127-128
//   resblocks[*]       LeakyReLU + dilated Conv1d (+residual)   hifigan_pretrained.py:64-71
// (conv_post, C_out = 1, is a sliding dot product: conv_post.h).
//
// Formulation.  Activations are channels-last [B, L, C] fp32 in HBM, so one time step is one
// contiguous, 16-byte aligned row of C floats whatever the halo offset is.  A convolution is the
// implicit GEMM
//     Y[t, co] = bias[co] + sum_{kappa, ci} act(X)[t - pad + kappa*dil, ci] * W[co, ci, kappa]
// with M = time, N = C_out, K = ks * C_in, computed with v_mfma_f32_32x32x2_f32: exact fp32
// (bitwise an fmaf chain, cdna_hip_programming.md section 3), at the fp32 peak of the chip.
// A ConvTranspose1d with stride u is u such convolutions ("phases") with ceil(k/u) taps each whose
// output rows interleave:  o = i*u + phase - p.
//
// Work split.  A 256-thread block (4 waves) owns a tile of  (WT*MT*32) time rows x (WC*32) output
// channels.  The input window (tile + dilated-tap halo) of CIC input channels is staged once into
// LDS with the input activation applied on the way (LeakyReLU, or the MRF mean of the previous
// stage followed by LeakyReLU, hifigan_pretrained.py:127,137), then every tap re-reads it at a
// shifted row: the dilated tap window lives in LDS, HBM sees each input row once per block.
//   A fragment (time x ci): one ds_read_b128 per 4 MFMAs; row stride CIC+4 floats = 4*odd, which
//                           makes both the b128 reads and the b128 staging writes conflict-free.
//   B fragment (ci x co):   one 16-byte global load per 4 MFMAs from weights repacked on the host
//                           into fragment order (pack_conv1d_weights), served by L2.
//   Accumulators:           MT tiles of 32x32 (16 VGPRs each).
// blockIdx.x % nz selects one of up to 8 independent problems of identical shape (the MRF branches of
// one stage, kernel sizes 3/7/11: three ResBlocks advance in one launch) or the ConvTranspose phase.
#pragma once
#include "device_info.h"
#ifndef IRIS_CONV_WEIGHT_RING
#define IRIS_CONV_WEIGHT_RING 4   // groups the generic conv kernel's weight fragments run ahead of their use (A/B builds: 2)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "diag_env.h"

namespace iris {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum InAct : int { IN_ACT_NONE = 0, IN_ACT_LRELU = 1, IN_ACT_MRF_LRELU = 2 };

constexpr int kMaxGroup = 8;

struct ConvProblem {
    const float* x;     // input  [B, L_in, C_in]   (or [B, C_in, L_in] when x_channels_first)
    const f32x4* wp;    // packed weights, see pack_conv1d_weights
    const float* bias;  // [C_out]
    const float* res;   // residual added in the epilogue [B, L_out, C_out], or nullptr
    float* y;           // output [B, L_out, C_out]
    int ks;             // taps
    int dil;            // dilation (rows between taps)
    int pad_left;       // input row of tap 0 for output row-index i is  i - pad_left
    int reserved;
    const f32x4* wp16;  // the same weights packed for the small-problem kernel (pack_conv1d_weights16), or nullptr
};

struct ConvLaunch {
    ConvProblem p[kMaxGroup];
    const float* xmrf[kMaxGroup];  // in_act == IN_ACT_MRF_LRELU: input = lrelu((xmrf[0]+...)/n_mrf)
    int n_mrf;
    int B, L_in, L_out, C_in, C_out;
    int n_idx;           // number of output row-indices (L_out for conv, L_in + taps - 1 for a phase)
    int out_stride;      // output row o = i*out_stride + out_off (+ phase)
    int out_off;
    int z_is_phase;      // blockIdx.z is a ConvTranspose phase of problem 0 (else a problem index)
    int64_t phase_wp_stride;  // f32x4 elements between the packed weights of consecutive phases
    int in_act;
    int x_channels_first;
    float slope;
    int out_act;         // 0 none, 1 tanh applied to (acc + bias) before the residual is added (PostNet)
    float* sum_y;        // MRF kernel only: when set, the launch is the last conv step of a stage and stores
                         // mean_j(branch output j) here instead of the per-branch outputs
    float sum_div;       // num_kernels as float
    unsigned long long* dbg;  // diagnostics only (stamp builds): 7 accumulators, else nullptr
    int ablate;          // diagnostics only (env IRIS_HIFIGAN_ABLATE): 1 skip staging, 2 weights from one
                         // address, 4 skip epilogue stores, 8 skip residual read.  Results are wrong.
    int z_serial;        // 1: every block loops over all nz_serial problems of its tile (equal-cost blocks)
    int nz_serial;
    int nz;              // problems (or phases) interleaved along blockIdx.x
    int n_co_blk;        // blocks along C_out
    int Gp;              // padded number of 8-channel groups in the packed weights
    int n_ct;            // number of 32-wide C_out tiles in the packed weights
    unsigned* dyn_counter;  // MRF kernel: when set (zero at launch), blocks take their 2nd, 3rd, ... tile from this
                            // counter instead of a fixed stride (large batches: evens out slow and fast CUs)
    int stagger, stagger_mod;  // diagnostics only (IRIS_HIFIGAN_STAGGER): blocks of the second residency generation sleep first
    int zb1, zb2;           // MRF kernel, one-branch-per-block mode: blocks [0, zb1) serve branch 2 (k = 11),
                            // [zb1, zb2) branch 1 (k = 7), [zb2, gridDim.x) branch 0 (k = 3)
};

__device__ __forceinline__ float lrelu1(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float slope) {
    f32x4 r;
    r.x = lrelu1(v.x, slope); r.y = lrelu1(v.y, slope);
    r.z = lrelu1(v.z, slope); r.w = lrelu1(v.w, slope);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Stage rows [in_row0, in_row0 + R) x channels [c0, c0 + CIC) of the (activated) input into LDS.
template <int CIC>
__device__ __forceinline__ void stage_input(const ConvLaunch& a, const ConvProblem& p, float* lds,
                                            int b, int in_row0, int R, int c0) {
    constexpr int S = CIC + 4;
    constexpr int QPR = CIC / 4;  // 16-byte quads per LDS row
    const int tid = threadIdx.x;
    const int L_in = a.L_in, C_in = a.C_in;
    if (!a.x_channels_first && (C_in & 3) == 0) {
        const int total = R * QPR;
        for (int idx = tid; idx < total; idx += 256) {
            const int r = idx / QPR, q = idx - r * QPR;
            const int row = in_row0 + r, ci = c0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row >= 0 && row < L_in && ci < C_in) {
                const size_t off = ((size_t)b * L_in + row) * C_in + ci;
                if (a.in_act == IN_ACT_MRF_LRELU) {
                    v = *reinterpret_cast<const f32x4*>(a.xmrf[0] + off);
                    for (int j = 1; j < a.n_mrf; ++j)
                        v += *reinterpret_cast<const f32x4*>(a.xmrf[j] + off);
                    v = v / (float)a.n_mrf;  // true division, hifigan_pretrained.py:137
                    v = lrelu4(v, a.slope);
                } else {
                    v = *reinterpret_cast<const f32x4*>(p.x + off);
                    if (a.in_act == IN_ACT_LRELU) v = lrelu4(v, a.slope);
                }
            }
            *reinterpret_cast<f32x4*>(lds + r * S + 4 * q) = v;
        }
    } else {
        // scalar path: channels-first input (the mel, hifigan_pretrained.py:228) or C_in % 4 != 0
        const int total = R * CIC;
        if (a.x_channels_first && a.in_act != IN_ACT_MRF_LRELU) {
            // the mel: eight loads in flight per thread instead of one (conv_pre is latency-bound: 16-64 blocks)
            constexpr int U = 8;
            for (int base = tid; base < total; base += 256 * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * 256;
                    const int c = idx / R, r = idx - c * R;         // lanes run along time
                    const int row = in_row0 + r, ci = c0 + c;
                    const bool ok = idx < total && row >= 0 && row < L_in && ci < C_in;
                    v[u] = ok ? p.x[((size_t)b * C_in + ci) * L_in + row] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * 256;
                    if (idx < total) {
                        const int c = idx / R, r = idx - c * R;
                        lds[r * S + c] = a.in_act == IN_ACT_LRELU ? lrelu1(v[u], a.slope) : v[u];
                    }
                }
            }
            return;
        }
        for (int idx = tid; idx < total; idx += 256) {
            int r, c;
            if (a.x_channels_first) { c = idx / R; r = idx - c * R; }   // lanes run along time
            else                    { r = idx / CIC; c = idx - r * CIC; }
            const int row = in_row0 + r, ci = c0 + c;
            float v = 0.f;
            if (row >= 0 && row < L_in && ci < C_in) {
                const size_t off = a.x_channels_first ? ((size_t)b * C_in + ci) * L_in + row
                                                      : ((size_t)b * L_in + row) * C_in + ci;
                if (a.in_act == IN_ACT_MRF_LRELU) {
                    v = a.xmrf[0][off];
                    for (int j = 1; j < a.n_mrf; ++j) v += a.xmrf[j][off];
                    v = lrelu1(v / (float)a.n_mrf, a.slope);
                } else {
                    v = p.x[off];
                    if (a.in_act == IN_ACT_LRELU) v = lrelu1(v, a.slope);
                }
            }
            lds[r * S + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// KS > 0: taps known at compile time (fully unrolled);  KS == 0: runtime tap count.
template <int KS, int WT, int WC, int MT, int CIC>
__device__ __forceinline__ void conv_body(const ConvLaunch& a, const ConvProblem& p,
                                          const f32x4* __restrict__ wp, int out_off, float* lds) {
    constexpr int S = CIC + 4;
    constexpr int T_BLK = WT * MT * 32;
    const int ks = KS > 0 ? KS : p.ks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int bid = blockIdx.x / a.nz;          // blockIdx.x = (tile index) * nz + z
    const int tile_co = bid % a.n_co_blk, tile_t = bid / a.n_co_blk;
    const int b = blockIdx.y;
    const int i0 = tile_t * T_BLK;
    const int R = T_BLK + (ks - 1) * p.dil;
    const int in_row0 = i0 - p.pad_left;
    const int ct = tile_co * WC + wc;          // this wave's 32-wide C_out tile
    const bool wave_active = ct < a.n_ct;      // wave-uniform
    const int lo = lane & 31, hi = lane >> 5;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const float* aptr = lds + (wt * MT * 32 + lo) * S + 4 * hi;
    const f32x4* wlane = wp + (size_t)ct * 64 + lane;
    const size_t wstep = (a.ablate & 2) ? 0 : (size_t)a.n_ct * 64;  // f32x4 elements per (tap, group)

    for (int c0 = 0; c0 < a.C_in; c0 += CIC) {
        if (c0 > 0) __syncthreads();
        if (!(a.ablate & 1)) stage_input<CIC>(a, p, lds, b, in_row0, R, c0);
        __syncthreads();
        if (wave_active) {
            const int g0 = c0 >> 3;
            // One "group" = 8 input channels of one tap = 4*MT MFMAs.  Fragments of group n+1
            // (LDS) and n+2 (weights, L2) are requested before the MFMAs of group n issue.
            constexpr int GPC = CIC / 8;                       // groups per tap in this chunk
            const int n_groups = ks * GPC;
            auto a_ptr = [&](int n) { const int kk = n / GPC, g = n - kk * GPC;
                                      return aptr + kk * p.dil * S + 8 * g; };
            auto b_ptr = [&](int n) { const int kk = n / GPC, g = n - kk * GPC;
                                      return wlane + ((size_t)kk * a.Gp + g0 + g) * wstep; };
            auto mfma_group = [&](const f32x4 (&av)[MT], const f32x4& bw) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[e], av[m][e], acc[m], 0, 0, 0);
            };
            auto load_a = [&](f32x4 (&av)[MT], int n) {
                const float* ap = a_ptr(n);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const f32x4*>(ap + m * 32 * S);
            };
            if constexpr (KS > 0) {
                constexpr int NG = KS * GPC;
                // Weight fragments run DBW groups ahead in a register ring (round 3: four instead of two: -1.5 us on the first
                // ConvTranspose1d and -0.6 us on conv_pre at 100 frames, neutral at 1000 -- profiles/r03o_conv_weight_ring_ab.txt).
                // sched_barrier(0): hipcc otherwise sinks every load back next to its first use
                // (global_load followed by vmcnt(0)), exposing the latency once per group.
                constexpr int DBW = IRIS_CONV_WEIGHT_RING;
                f32x4 av[2][MT], bwr[DBW + 1];
#pragma unroll
                for (int d = 0; d < DBW; ++d) bwr[d] = *b_ptr(d < NG ? d : NG - 1);
                load_a(av[0], 0);
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    if (n + 1 < NG) load_a(av[(n + 1) & 1], n + 1);
                    if (n + DBW < NG) bwr[(n + DBW) % (DBW + 1)] = *b_ptr(n + DBW);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_group(av[n & 1], bwr[n % (DBW + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int n = 0; n < n_groups; ++n) {
                    f32x4 av[MT];
                    load_a(av, n);
                    const f32x4 bw = *b_ptr(n);
                    mfma_group(av, bw);
                }
            }
        }
    }

    if (!wave_active) return;
    // Epilogue.  The MFMA is issued as D = W_frag x X_frag^T, i.e. D[co][t]: lane = time step
    // (col = lane & 31), registers 4g..4g+3 = the 4 consecutive channels ct*32 + 8g + 4*(lane>>5) + {0..3}:
    // one 16-byte piece of a channels-last row.
    const int i = i0 + wt * MT * 32 + lo;                    // row index of (m = 0); m adds 32
    if ((a.C_out & 3) == 0) {
        // 16-byte path.  Bias/residual are added in place and every store reads registers that are not
        // rewritten afterwards: hipcc re-uses a dwordx4 store's data VGPRs two instructions later, which
        // was observed to corrupt stores under load (see mrf_conv_mfma_f32.h).
        size_t offs[MT];
        bool ok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int im = i + m * 32;
            const int o = im * a.out_stride + out_off;
            ok[m] = im < a.n_idx && o >= 0 && o < a.L_out;
            offs[m] = ((size_t)b * a.L_out + (ok[m] ? o : 0)) * a.C_out + ct * 32 + 4 * hi;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ct * 32 + 8 * g + 4 * hi;
            if (co < a.C_out) {
                const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
                    if (p.res && ok[m] && !(a.ablate & 8)) r4 = *reinterpret_cast<const f32x4*>(p.res + offs[m] + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[m][4 * g + e] + bias4[e];
                        if (a.out_act == 1) v = tanhf(v);
                        acc[m][4 * g + e] = v + r4[e];
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = ct * 32 + 8 * g + 4 * hi;
                if (ok[m] && co < a.C_out && !(a.ablate & 4)) {
                    const f32x4 v = {acc[m][4 * g + 0], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(p.y + offs[m] + 8 * g) = v;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        // scalar path for channel counts that are not a multiple of 4
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int im = i + m * 32;
            const int o = im * a.out_stride + out_off;
            if (!(im < a.n_idx && o >= 0 && o < a.L_out)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < a.C_out) {
                    const size_t off = ((size_t)b * a.L_out + o) * a.C_out + co;
                    float v = acc[m][r] + p.bias[co];
                    if (a.out_act == 1) v = tanhf(v);
                    if (p.res && !(a.ablate & 8)) v += p.res[off];
                    if (!(a.ablate & 4)) p.y[off] = v;
                }
            }
        }
    }
}

template <int WT, int WC, int MT, int CIC>
__device__ __forceinline__ void conv_dispatch(const ConvLaunch& a, const ConvProblem& p, const f32x4* wp,
                                              int out_off, float* lds) {
    switch (p.ks) {
        case 2:  conv_body<2,  WT, WC, MT, CIC>(a, p, wp, out_off, lds); break;
        case 3:  conv_body<3,  WT, WC, MT, CIC>(a, p, wp, out_off, lds); break;
        case 7:  conv_body<7,  WT, WC, MT, CIC>(a, p, wp, out_off, lds); break;
        case 11: conv_body<11, WT, WC, MT, CIC>(a, p, wp, out_off, lds); break;
        default: conv_body<0,  WT, WC, MT, CIC>(a, p, wp, out_off, lds); break;
    }
}

template <int WT, int WC, int MT, int CIC>
__global__ void __launch_bounds__(256) conv_mfma_f32_kernel(const ConvLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (a.z_serial) {
        // Equal-cost blocks: one block runs ALL problems (MRF branches, k = 3/7/11) of its tile one
        // after the other, so every block of the launch costs the same whatever mix of kernel
        // sizes the stage has.
        // The starting branch rotates with the block index (a.z_serial == 2): blocks that share a CU
        // are then in different phases (staging / MFMA / epilogue) instead of marching in lock-step.
        const int rot = a.z_serial == 2 ? (int)(blockIdx.x % a.nz_serial) : 0;
        for (int zi = 0; zi < a.nz_serial; ++zi) {
            int z = a.nz_serial - 1 - zi + rot;
            if (z >= a.nz_serial) z -= a.nz_serial;
            conv_dispatch<WT, WC, MT, CIC>(a, a.p[z], a.p[z].wp, a.out_off, lds);
            if (zi + 1 < a.nz_serial) __syncthreads();
        }
        return;
    }
    // z = MRF branch (or ConvTranspose phase) is the fastest-varying part of blockIdx.x, so blocks
    // that are dispatched together -- and share a CU -- mix the cheap (k=3) and the expensive (k=11)
    // branches; branches are taken heaviest-first (the packed order is k ascending).
    const int zr = blockIdx.x % a.nz;
    const int z = a.z_is_phase ? zr : a.nz - 1 - zr;
    const ConvProblem& p = a.p[a.z_is_phase ? 0 : z];
    const f32x4* wp = p.wp + (a.z_is_phase ? (int64_t)z * a.phase_wp_stride : 0);
    const int out_off = a.out_off + (a.z_is_phase ? z : 0);
    conv_dispatch<WT, WC, MT, CIC>(a, p, wp, out_off, lds);
}

// ---------------------------------------------------------------------------------------------
// Host side: weight repacking into B-fragment order.
//   packed[((kap*Gp + g)*n_ct + ct)*64 + lane] (an f32x4, component e)
//       = W[co = ct*32 + (lane&31)][ci = 8g + 4*(lane>>5) + e][kap]      (0 outside C_out/C_in)
// Gp = groups padded to a multiple of 8 (64 channels) so that any chunking of C_in stays inside.
inline int packed_groups(int C_in) { return ((C_in + 63) / 64) * 8; }
inline int packed_cotiles(int C_out) { return (C_out + 31) / 32; }
inline size_t packed_conv1d_floats(int C_in, int C_out, int ks) {
    return (size_t)ks * packed_groups(C_in) * packed_cotiles(C_out) * 64 * 4;
}

// w: reference Conv1d layout [C_out][C_in][ks] (hifigan_pretrained.py:50-57).
// Taps [kap0, kap1) only (kap1 < 0: all): the pieces are disjoint in `out`, so iris_hifigan_create packs them on several
// host threads (host_parallel.h).
inline void pack_conv1d_weights(const float* w, int C_in, int C_out, int ks, float* out, int kap0 = 0, int kap1 = -1) {
    const int Gp = packed_groups(C_in), n_ct = packed_cotiles(C_out);
    if (kap1 < 0 || kap1 > ks) kap1 = ks;
    for (int kap = kap0; kap < kap1; ++kap)
        for (int g = 0; g < Gp; ++g)
            for (int ct = 0; ct < n_ct; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = ct * 32 + (lane & 31);
                        const int ci = 8 * g + 4 * (lane >> 5) + e;
                        float v = 0.f;
                        if (co < C_out && ci < C_in) v = w[((size_t)co * C_in + ci) * ks + kap];
                        out[((((size_t)kap * Gp + g) * n_ct + ct) * 64 + lane) * 4 + e] = v;
                    }
}

// The same weights in the fragment order of v_mfma_f32_16x16x4_f32 as the small-problem kernel issues it
// (mrf_small_f32.h): C_in, C_out multiples of 16.
//   packed16[((kap*(C_in/16) + gp)*(C_out/16) + ct16)*64 + lane] (an f32x4)
//       = W[co = 16 ct16 + (lane & 15)][ci = 16 gp + {0, 2, 8, 10}[e] + {0, 4, 1, 5}[lane >> 4]][kap]     e = 0..3
// i.e. the four A operands of the four MFMAs that cover 16 input channels, for K-quarter kq = lane >> 4.
inline size_t packed16_conv1d_floats(int C_in, int C_out, int ks) { return (size_t)ks * C_in * C_out; }
inline void pack_conv1d_weights16(const float* w, int C_in, int C_out, int ks, float* out, int kap0 = 0, int kap1 = -1) {
    static const int kq_ch[4] = {0, 4, 1, 5}, e_ch[4] = {0, 2, 8, 10};
    const int ngp = C_in / 16, nct = C_out / 16;
    if (kap1 < 0 || kap1 > ks) kap1 = ks;
    for (int kap = kap0; kap < kap1; ++kap)
        for (int gp = 0; gp < ngp; ++gp)
            for (int ct = 0; ct < nct; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = 16 * ct + (lane & 15);
                        const int ci = 16 * gp + e_ch[e] + kq_ch[lane >> 4];
                        out[((((size_t)kap * ngp + gp) * nct + ct) * 64 + lane) * 4 + e] = w[((size_t)co * C_in + ci) * ks + kap];
                    }
}

// ConvTranspose1d(C_in->C_out, k, stride u, padding p): weight layout [C_in][C_out][k]
// (hifigan_pretrained.py:101-107).  Output o gets  sum_m x[i0 - m] * w[:, :, phase + m*u]  with
// q = o + p, phase = q % u, i0 = q / u.  Phase `ph` is therefore a Conv1d with taps = ceil(k/u),
// dilation 1, pad_left = taps - 1 and tap kap holding w[:, :, ph + (taps-1-kap)*u].
inline int convt_taps(int k, int u) { return (k + u - 1) / u; }
inline size_t packed_convt_phase_floats(int C_in, int C_out, int k, int u) {
    return packed_conv1d_floats(C_in, C_out, convt_taps(k, u));
}
inline void pack_convt_weights(const float* w, int C_in, int C_out, int k, int u, float* out, int ph0 = 0, int ph1 = -1) {
    const int taps = convt_taps(k, u);
    const int Gp = packed_groups(C_in), n_ct = packed_cotiles(C_out);
    const size_t phase_floats = packed_convt_phase_floats(C_in, C_out, k, u);
    if (ph1 < 0 || ph1 > u) ph1 = u;
    for (int ph = ph0; ph < ph1; ++ph)
        for (int kap = 0; kap < taps; ++kap) {
            const int kk = ph + (taps - 1 - kap) * u;  // index into the reference kernel axis
            for (int g = 0; g < Gp; ++g)
                for (int ct = 0; ct < n_ct; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int co = ct * 32 + (lane & 31);
                            const int ci = 8 * g + 4 * (lane >> 5) + e;
                            float v = 0.f;
                            if (co < C_out && ci < C_in && kk < k)
                                v = w[((size_t)ci * C_out + co) * k + kk];
                            out[ph * phase_floats +
                                ((((size_t)kap * Gp + g) * n_ct + ct) * 64 + lane) * 4 + e] = v;
                        }
        }
}

// ---------------------------------------------------------------------------------------------
// Launch: picks the tile shape from C_out / C_in.
struct ConvTile { int WT, WC, MT, CIC, T_BLK, CO_BLK; };

inline ConvTile pick_tile(int C_in, int C_out) {
    ConvTile t;
    t.MT = 2;
    if (C_out <= 32)      { t.WT = 4; t.WC = 1; }
    else if (C_out <= 64) { t.WT = 2; t.WC = 2; }
    else                  { t.WT = 1; t.WC = 4; }
    t.CIC = (C_in <= 32 && t.WC == 1) ? 32 : 64;
    t.T_BLK = t.WT * t.MT * 32;
    t.CO_BLK = t.WC * 32;
    return t;
}

#ifndef IRIS_KERNELS_ONLY
// Fills the derived fields of `a` (n_co_blk, Gp, n_ct) and launches. `nz` = problems or phases.
inline hipError_t launch_conv(ConvLaunch& a, int nz, hipStream_t stream) {
    const ConvTile t = pick_tile(a.C_in, a.C_out);
    a.n_co_blk = (a.C_out + t.CO_BLK - 1) / t.CO_BLK;
    a.Gp = packed_groups(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    int span = 0;
    const int np = a.z_is_phase ? 1 : nz;
    for (int j = 0; j < np; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
    }
    const int serial_env = IRIS_DIAG_ENV("IRIS_HIFIGAN_ZSERIAL", 1);
    a.z_serial = (!a.z_is_phase && nz > 1 && serial_env) ? serial_env : 0;
    a.nz_serial = nz;
    if (a.z_serial) nz = 1;
    a.nz = nz;
    // Small grids (conv_pre, the first upsamplers, short utterances): a launch costs one tile's serial
    // time, so when the grid cannot give every CU two blocks the tile height is halved (MT = 1).
    const int n_cu = device_cu_count();
    const int mt_env = IRIS_DIAG_ENV("IRIS_HIFIGAN_CONV_MT", 0);
    const long long blocks2 = (long long)((a.n_idx + t.T_BLK - 1) / t.T_BLK) * a.n_co_blk * nz * a.B;
    // (the last ConvTranspose1d -- K = 2 taps x 64 channels, 32 output channels: three quarters of a block's life is window load
    //  and store -- runs better as four 35 KB blocks per CU than as two 70 KB ones: 50 -> 43 us at batch 1 x 1000 frames)
#ifdef IRIS_CONV_LAST_UPS_MT2
    const bool tall_small_k = false;         // (A/B build)
#else
    const bool tall_small_k = a.z_is_phase && t.WT == 4 && a.C_in <= 64;
#endif
    const int MT = mt_env ? mt_env : ((blocks2 < 2LL * n_cu || tall_small_k) ? 1 : 2);
    const int T_BLK = t.WT * MT * 32;
    // conv_pre of the V1 generator (80 mel bins, channels-first -> 512): one 80-channel chunk instead of 64 + 16.  The launch
    // has 64-128 blocks, so its time is one block's serial time, and each chunk costs a staging round trip and two barriers.
    const int CIC = (a.C_in == 80 && a.x_channels_first && t.WT == 1) ? 80 : t.CIC;
    const size_t lds_bytes = (size_t)(T_BLK + span) * (CIC + 4) * sizeof(float);
    const int n_t = (a.n_idx + T_BLK - 1) / T_BLK;
    a.ablate = IRIS_DIAG_ENV("IRIS_HIFIGAN_ABLATE", 0);
    dim3 grid((unsigned)(n_t * a.n_co_blk * nz), (unsigned)a.B, 1u), block(256);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
#define IRIS_LAUNCH_K(...)                                                                        \
    do {                                                                                          \
        auto kfn = __VA_ARGS__;                                                                   \
        { const hipError_t e__ = ::iris::launch_kernel_named(#__VA_ARGS__, kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
    } while (0)
#define IRIS_LAUNCH(WT_, WC_, CIC_)                                                               \
    do { if (MT == 1) IRIS_LAUNCH_K(conv_mfma_f32_kernel<WT_, WC_, 1, CIC_>);                     \
         else         IRIS_LAUNCH_K(conv_mfma_f32_kernel<WT_, WC_, 2, CIC_>); } while (0)
    if (t.WT == 4 && CIC == 32)        IRIS_LAUNCH(4, 1, 32);
    else if (t.WT == 4)                IRIS_LAUNCH(4, 1, 64);
    else if (t.WT == 2)                IRIS_LAUNCH(2, 2, 64);
    else if (CIC == 80)                IRIS_LAUNCH(1, 4, 80);
    else                               IRIS_LAUNCH(1, 4, 64);
#undef IRIS_LAUNCH_K
#undef IRIS_LAUNCH
    return hipSuccess;       // (every launch above has reported its own status)
}
#endif  // IRIS_KERNELS_ONLY

}  // namespace iris
