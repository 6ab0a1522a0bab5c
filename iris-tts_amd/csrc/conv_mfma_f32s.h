// conv_mfma_f32s.h -- MRF ResBlock conv step with fp32 storage and split-bf16 products (dtype IRIS_HIFIGAN_F32_SPLIT).
//
// Same layer as mrf_conv_mfma_f32.h (reference: ResBlock.forward, src/iris/hifigan_pretrained.py:64-71): activations,
// bias, residual and output are fp32 in HBM exactly as in the fp32 path (same workspace), accumulation is fp32.
// Only the products differ: every fp32 operand is split into two bf16 terms, v = hi + mid + O(2^-16 |v|) with
// hi = bf16(v), mid = bf16(v - hi), and a product x*w is formed as  hi*hi + hi*mid + mid*hi  on
// v_mfma_f32_32x32x16_bf16 (three bf16 MFMAs = 3/16 of the fp32 MFMA time per product; the dropped terms are
// O(2^-16) of the product).  Measured on the CPU restatement of this scheme: <= 2.5e-5 max-abs on the waveform
// against the fp32 generator, inside north_star's 1e-4 -- but it is NOT the exact fp32 arithmetic of the parity
// path, so it is an opt-in mode and never the headline.
//
// Structure = conv_mfma_bf16.h: a block owns (WT*MT*32) rows x (WC*NT*32) channels; the activated window is staged
// into LDS as two bf16 planes (hi, mid; same bytes as an fp32 window), weights are packed on the host as two
// planes of MFMA fragments; D = W x X^T; per-wave LDS transpose in the epilogue for coalesced 16-byte fp32 stores.
#pragma once
#include <vector>

#include "conv_mfma_bf16.h"

namespace iris {
namespace s3 {

using b16::bf16x8;
using b16::f32x4;
using b16::f32x16;
using b16::u32x2;
using b16::u32x4;
using b16::pack_bf2;
using b16::bf_lo;
using b16::bf_hi;
using b16::lrelu1;
using b16::make_rsrc;
using b16::buf_load4;
using b16::kOob;

constexpr int kMaxGroup = 4;

struct Problem {
    const float* x;       // fp32 [B, L, C]
    const void* wp;       // packed bf16 fragments: hi plane, then (plane_bytes later) the mid plane
    const float* bias;    // [C] fp32
    const float* res;     // fp32 residual [B, L, C] or nullptr
    float* y;             // fp32 output [B, L, C]
    int ks, dil, pad_left, reserved;
};

struct Launch {
    Problem p[kMaxGroup];
    int B, L, C;          // C_in == C_out
    float slope;
    int nz;               // problems interleaved along blockIdx.x
    int n_co_blk, Qp, n_ct;
    unsigned plane_bytes[kMaxGroup];   // bytes of one weight plane of problem j (= ks * Qp * n_ct * 1024)
    // ConvTranspose1d as u phase convolutions (conv_mfma_f32.h): all zero for a plain Conv1d
    int C_in;             // input channels (0: = C)
    int L_in;             // input rows (0: = L); L is then the OUTPUT length u * L_in
    int n_idx;            // output row-indices per phase (0: = L)
    int out_stride;       // output row o = i * out_stride + out_off + phase   (0: stride 1, offset 0)
    int out_off;
    int z_is_phase;       // the z part of blockIdx.x is the phase of problem 0; its weights start phase_bytes further per phase
    unsigned phase_bytes;
    int ablate;           // diagnostics only (env IRIS_S3_ABLATE): 1 no staging loads, 2 no MFMA loop, 4 no stores, 8 no
                          // residual loads.  Results are wrong.
    float* sum_y;         // set: last conv step of a stage -- a block runs ALL nz branches of its tile in the order
                          // 0, 1, .. (the reference's summation order, hifigan_pretrained.py:131-137) and stores only
                          // their mean here; the per-branch outputs are not written
    float sum_div;        // nz as float
};

// Rows [in_row0, in_row0 + R) x channels [c0, c0 + CIC) of LeakyReLU(x) -> two bf16 planes in LDS.
template <int CIC>
__device__ __forceinline__ void stage_window(const Launch& a, const Problem& p, char* lds_hi, char* lds_mid, int b,
                                             int in_row0, int R, int c0) {
    constexpr int SB = CIC * 2 + 16;
    constexpr int PPR = CIC / 4;          // 16-byte fp32 pieces (4 channels) per row
    const int tid = threadIdx.x;
    const int Lin = a.L_in ? a.L_in : a.L, Cin = a.C_in ? a.C_in : a.C;
    const unsigned tensor_bytes = (unsigned)Lin * (unsigned)Cin * 4u;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + (size_t)b * Lin * Cin, tensor_bytes);
    const int total = R * PPR;
    constexpr int U = 4;
    for (int base = 0; base < total; base += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + tid;
            const int r = idx / PPR, pc = idx - r * PPR;
            const int row = in_row0 + r;
            const bool ok = idx < total && row >= 0 && row < Lin;
            v[u] = buf_load4(xr, (ok && !(a.ablate & 1)) ? (unsigned)(row * Cin + c0 + 4 * pc) * 4u : kOob, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + tid;
            const int r = idx / PPR, pc = idx - r * PPR;
            float f[4], h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned w = v[u][e];      // (bit_cast straight from a vector element reads element 0: copy first)
                f[e] = lrelu1(__builtin_bit_cast(float, w), a.slope);
            }
            u32x2 hi, mid;
            hi[0] = pack_bf2(f[0], f[1]); hi[1] = pack_bf2(f[2], f[3]);
            h[0] = bf_lo(hi[0]); h[1] = bf_hi(hi[0]); h[2] = bf_lo(hi[1]); h[3] = bf_hi(hi[1]);
            mid[0] = pack_bf2(f[0] - h[0], f[1] - h[1]); mid[1] = pack_bf2(f[2] - h[2], f[3] - h[3]);
            if (idx < total) {
                *reinterpret_cast<u32x2*>(lds_hi + r * SB + pc * 8) = hi;
                *reinterpret_cast<u32x2*>(lds_mid + r * SB + pc * 8) = mid;
            }
        }
    }
}

template <int WT, int WC, int MT, int NT, int CIC, int MINB, bool ZS>
__global__ void __launch_bounds__(256, MINB) conv_mfma_f32s_kernel(const Launch a) {
    extern __shared__ __attribute__((aligned(16))) char lds_s3[];
    char* lds = lds_s3;
    constexpr int SB = CIC * 2 + 16;
    constexpr int QPC = CIC / 16;
    constexpr int QL = QPC == 4 ? 2 : 1;
    static_assert(QPC == 4 || QPC == 2, "CIC must be 32 or 64");
    constexpr int T_BLK = WT * MT * 32;
    constexpr int D = 2;                                   // weight ring depth (groups of 3*MT*NT MFMAs)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;

    // blockIdx.x -> (time tile, branch z heaviest first, C_out block); ZS: (time tile, C_out block), all branches here
    const int sub = blockIdx.x % a.n_co_blk, item = blockIdx.x / a.n_co_blk;
    const int nzp = ZS ? 1 : a.nz;
    const int zr = item % nzp, tile_t = item / nzp;
    const int b = blockIdx.y;
    const int i0 = tile_t * T_BLK;
    constexpr int RS = NT * 32 * 4 + 16;
    constexpr int PPRO = NT * 8;                       // 16-byte fp32 pieces per row of this wave's channel span
    constexpr int NP = 4 * NT;                         // pieces per lane and 32-row m-tile
    f32x4 msum[ZS ? MT : 1][ZS ? NP : 1];              // ZS: running sum of the branch outputs, in output-piece layout
    const int n_pass = ZS ? a.nz : 1;
  for (int zi = 0; zi < n_pass; ++zi) {
    const int z = ZS ? zi : (a.z_is_phase ? zr : a.nz - 1 - zr);
    Problem p = a.p[0];
    unsigned plane_bytes = a.plane_bytes[0];
    if (!a.z_is_phase) {
        if (z == 1) { p = a.p[1]; plane_bytes = a.plane_bytes[1]; }
        if (z == 2) { p = a.p[2]; plane_bytes = a.plane_bytes[2]; }
        if (z == 3) { p = a.p[3]; plane_bytes = a.plane_bytes[3]; }
    }
    const int Cin = a.C_in ? a.C_in : a.C;
    const int ostride = a.out_stride ? a.out_stride : 1;
    const int ooff = a.out_off + (a.z_is_phase ? z : 0);
    if (ZS && zi > 0) __syncthreads();               // the previous branch's scratch reads are done
    const int ks = p.ks;
    const int span = (ks - 1) * p.dil;
    const int R = T_BLK + span;
    const int in_row0 = i0 - p.pad_left;
    const int ct0 = (sub * WC + wc) * NT;
    char* const lds_hi = lds;
    char* const lds_mid = lds + R * SB;              // second plane right behind the first (SB is a multiple of 16)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;

    const unsigned q_bytes = (unsigned)a.n_ct * 1024u;
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;
    const unsigned phase_off = a.z_is_phase ? (unsigned)z * a.phase_bytes : 0u;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc((const char*)p.wp + phase_off, plane_bytes + (unsigned)ks * tap_bytes);
    const unsigned wvoff = (unsigned)ct0 * 1024u + (unsigned)lane * 16u;
    const int a_off = (wt * MT * 32 + lo) * SB + hi * 16;
    const int dil_bytes = p.dil * SB;
    const int NG = ks * QPC;

    for (int c0 = 0; c0 < Cin; c0 += CIC) {
        if (c0 > 0) __syncthreads();
        stage_window<CIC>(a, p, lds_hi, lds_mid, b, in_row0, R, c0);
        __syncthreads();
        const unsigned q0_bytes = (unsigned)(c0 >> 4) * q_bytes;
        auto w_soff = [&](int n) -> unsigned {      // groups past the last tap fall outside the hi plane's range -> clamp below
            return (unsigned)(n >> QL) * tap_bytes + q0_bytes + (unsigned)(n & (QPC - 1)) * q_bytes;
        };
        auto load_w = [&](u32x4 (&wh)[NT], u32x4 (&wm)[NT], int n) {
            const bool in = n < NG;                 // (the ring asks for up to D groups beyond the end)
            const unsigned so = w_soff(in ? n : 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wh[nt] = buf_load4(wr, in ? wvoff + (unsigned)nt * 1024u : kOob, so);
                wm[nt] = buf_load4(wr, in ? wvoff + (unsigned)nt * 1024u : kOob, so + plane_bytes);
            }
        };
        auto load_a = [&](u32x4 (&ah)[MT], u32x4 (&am)[MT], int n) {
            int tap = n >> QL;
            tap = tap < ks ? tap : ks - 1;
            const int off = a_off + tap * dil_bytes + (n & (QPC - 1)) * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = *reinterpret_cast<const u32x4*>(lds_hi + off + m * 32 * SB);
                am[m] = *reinterpret_cast<const u32x4*>(lds_mid + off + m * 32 * SB);
            }
        };
        u32x4 wh[D][NT], wm[D][NT], ah[2][MT], am[2][MT];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            load_w(wh[i], wm[i], i);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_a(ah[0], am[0], 0);
        for (int n0 = 0; n0 < ((a.ablate & 2) ? 0 : NG); n0 += D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int n = n0 + i;
                load_a(ah[(i + 1) & 1], am[(i + 1) & 1], n + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        // smallest terms first: (w_mid, x_hi) + (w_hi, x_mid) + (w_hi, x_hi)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wm[i][nt]), __builtin_bit_cast(bf16x8, ah[i & 1][m]), acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wh[i][nt]), __builtin_bit_cast(bf16x8, am[i & 1][m]), acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wh[i][nt]), __builtin_bit_cast(bf16x8, ah[i & 1][m]), acc[m][nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                load_w(wh[i], wm[i], n + D);
            }
        }
    }

    // Epilogue: (acc + bias) -> per-wave fp32 scratch [row][channel] -> 16-byte pieces (4 channels of one row), + residual,
    // coalesced fp32 stores.  The scratch aliases the window: every wave must be done with it first.
    __syncthreads();
    char* const scr = lds + wave * (32 * RS);
    const unsigned tensor_bytes = (unsigned)a.L * (unsigned)a.C * 4u;
    const size_t ob = (size_t)b * a.L * a.C;
    const __amdgpu_buffer_rsrc_t yr = make_rsrc((ZS ? a.sum_y : p.y) + ob, tensor_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res + ob : p.y, (p.res && !(a.ablate & 8)) ? tensor_bytes : 0u);
    unsigned pv[NP];
    int pscr[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = j * 64 + lane;
        const int row_l = q / PPRO, pc = q - row_l * PPRO;
        pscr[j] = row_l * RS + pc * 16;
        // output row o = i * stride + offset: rows < 0 wrap to >= 2^31 and rows >= L exceed num_records (both dropped)
        pv[j] = (unsigned)(((i0 + wt * MT * 32 + row_l) * ostride + ooff) * a.C + ct0 * 32 + 4 * pc) * 4u;
    }
    f32x4 bias4[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + (ct0 + nt) * 32 + 8 * g + 4 * hi);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const unsigned msoff = (unsigned)(m * 32 * ostride * a.C) * 4u;
        u32x4 resv[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) resv[j] = buf_load4(rr, pv[j], msoff);
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
            const f32x4 v = *reinterpret_cast<const f32x4*>(scr + pscr[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned rw = resv[j][e];
                float o = v[e] + __builtin_bit_cast(float, rw);
                if constexpr (ZS) {
                    o = zi == 0 ? o : msum[m][j][e] + o;          // xs = r0; xs += r1; xs += r2
                    msum[m][j][e] = o;
                    if (zi == n_pass - 1) o = o / a.sum_div;      // true division, hifigan_pretrained.py:137
                }
                outp[j][e] = __builtin_bit_cast(unsigned, o);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!ZS || zi == n_pass - 1) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)((a.ablate & 4) ? kOob : pv[j]), (int)msoff, 0);
        }
        asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 stores (see mrf_conv_mfma_f32.h)
        __builtin_amdgcn_sched_barrier(0);
        // store data stays live until every store of the group has issued (store-data note in mrf_conv_mfma_f32.h)
#pragma unroll
        for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------
// Two planes of the bf16 fragment layout of conv_mfma_bf16.h (pack_conv1d_bf16): hi = bf16(w), mid = bf16(w - hi).
inline size_t packed_plane_halfs(int C_in, int C_out, int ks) { return b16::packed_conv1d_halfs(C_in, C_out, ks); }
inline void pack_conv1d_split(const float* w, int C_in, int C_out, int ks, uint16_t* out) {
    const size_t n = (size_t)C_out * C_in * ks;
    std::vector<float> hi(n), mid(n);
    for (size_t i = 0; i < n; ++i) {
        const uint16_t h = b16::f32_to_bf16(w[i]);
        uint32_t u = (uint32_t)h << 16;
        float hf;
        memcpy(&hf, &u, 4);
        hi[i] = hf;
        mid[i] = w[i] - hf;
    }
    b16::pack_conv1d_bf16(hi.data(), C_in, C_out, ks, out);
    b16::pack_conv1d_bf16(mid.data(), C_in, C_out, ks, out + packed_plane_halfs(C_in, C_out, ks));
}

// ConvTranspose1d weights [C_in][C_out][k]: all phases of the hi plane, then all phases of the mid plane
inline size_t packed_convt_plane_halfs(int C_in, int C_out, int k, int u) { return b16::packed_convt_phase_halfs(C_in, C_out, k, u) * u; }
inline void pack_convt_split(const float* w, int C_in, int C_out, int k, int u, uint16_t* out) {
    const size_t n = (size_t)C_in * C_out * k;
    std::vector<float> hi(n), mid(n);
    for (size_t i = 0; i < n; ++i) {
        const uint16_t h = b16::f32_to_bf16(w[i]);
        uint32_t bits = (uint32_t)h << 16;
        float hf;
        memcpy(&hf, &bits, 4);
        hi[i] = hf;
        mid[i] = w[i] - hf;
    }
    b16::pack_convt_bf16(hi.data(), C_in, C_out, k, u, out);
    b16::pack_convt_bf16(mid.data(), C_in, C_out, k, u, out + packed_convt_plane_halfs(C_in, C_out, k, u));
}

struct Tile { int WT, WC, MT, NT, CIC, MINB, T_BLK, CO_BLK; };
inline Tile pick_tile(int C) {
    Tile t;
    if (C <= 32)      { t.WT = 4; t.WC = 1; t.MT = 2; t.NT = 1; t.CIC = 32; }
    else if (C <= 64) { t.WT = 2; t.WC = 2; t.MT = 2; t.NT = 1; t.CIC = 64; }
    else              { t.WT = 2; t.WC = 2; t.MT = 2; t.NT = 2; t.CIC = 64; }
    t.MINB = 2;
    t.T_BLK = t.WT * t.MT * 32;
    t.CO_BLK = t.WC * t.NT * 32;
    return t;
}

inline bool applicable(const Launch& a, int nz) {
    if (nz < 1 || (!a.z_is_phase && nz > kMaxGroup) || a.C < 32 || (a.C & 31)) return false;
    const Tile t = pick_tile(a.C);
    const int Cin = a.C_in ? a.C_in : a.C, Lin = a.L_in ? a.L_in : a.L;
    if (Cin % t.CIC || a.C % t.CO_BLK) return false;
    if ((double)a.L * a.C * 4.0 >= 2147483648.0 - 4194304.0 || (double)Lin * Cin * 4.0 >= 2147483648.0 - 4194304.0)
        return false;                                                            // 32-bit buffer offsets
    return true;
}

inline hipError_t launch(Launch& a, int nz, hipStream_t stream) {
    Tile t = pick_tile(a.C);
    a.n_co_blk = a.C / t.CO_BLK;
    const int Cin = a.C_in ? a.C_in : a.C;
    const int n_rows = a.n_idx ? a.n_idx : a.L;
    a.Qp = b16::packed_qsteps(Cin);
    a.n_ct = b16::packed_cotiles(a.C);
    a.nz = nz;
    int span = 0;
    for (int j = 0; j < (a.z_is_phase ? 1 : nz); ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
        if (!a.z_is_phase) a.plane_bytes[j] = (unsigned)(packed_plane_halfs(a.C, a.C, a.p[j].ks) * 2);   // (phases: set by the caller)
    }
    const bool zs = a.sum_y != nullptr;
    a.sum_div = (float)nz;
    const int ablate_env = IRIS_DIAG_ENV("IRIS_S3_ABLATE", 0);
    a.ablate = ablate_env;
    // small grids (short utterances at batch 1): a launch lasts as long as its longest block, so the tile height is
    // halved when the grid has fewer than 2.5 blocks per CU (measured: 1.05 -> 0.88 ms at T = 100)
    static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev);
                                 (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int mt_env = IRIS_DIAG_ENV("IRIS_S3_MT", 0);
    const long long blocks2 = (long long)((n_rows + t.T_BLK - 1) / t.T_BLK) * (zs ? 1 : nz) * a.n_co_blk * a.B;
    if (mt_env ? mt_env == 1 : 2 * blocks2 < 5LL * n_cu) { t.MT = 1; t.T_BLK = t.WT * 32; }
    const int SB = t.CIC * 2 + 16;
    const size_t window = 2 * (size_t)(t.T_BLK + span) * SB;
    const size_t scratch = (size_t)4 * 32 * (t.NT * 32 * 4 + 16);
    const size_t lds_bytes = window > scratch ? window : scratch;
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const int n_t = (n_rows + t.T_BLK - 1) / t.T_BLK;
    dim3 grid((unsigned)(n_t * (zs ? 1 : nz) * a.n_co_blk), (unsigned)a.B, 1u), block(256);
#define IRIS_S3_LAUNCH(...)                                                                       \
    do {                                                                                          \
        auto kfn = __VA_ARGS__;                                                                   \
        { const hipError_t e__ = ::iris::launch_kernel_named("conv_mfma_f32s_kernel", kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
        return hipSuccess;                                                                        \
    } while (0)
#define IRIS_S3_CASE(WT_, WC_, NT_, CIC_)                                                         \
    if (t.WT == WT_ && t.WC == WC_ && t.NT == NT_) {                                              \
        if (zs) { if (t.MT == 1) IRIS_S3_LAUNCH(conv_mfma_f32s_kernel<WT_, WC_, 1, NT_, CIC_, 2, true>);    \
                  else           IRIS_S3_LAUNCH(conv_mfma_f32s_kernel<WT_, WC_, 2, NT_, CIC_, 2, true>); }  \
        else    { if (t.MT == 1) IRIS_S3_LAUNCH(conv_mfma_f32s_kernel<WT_, WC_, 1, NT_, CIC_, 2, false>);   \
                  else           IRIS_S3_LAUNCH(conv_mfma_f32s_kernel<WT_, WC_, 2, NT_, CIC_, 2, false>); } \
    }
    IRIS_S3_CASE(4, 1, 1, 32)
    IRIS_S3_CASE(2, 2, 1, 64)
    IRIS_S3_CASE(2, 2, 2, 64)
#undef IRIS_S3_CASE
#undef IRIS_S3_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace s3
}  // namespace iris
