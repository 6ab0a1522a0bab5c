// convt_mfma_bf16.h -- LeakyReLU + ConvTranspose1d of the upsampling ladder as ONE dense GEMM per launch, bf16 storage
// (round 4; the bf16 sibling of convt_mfma_f32.h).
//
// Reference layer: ups[i] = LeakyReLU(0.1) -> ConvTranspose1d(C_in -> C_in / 2, k, stride u, padding (k - u) / 2),
// src/iris/hifigan_pretrained.py:97-109,127-128 (Keras twin src/iris/vocoder.py:85-91,114-115).  The reference has no
// bf16 path: the rounding points are those of conv_mfma_bf16.h (tests/test_gpu_bf16.py's restatement, parity unpinned).
//
// Formulation (see convt_mfma_f32.h): with two taps per phase (k = 2u, every V1 upsampler) the u phase convolutions read
// the same rows of x and the u outputs of row index i are contiguous in the channels-last output, so the layer is the GEMM
//       [L_in + 1, 2 C_in]  x  [2 C_in, u C_out]
// whose output matrix is the output tensor shifted by (k - u) / 2 rows.  The polyphase launches of conv_mfma_bf16.h give
// every (tile, phase, C_out block) its own short-lived block: one window load, 8-64 MFMAs per wave, one epilogue -- at the
// bf16 matrix rate a block's MFMAs last 0.4-1 us, less than the round trip in front of them (configs[2]: 500 TFLOP/s on
// the second upsampler, 2.4 TB/s on the HBM-bound last one).  Here a 256-thread block owns (WR MT 32) rows x (WC NT 32)
// columns of the GEMM, is persistent over tiles, requests the NEXT K chunk's (or tile's) window into registers while the
// current chunk's MFMAs run and writes it to LDS behind them; weight fragments run DB groups ahead in a register ring
// that reaches across chunk and tile boundaries.
// Tile order: the column blocks of a row tile read the same window; they are consecutive jobs of ONE XCD (blockIdx.x % 8),
// so the window is fetched from HBM once and found in that XCD's L2 by the others.
// Every output element is the same fp32 chain in the same order as in the polyphase kernel -- accumulator started at the
// bias, chunks of the polyphase kernel's CIC channels, tap-major inside a chunk, 16 channels per MFMA, rounded to bf16
// once -- so the launch plan never changes a sample (tools/bitwise_sweep.py against -DIRIS_CONVT_GEMM_B16_DEFAULT=0).
#pragma once
#include "conv_mfma_bf16.h"

namespace iris {
namespace b16 {

#ifndef IRIS_CONVT_GEMM_B16_DEFAULT
#define IRIS_CONVT_GEMM_B16_DEFAULT 1     // (A/B builds: 0 = the polyphase launches of conv_mfma_bf16.h)
#endif

#ifndef IRIS_CONVT_B16_WIDE_A
#define IRIS_CONVT_B16_WIDE_A 1           // (A/B builds: 0 = never the 64 x 256 block shape)
#endif

struct ConvtLaunch {
    const uint16_t* x[3];    // NIN = 1: x[0]; NIN = 3: the previous stage's three branch outputs (MRF mean formed while staging)
    const void* wp;          // u phase blobs of pack_convt_bf16
    const float* bias;       // [C_out]
    uint16_t* y;             // [B, L_out, C_out] bf16
    int B, L_in, L_out, C_in, C_out;
    int u, out_off;          // output row of (row index i, phase ph) = i * u + out_off + ph;  out_off = -(k - u) / 2
    int n_idx;               // GEMM rows per batch item: L_in + 1
    int Qp, n_ct;            // packed_qsteps(C_in), packed_cotiles(C_out): the layout of one phase blob
    unsigned phase_bytes;    // bytes between consecutive phase blobs
    int n_row_tiles;         // row tiles per batch item
    int n_items;             // row tiles over the whole batch
    int n_col_blk;           // column blocks of a row tile
    int xcd_order;           // jobs dealt to the XCDs by row item (else one list over the whole grid)
    int jobs_per_xcd;        // xcd_order: ceil(n_items / 8) * n_col_blk; else n_items * n_col_blk
    int in_act;              // IN_ACT_NONE / IN_ACT_LRELU (NIN = 1), IN_ACT_MRF_LRELU (NIN = 3)
    float slope, inv_n;
};

template <int MT, int NT, int WR, int WC, int CIC, int NIN, int MINB>
__global__ void __launch_bounds__(256, MINB) convt_mfma_bf16_kernel(const ConvtLaunch a) {
    static_assert(NIN == 1 || NIN == 3, "one input tensor, or the three branch outputs of the previous stage");
    static_assert(WR * WC == 4, "four waves per block");
    extern __shared__ __attribute__((aligned(16))) char lds_b16[];
    char* const lds = lds_b16;
    constexpr int SB = CIC * 2 + 16;                         // window row stride (bytes) = 16 * odd
    constexpr int PPR = CIC / 8;                             // 16-byte pieces (8 channels) per window row
    constexpr int QPC = CIC / 16;                            // 16-channel MFMA steps per tap and chunk
    constexpr int R_BLK = WR * MT * 32;
    constexpr int WIN = R_BLK + 1;                           // window rows of a tile (two taps)
    constexpr int NQ = (WIN * PPR + 255) / 256;              // staged pieces per thread, tensor and chunk
    constexpr int RPI = 256 / PPR;                           // rows advanced per staged piece
    constexpr int NG = 2 * QPC;                              // MFMA groups (16 input channels of one tap) per chunk
    constexpr int LPG = (NQ + NG - 1) / NG;                  // staging requests per group
    constexpr int DB = 4;                                    // weight fragments requested DB groups ahead
    static_assert(DB <= QPC && NG > DB, "the ring reaches into tap 0 of the next phase only");
    constexpr int RS = NT * 32 * 4 + 16;                     // epilogue scratch row stride (bytes) = 16 * odd
    constexpr int PPRO = NT * 4;                             // 16-byte bf16 pieces per row of a wave's column span
    constexpr int NP = 2 * NT;                               // pieces per lane and m-tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave - wr * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int C = a.C_in;
    const int n_chunks = C / CIC;
    const float slope = a.in_act == IN_ACT_NONE ? 1.f : a.slope;
    const unsigned in_bytes = (unsigned)a.L_in * (unsigned)C * 2u;
    const unsigned out_bytes = (unsigned)a.L_out * (unsigned)a.C_out * 2u;
    const unsigned q_bytes = (unsigned)a.n_ct * 1024u;       // bytes per (tap, 16-channel step) of one phase blob
    const unsigned tap_bytes = (unsigned)a.Qp * q_bytes;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp, (unsigned)a.u * a.phase_bytes);
    char* const scr = lds + WIN * SB + wave * (32 * RS);     // this wave's epilogue scratch
    float* const lds_bias = reinterpret_cast<float*>(lds + WIN * SB + 4 * (32 * RS));

    for (int i = tid; i < (a.C_out >> 2); i += 256)          // bias table (visible after the first barrier below)
        *reinterpret_cast<f32x4*>(lds_bias + 4 * i) = *reinterpret_cast<const f32x4*>(a.bias + 4 * i);

    // ---- staging: piece i of this thread = row r_lane + i * RPI of the window, channels [c0 + 8 p_lane, +8) ----
    const int r_lane = tid / PPR, p_lane = tid - r_lane * PPR;
    const unsigned row_stride = (unsigned)(RPI * C) * 2u;
    char* const lds_wr = lds + r_lane * SB + p_lane * 16;
    u32x4 st[NIN][NQ];
    auto stage_vbase = [&](int in_row0, int c0) -> unsigned {   // rows < 0 wrap to >= 2^31, rows >= L_in exceed num_records: both read 0
        return (unsigned)((in_row0 + r_lane) * C + c0 + 8 * p_lane) * 2u;
    };
    auto stage_load_one = [&](int i, size_t x_off, unsigned bytes, unsigned vbase) {
        const unsigned voff = r_lane + i * RPI < WIN ? vbase + (unsigned)i * row_stride : kOob;
#pragma unroll
        for (int t = 0; t < NIN; ++t) st[t][i] = buf_load4(make_rsrc(a.x[t] + x_off, bytes), voff, 0);
    };
    auto stage_write_all = [&]() {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (r_lane + i * RPI < WIN) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (NIN == 3) {     // ((x0 + x1) + x2) + 0, times 1/n: conv_mfma_bf16.h's stage_window
                        float l = ((bf_lo(st[0][i][e]) + bf_lo(st[1][i][e])) + bf_lo(st[2][i][e])) + 0.f;
                        float h = ((bf_hi(st[0][i][e]) + bf_hi(st[1][i][e])) + bf_hi(st[2][i][e])) + 0.f;
                        l *= a.inv_n; h *= a.inv_n;
                        o[e] = pack_bf2(lrelu1(l, slope), lrelu1(h, slope));
                    } else {
                        o[e] = pack_bf2(lrelu1(bf_lo(st[0][i][e]), slope), lrelu1(bf_hi(st[0][i][e]), slope));
                    }
                }
                *reinterpret_cast<u32x4*>(lds_wr + i * RPI * SB) = o;
            }
    };

    // ---- jobs: XCD x (= blockIdx.x % 8) owns the row items x, x + 8, ...; its blocks walk (item, column block) with the
    //      column block fastest.  job -> tile ----
    struct Tile { size_t x_off, y_off; int i0, gct0; unsigned wvoff[NT]; };
    // (xcd_order = 0: one job list for the whole grid -- a single column block, or so few row items that dealing them to
    //  XCDs would leave some of the eight with a job more than the others)
    const int xmul = a.xcd_order ? 8 : 1;
    const int xcd = a.xcd_order ? (int)(blockIdx.x & 7) : 0;
    const int slot = a.xcd_order ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int slots = a.xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    auto job_valid = [&](int job) { return job < a.jobs_per_xcd && (job / a.n_col_blk) * xmul + xcd < a.n_items; };
    auto make_tile = [&](int job) {
        Tile t;
        const int il = job / a.n_col_blk, cb = job - il * a.n_col_blk;
        const int item = il * xmul + xcd;
        const int b = item / a.n_row_tiles, rt = item - b * a.n_row_tiles;
        t.x_off = (size_t)b * a.L_in * C;
        t.y_off = (size_t)b * a.L_out * a.C_out;
        t.i0 = rt * R_BLK;
        t.gct0 = (cb * WC + wc) * NT;                        // this wave's first 32-wide column tile of the u * C_out columns
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int gct = t.gct0 + nt;
            const int ph = gct / a.n_ct, ct = gct - ph * a.n_ct;
            t.wvoff[nt] = (unsigned)ph * a.phase_bytes + (unsigned)(ct * 64 + lane) * 16u;
        }
        return t;
    };

    const char* a_lane = lds + (wr * MT * 32 + lo) * SB + hi * 16;
    f32x16 acc[MT][NT];
    u32x4 bw[DB + 1][NT];

    int job = slot;
    if (!job_valid(job)) return;
    Tile t = make_tile(job);
    {   // prologue: the first window and the first weight fragments
        const unsigned vb0 = stage_vbase(t.i0 - 1, 0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) stage_load_one(i, t.x_off, in_bytes, vb0);
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bw[d][nt] = buf_load4(wrs, t.wvoff[nt], (unsigned)d * q_bytes);
        stage_write_all();
        __syncthreads();
    }
    for (;;) {
        const int job_next = job + slots;
        const bool more = job_valid(job_next);
        const Tile tn = make_tile(more ? job_next : job);
        // the accumulators start at the bias: (nt, r) -> channel ct * 32 + 8 (r >> 2) + 4 hi + (r & 3)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int gct = t.gct0 + nt;
            const int ct = gct % a.n_ct;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + ct * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][nt][4 * g + e] = b4[e];
            }
        }

        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool last = chunk + 1 == n_chunks;
            const bool has_next = !last || more;
            // the phase that follows: the next chunk of this tile, or chunk 0 of the block's next tile
            const Tile& tq = last ? tn : t;
            const int cq = last ? 0 : chunk + 1;
            const unsigned in_bytes_n = has_next ? in_bytes : 0u;       // (nothing follows: zero-length descriptors, the loads return 0)
            const unsigned vbn = stage_vbase(tq.i0 - 1, cq * CIC);
            const unsigned wsoff0 = (unsigned)(chunk * QPC) * q_bytes;
            const unsigned wsoffn = (unsigned)(cq * QPC) * q_bytes;
            auto a_ptr = [&](int n) { return a_lane + (n / QPC) * SB + (n % QPC) * 32; };
            auto b_load = [&](int n, int nt) {               // group n of this chunk, or group n - NG of the next phase
                if (n < NG)
                    return buf_load4(wrs, t.wvoff[nt], wsoff0 + (unsigned)(n / QPC) * tap_bytes + (unsigned)(n % QPC) * q_bytes);
                return buf_load4(wrs, has_next ? tq.wvoff[nt] : kOob, wsoffn + (unsigned)(n - NG) * q_bytes);   // n - NG < DB <= QPC: tap 0
            };
            u32x4 av[2][MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const u32x4*>(a_ptr(0) + m * 32 * SB);
#pragma unroll
            for (int n = 0; n < NG; ++n) {
#pragma unroll
                for (int l = 0; l < LPG; ++l)
                    if (n * LPG + l < NQ) stage_load_one(n * LPG + l, tq.x_off, in_bytes_n, vbn);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bw[(n + DB) % (DB + 1)][nt] = b_load(n + DB, nt);
                if (n + 1 < NG) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        av[(n + 1) & 1][m] = *reinterpret_cast<const u32x4*>(a_ptr(n + 1) + m * 32 * SB);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, bw[n % (DB + 1)][nt]), __builtin_bit_cast(bf16x8, av[n & 1][m]),
                            acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // the next phase expects its groups 0 .. DB-1 in ring slots 0 .. DB-1: they were loaded into (NG + d) % (DB + 1)
                u32x4 tmp[DB][NT];
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) tmp[d][nt] = bw[(NG + d) % (DB + 1)][nt];
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bw[d][nt] = tmp[d][nt];
            }
            if (last) {
                // Epilogue (conv_mfma_bf16.h's): each wave turns its 32-row m-tiles through a private LDS scratch (fp32,
                // [row][channel]) and reads them back as 16-byte bf16 pieces that are contiguous in the output row --
                // here across the phases too: the NT 32-wide column tiles of a wave are 64 NT contiguous bytes of row i.
                // (acc) + 0 is rounded to bf16 once (the polyphase kernel's residual-free (acc + bias) + 0).
                const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y + t.y_off, out_bytes);
                int prow[NP], pph[NP], pco[NP], pscr[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int q = j * 64 + lane;
                    const int row_l = q / PPRO, pc = q - row_l * PPRO;
                    const int gct = t.gct0 + (pc >> 2);
                    const int ph = gct / a.n_ct, ct = gct - ph * a.n_ct;
                    prow[j] = row_l; pph[j] = ph; pco[j] = ct * 32 + 8 * (pc & 3);
                    pscr[j] = row_l * RS + pc * 32;
                }
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
                    unsigned pvoff[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(scr + pscr[j]);
                        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(scr + pscr[j] + 16);
                        outp[j][0] = pack_bf2(lo4[0] + 0.f, lo4[1] + 0.f);
                        outp[j][1] = pack_bf2(lo4[2] + 0.f, lo4[3] + 0.f);
                        outp[j][2] = pack_bf2(hi4[0] + 0.f, hi4[1] + 0.f);
                        outp[j][3] = pack_bf2(hi4[2] + 0.f, hi4[3] + 0.f);
                        const int i = t.i0 + (wr * MT + m) * 32 + prow[j];      // GEMM row of this piece
                        const int o = i * a.u + a.out_off + pph[j];
                        const bool ok = i < a.n_idx && o >= 0 && o < a.L_out;
                        pvoff[j] = ok ? (unsigned)(o * a.C_out + pco[j]) * 2u : kOob;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(outp[j], yr, (int)pvoff[j], 0, 0);
                    asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 stores (see mrf_conv_mfma_f32.h)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NP; ++j) asm volatile("" :: "v"(outp[j]));      // store data stays live until the group has issued
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();          // the scratch is rewritten by the next m-tile
                }
            }
            if (has_next) {
                __syncthreads();          // every wave is done reading this chunk's window
                stage_write_all();
                __syncthreads();
            }
        }
        if (!more) break;
        job = job_next;
        t = tn;
    }
}

#ifndef IRIS_KERNELS_ONLY
struct ConvtTile { int MT, NT, WR, WC, CIC, MINB; };

// True when LeakyReLU + ConvTranspose1d (k, u) with n_in input tensors can take the GEMM kernel.
inline bool convt_gemm_b16_applicable(int C_in, int C_out, int k, int u, int L_in, int L_out, int B, int n_in, float slope) {
    if (!IRIS_DIAG_ENV("IRIS_B16_CONVT_GEMM", IRIS_CONVT_GEMM_B16_DEFAULT)) return false;
    if (u < 1 || k != 2 * u) return false;                                  // two taps per phase (every V1 upsampler)
    if ((C_in & 63) || (C_out & 31)) return false;
    if (n_in != 1 && n_in != 3) return false;
    if (((u * (C_out / 32)) & 1)) return false;                             // at least the 128 x 64 block shape
    if ((uint64_t)L_in * C_in * 2u >= 0x7fffffffull || (uint64_t)L_out * C_out * 2u >= 0x7fffffffull) return false;   // 32-bit buffer offsets
    if ((uint64_t)u * packed_convt_phase_halfs(C_in, C_out, k, u) * 2u >= 0x7fffffffull) return false;
    if ((uint64_t)(L_in + 1 + 255) * (uint64_t)B >= 0x3fffffffull) return false;
    (void)slope;
    return true;
}

inline hipError_t launch_convt_gemm_b16(ConvtLaunch& a, int k, hipStream_t stream) {
    a.n_idx = a.L_in + 1;
    a.out_off = -(k - a.u) / 2;
    a.Qp = packed_qsteps(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    a.phase_bytes = (unsigned)(packed_convt_phase_halfs(a.C_in, a.C_out, k, a.u) * 2);
    const int n_cols32 = a.u * a.n_ct;
    const int n_cu = device_cu_count();
    // K chunk = the polyphase kernel's (pick_tile): 128 channels for the wide layers, else 64 -- the same fp32 chains
    const int CIC = (a.C_out > 64 && a.C_in % 128 == 0) ? 128 : 64;
    // Shape: 64 x 256 (each weight fragment feeds two row tiles, each activation fragment two column tiles) where the
    // output is wide and that still gives every CU two blocks; 64 x 128 otherwise; 128 x 64 for the last upsampler.
    ConvtTile t;
    auto blocks_of = [&](int rows, int cols32) { return (long long)((a.n_idx + rows - 1) / rows) * (n_cols32 / cols32) * a.B; };
    if (CIC == 128) {
        if (IRIS_CONVT_B16_WIDE_A && (n_cols32 & 7) == 0 && blocks_of(64, 8) >= 2LL * n_cu) t = ConvtTile{2, 2, 1, 4, 128, 2};
        else if ((n_cols32 & 3) == 0) t = ConvtTile{2, 1, 1, 4, 128, 3};
        else return hipErrorInvalidValue;
    } else {
        if ((n_cols32 & 3) == 0) t = ConvtTile{2, 1, 1, 4, 64, 3};
        else t = ConvtTile{2, 1, 2, 2, 64, 3};
    }
    const int rows = t.WR * t.MT * 32;
    a.n_row_tiles = (a.n_idx + rows - 1) / rows;
    a.n_col_blk = n_cols32 / (t.WC * t.NT);
    const long long n_items = (long long)a.n_row_tiles * a.B;
    if (n_items * a.n_col_blk > 0x3fffffffLL) return hipErrorInvalidValue;
    a.n_items = (int)n_items;
    // XCD order where column blocks share a window and the eight XCDs get the same number of row items (+-12.5 %)
    const long long items8 = ((n_items + 7) / 8) * 8;
    a.xcd_order = a.n_col_blk > 1 && (items8 - n_items) * 8 <= n_items;
    const int xm = a.xcd_order ? 8 : 1;
    a.jobs_per_xcd = a.xcd_order ? (int)(items8 / 8) * a.n_col_blk : (int)(n_items * a.n_col_blk);
    a.inv_n = a.in_act == IN_ACT_MRF_LRELU ? 1.0f / 3.0f : 1.0f;
    const bool three = a.in_act == IN_ACT_MRF_LRELU;
    // persistent grid: at most per_cu blocks per CU (per XCD: per_cu * CUs / 8), evened out so that every block walks the
    // same number of jobs (+-1)
    const size_t lds_bytes = (size_t)(rows + 1) * (CIC * 2 + 16) + (size_t)4 * 32 * (t.NT * 128 + 16) + (size_t)a.C_out * 4;
    long long per_cu = (!three && t.NT == 1) ? 4 : t.MINB;      // (the one-input narrow forms compile to <= 128 VGPRs)
    { const long long by_lds = (long long)(160 * 1024) / (long long)lds_bytes; if (by_lds < per_cu) per_cu = by_lds; }
    if (per_cu < 1) return hipErrorInvalidValue;
    const long long cap = per_cu * (n_cu / xm);
    long long g = a.jobs_per_xcd < cap ? a.jobs_per_xcd : cap;
    if (g < 1) g = 1;
    const long long rounds = (a.jobs_per_xcd + g - 1) / g;
    g = (a.jobs_per_xcd + rounds - 1) / rounds;
    g *= xm;
    dim3 grid((unsigned)g, 1u, 1u), block(256);
#define IRIS_CONVT_B16_CASE(MT_, NT_, WR_, WC_, CIC_, MINB_)                                                                   \
    if (t.MT == MT_ && t.NT == NT_ && t.WR == WR_ && t.CIC == CIC_) {                                                         \
        if (three) return ::iris::launch_kernel_named("convt_mfma_bf16_kernel<" #MT_ ", " #NT_ ", " #WR_ ", " #WC_ ", " #CIC_ ", 3>", \
                              convt_mfma_bf16_kernel<MT_, NT_, WR_, WC_, CIC_, 3, MINB_>, grid, block, lds_bytes, stream, a);  \
        return ::iris::launch_kernel_named("convt_mfma_bf16_kernel<" #MT_ ", " #NT_ ", " #WR_ ", " #WC_ ", " #CIC_ ", 1>",      \
                              convt_mfma_bf16_kernel<MT_, NT_, WR_, WC_, CIC_, 1, MINB_>, grid, block, lds_bytes, stream, a);  \
    }
    IRIS_CONVT_B16_CASE(2, 2, 1, 4, 128, 2)
    IRIS_CONVT_B16_CASE(2, 1, 1, 4, 128, 3)
    IRIS_CONVT_B16_CASE(2, 1, 1, 4, 64, 3)
    IRIS_CONVT_B16_CASE(2, 1, 2, 2, 64, 3)
#undef IRIS_CONVT_B16_CASE
    return hipErrorInvalidValue;
}

// LeakyReLU + ConvTranspose1d described as a polyphase Launch (z_is_phase, nz = u): the GEMM kernel where it applies,
// else the polyphase launches.
inline hipError_t launch_convt_bf16(Launch& a, int k, int u, hipStream_t stream) {
    const int n_in = a.in_act == IN_ACT_MRF_LRELU ? a.n_mrf : 1;
    if (!a.x_f32_cf && convt_gemm_b16_applicable(a.C_in, a.C_out, k, u, a.L_in, a.L_out, a.B, n_in, a.slope)) {
        ConvtLaunch g; memset(&g, 0, sizeof(g));
        if (a.in_act == IN_ACT_MRF_LRELU) { g.x[0] = a.xmrf[0]; g.x[1] = a.xmrf[1]; g.x[2] = a.xmrf[2]; }
        else { g.x[0] = (const uint16_t*)a.p[0].x; g.x[1] = g.x[0]; g.x[2] = g.x[0]; }
        g.wp = a.p[0].wp; g.bias = a.p[0].bias; g.y = a.p[0].y;
        g.B = a.B; g.L_in = a.L_in; g.L_out = a.L_out; g.C_in = a.C_in; g.C_out = a.C_out;
        g.u = u; g.in_act = a.in_act; g.slope = a.slope;
        const hipError_t e = launch_convt_gemm_b16(g, k, stream);
        if (e != hipErrorInvalidValue) return e;
    }
    return launch_conv_bf16(a, u, stream);
}
#endif  // IRIS_KERNELS_ONLY

}  // namespace b16
}  // namespace iris
