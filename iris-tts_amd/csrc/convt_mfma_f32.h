// convt_mfma_f32.h -- LeakyReLU + ConvTranspose1d of the upsampling ladder as ONE dense GEMM per launch (round 4).
//
// Reference layer: ups[i] = LeakyReLU(0.1) -> ConvTranspose1d(C_in -> C_in / 2, k, stride u, padding (k - u) / 2),
// src/iris/hifigan_pretrained.py:97-109,127-128 (Keras twin src/iris/vocoder.py:85-91,114-115).
//
// Formulation.  With taps = k / u the polyphase form (conv_mfma_f32.h) is  y[i u + ph - p, co] = b[co] +
// sum_{kap, ci} x[i - (taps - 1) + kap, ci] W_ph[kap][ci][co]  for every phase ph = 0 .. u - 1.  All u phases read the same
// rows of x, and the u outputs of row index i are CONTIGUOUS in the channels-last output (rows i u - p .. i u - p + u - 1
// of [L_out, C_out]): the layer is the GEMM
//       [L_in + taps - 1, taps C_in]  x  [taps C_in, u C_out]
// whose output matrix is the output tensor itself, shifted by p rows.  Round 1-3 ran it as u separate convolutions in one
// launch (blockIdx.z = phase): every phase block re-staged the same window, a block lived for C_in / 64 chunks of 16
// MFMA groups with a load round trip and two barriers in front of each (48-85 TFLOP/s, 5 % of the headline step at 0.48
// of the matrix peak).  Here a block owns  (WR MT 32) rows x (WC NT 32) columns of that GEMM, walks the K chunks with the
// NEXT chunk's window requested during the current chunk's MFMAs (register staging: issue early, write late -- the
// schedule of mrf_conv_mfma_f32.h) and is persistent over tiles, column blocks fastest so that blocks that share a window
// run together and find it in L2.
// Every output element is the same fmaf chain in the same order as in the polyphase kernel (chunk-major, tap, 8-channel
// group in the order 0,4,1,5,2,6,3,7, bias added last): bit-identical, so the launch plan never changes a sample.
#pragma once
#include "mrf_conv_mfma_f32.h"

namespace iris {

#ifndef IRIS_CONVT_GEMM_DEFAULT
#define IRIS_CONVT_GEMM_DEFAULT 1        // (A/B builds: 0 = the polyphase launches of conv_mfma_f32.h)
#endif

struct ConvtLaunch {
    const float* x;          // [B, L_in, C_in]  (LeakyReLU is applied while the window is staged)
    const float* x1;         // NIN = 3 (the previous stage left its three branch outputs): the input is
    const float* x2;         //   LeakyReLU(((x + x1) + x2) / 3) -- the MRF mean of hifigan_pretrained.py:131-137, formed while staging
    const f32x4* wp;         // u phase blobs of pack_convt_weights
    const float* bias;       // [C_out]
    float* y;                // [B, L_out, C_out]
    int B, L_in, L_out, C_in, C_out;
    int u, out_off;          // output row of (row index i, phase ph) = i * u + out_off + ph;  out_off = -(k - u) / 2
    int n_idx;               // GEMM rows per batch item: L_in + taps - 1
    int Gp, n_ct;            // packed_groups(C_in), packed_cotiles(C_out): the layout of one phase blob
    unsigned phase_bytes;    // bytes between consecutive phase blobs
    int n_row_tiles, n_col_blk, n_tiles;   // per batch item / per batch item / over the whole batch
    int n_items;             // row tiles over the whole batch
    int xcd_order;           // jobs dealt to the XCDs by row item (the column blocks of a row tile share a window: one XCD's L2
                             // then fetches it once); else one job list over the whole grid
    int jobs_per_xcd;        // xcd_order: ceil(n_items / 8) * n_col_blk; else n_tiles
    float slope;
};

template <int MT, int NT, int WR, int WC, int TAPS, int NIN>
__global__ void __launch_bounds__(256, 2) convt_mfma_f32_kernel(const ConvtLaunch a) {
    static_assert(NIN == 1 || NIN == 3, "one input tensor, or the three branch outputs of the previous stage");
    static_assert(WR * WC == 4, "four waves per block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CIC = 64, S = CIC + 4, QPR = CIC / 4, GPC = CIC / 8;
    constexpr int R_BLK = WR * MT * 32;
    constexpr int WIN = R_BLK + TAPS - 1;                    // window rows of a tile
    constexpr int NQ = (WIN * QPR + 255) / 256;              // staged 16-byte quads per thread and chunk
    constexpr int RPI = 256 / QPR;                           // rows advanced per staged quad
    constexpr int NG = TAPS * GPC;                           // MFMA groups (8 input channels of one tap) per chunk
    // weight fragments are requested DB groups ahead: ~2,000 cycles of MFMAs (a group is 4 MT NT MFMAs of 64 cycles), which
    // covers an L2 round trip under load -- two groups ahead left the 32 x 128 block parked 41 % of its cycles (PMC)
    constexpr int DB = MT * NT >= 4 ? 2 : (MT * NT == 2 ? 4 : 8);
    static_assert(NG >= NQ && NG > DB && DB <= GPC, "a chunk must be long enough to request the next window; the ring reaches into tap 0 only");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave - wr * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int C = a.C_in;
    const int n_chunks = C / CIC;                            // C_in % 64 == 0 (convt_gemm_applicable)
    const float slope = a.slope;
    const unsigned in_bytes = (unsigned)a.L_in * (unsigned)C * 4u;
    const unsigned out_bytes = (unsigned)a.L_out * (unsigned)a.C_out * 4u;
    const unsigned wbytes_group = (unsigned)a.n_ct * 1024u;  // bytes per (tap, group) of one phase blob
    const unsigned tap_bytes = (unsigned)a.Gp * wbytes_group;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp, (unsigned)a.u * a.phase_bytes);
    float* const lds_bias = lds + WIN * S;

    // bias table (read in the epilogues; visible after the first barrier below)
    for (int i = tid; i < (a.C_out >> 2); i += 256)
        *reinterpret_cast<f32x4*>(lds_bias + 4 * i) = *reinterpret_cast<const f32x4*>(a.bias + 4 * i);

    // ---- staging: quad i of this thread = row r_lane + i * RPI of the window, channels [c0 + 4 q_lane, +4) ----
    const int r_lane = tid / QPR, q_lane = tid - r_lane * QPR;
    const unsigned row_stride = (unsigned)(RPI * C) * 4u;
    float* const lds_wr = lds + r_lane * S + 4 * q_lane;
    f32x4 st[NIN][NQ];
    auto stage_vbase = [&](int in_row0, int c0) -> unsigned {   // rows < 0 wrap to >= 2^31, rows >= L_in exceed num_records: both read 0
        return (unsigned)((in_row0 + r_lane) * C + c0 + 4 * q_lane) * 4u;
    };
    // (the NIN tensors have one shape: one offset serves all; x_off selects the batch item)
    auto stage_load_one = [&](int i, size_t x_off, unsigned bytes, unsigned vbase) {
        const unsigned voff = r_lane + i * RPI < WIN ? vbase + (unsigned)i * row_stride : kOobOffset;
        st[0][i] = buf_load4(make_rsrc(a.x + x_off, bytes), voff, 0);
        if constexpr (NIN == 3) {
            st[1][i] = buf_load4(make_rsrc(a.x1 + x_off, bytes), voff, 0);
            st[2][i] = buf_load4(make_rsrc(a.x2 + x_off, bytes), voff, 0);
        }
    };
    auto stage_write_all = [&]() {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (r_lane + i * RPI < WIN) {
                f32x4 v = st[0][i];           // LeakyReLU(x) = max(x, slope x) for 0 <= slope <= 1 (checked by the host)
                if constexpr (NIN == 3) {     // xs = rb0; xs += rb1; xs += rb2; x = xs / 3 (true division), as the polyphase kernel stages it
                    v = v + st[1][i];
                    v = v + st[2][i];
                    v = v / 3.0f;
                }
                v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
                v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
                *reinterpret_cast<f32x4*>(lds_wr + i * RPI * S) = v;
            }
    };

    // ---- a tile = (batch item, row tile, column block); column blocks fastest ----
    struct Tile { size_t x_off, y_off; int i0; unsigned wvoff[NT]; int ch[NT]; int ph[NT]; };
    const int xmul = a.xcd_order ? 8 : 1;
    const int xcd = a.xcd_order ? (int)(blockIdx.x & 7) : 0;
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
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int gct = (cb * WC + wc) * NT + nt;        // 32-wide column tile of the u * C_out columns
            const int ph = gct / a.n_ct, ct = gct - ph * a.n_ct;
            t.ph[nt] = ph;
            t.ch[nt] = ct * 32 + 4 * hi;
            t.wvoff[nt] = (unsigned)ph * a.phase_bytes + (unsigned)(ct * 64 + lane) * 16u;
        }
        return t;
    };

    const float* aptr = lds + (wr * MT * 32 + lo) * S + 4 * hi;
    f32x16 acc[MT][NT];
    f32x4 bw[DB + 1][NT];

    int tile = a.xcd_order ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;       // (a job index of this block's list)
    if (!job_valid(tile)) return;
    Tile t = make_tile(tile);
    {   // prologue: the first window and the first weight fragments
        const unsigned vb0 = stage_vbase(t.i0 - (TAPS - 1), 0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) stage_load_one(i, t.x_off, in_bytes, vb0);
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bw[d][nt] = buf_load4(wrs, t.wvoff[nt], (unsigned)d * wbytes_group);
        stage_write_all();
        __syncthreads();
    }
    for (;;) {
        const int tile_next = tile + slots;
        const bool more = job_valid(tile_next);
        const Tile tn = make_tile(more ? tile_next : tile);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;

        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool last = chunk + 1 == n_chunks;
            const bool has_next = !last || more;
            // the phase that follows: the next chunk of this tile, or chunk 0 of the block's next tile
            const Tile& tq = last ? tn : t;
            const int cq = last ? 0 : chunk + 1;
            const unsigned in_bytes_n = has_next ? in_bytes : 0u;       // (nothing follows: zero-length descriptors, the loads return 0)
            const unsigned vbn = stage_vbase(tq.i0 - (TAPS - 1), cq * CIC);
            const unsigned wsoff0 = (unsigned)(chunk * GPC) * wbytes_group;
            const unsigned wsoffn = (unsigned)(cq * GPC) * wbytes_group;
            auto a_ptr = [&](int n) { return aptr + (n / GPC) * S + 8 * (n % GPC); };
            auto b_load = [&](int n, int nt) {               // group n of this chunk, or group n - NG of the next phase
                if (n < NG)
                    return buf_load4(wrs, t.wvoff[nt], wsoff0 + (unsigned)(n / GPC) * tap_bytes + (unsigned)(n % GPC) * wbytes_group);
                return buf_load4(wrs, has_next ? tq.wvoff[nt] : kOobOffset, wsoffn + (unsigned)(n - NG) * wbytes_group);   // n - NG < DB <= GPC: tap 0
            };
            f32x4 av[2][MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(a_ptr(0) + m * 32 * S);
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                if (n < NQ) stage_load_one(n, tq.x_off, in_bytes_n, vbn);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bw[(n + DB) % (DB + 1)][nt] = b_load(n + DB, nt);
                if (n + 1 < NG) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        av[(n + 1) & 1][m] = *reinterpret_cast<const f32x4*>(a_ptr(n + 1) + m * 32 * S);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[n % (DB + 1)][nt][e], av[n & 1][m][e], acc[m][nt], 0, 0, 0);
                {   // one request slotted behind each of the first MFMAs (0x8 MFMA, 0x100 DS read, 0x20 VMEM read)
                    int ds_left = (n + 1 < NG) ? MT : 0, vm_left = NT + (n < NQ ? NIN : 0);
#pragma unroll
                    for (int k = 0; k < 4 * MT * NT; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                        else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // the next phase expects its groups 0 .. DB-1 in ring slots 0 .. DB-1: they were loaded into (NG + d) % (DB + 1)
                f32x4 tmp[DB][NT];
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) tmp[d][nt] = bw[(NG + d) % (DB + 1)][nt];
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bw[d][nt] = tmp[d][nt];
            }
            bool stored = false;
            if (last) {
                // Epilogue: D = W_frag x X_frag^T -- a lane holds ONE row (lane & 31) and, in registers 4g .. 4g+3, the four
                // consecutive channels 8g + 4 (lane >> 5) + {0..3} of its 32-column tile: one 16-byte piece of a channels-last
                // row.  Bias is added in place and the stores read the accumulators themselves, which nothing rewrites before
                // the next tile's zero-init behind the LDS write and the barriers below (see mrf_conv_mfma_f32.h on why store
                // data must not sit in short-lived registers).
                const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y + t.y_off, out_bytes);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + t.ch[nt] + 8 * g);
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[m][nt][4 * g + e] = acc[m][nt][4 * g + e] + b4[e];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int i = t.i0 + wr * MT * 32 + m * 32 + lo;           // GEMM row of this lane
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int o = i * a.u + a.out_off + t.ph[nt];
                        const bool ok = i < a.n_idx && o >= 0 && o < a.L_out;
                        const unsigned voff = ok ? (unsigned)(o * a.C_out + t.ch[nt]) * 4u : kOobOffset;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& src = acc[m][nt];
                            const f32x4 v = {src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                            buf_store4(v, yr, voff, (unsigned)(8 * g) * 4u);
                        }
                    }
                }
                asm volatile("s_nop 1");
                stored = true;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_next) {
                __syncthreads();          // every wave is done reading this chunk's window
                stage_write_all();
                if (stored) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[m][nt][r]));   // keep-alive of the store data
                }
                __syncthreads();
            }
        }
        if (!more) break;
        tile = tile_next;
        t = tn;
    }
}

#ifndef IRIS_KERNELS_ONLY
// The block shapes: rows x columns of the [L_in + taps - 1, u C_out] output matrix per 256-thread block.
struct ConvtTile { int MT, NT, WR, WC; };
inline int convt_blocks(const ConvtTile& t, int n_idx, int n_cols32, int B) {
    const int rows = t.WR * t.MT * 32, cols = t.WC * t.NT;
    return ((n_idx + rows - 1) / rows) * (n_cols32 / cols) * B;
}

// True when LeakyReLU + ConvTranspose1d `l` on a single input tensor can take the GEMM kernel.
inline bool convt_gemm_applicable(int C_in, int C_out, int k, int u, int L_in, int L_out, float slope) {
    if (!IRIS_DIAG_ENV("IRIS_HIFIGAN_CONVT_GEMM", IRIS_CONVT_GEMM_DEFAULT)) return false;
    if (u < 1 || k != 2 * u) return false;                                  // two taps per phase (every V1 upsampler)
    if ((C_in & 63) || (C_out & 31)) return false;
    if (((u * (C_out / 32)) & 1)) return false;                             // at least the 128 x 64 block shape
    if (!(slope >= 0.f && slope <= 1.f)) return false;
    if ((uint64_t)L_in * C_in * 4u >= 0x7fffffffull || (uint64_t)L_out * C_out * 4u >= 0x7fffffffull) return false;   // 32-bit buffer offsets
    if ((uint64_t)u * packed_convt_phase_floats(C_in, C_out, k, u) * 4u >= 0x7fffffffull) return false;
    return true;
}

#ifndef IRIS_CONVT_XCD_ORDER
#define IRIS_CONVT_XCD_ORDER 1           // (A/B builds: 0 = one job list over the whole grid, as before)
#endif
#ifndef IRIS_CONVT_PER_CU
#define IRIS_CONVT_PER_CU 4              // resident blocks per CU of the light shapes (70-118 VGPRs, 9-35 KB of LDS); 64 x 256 blocks: two
#endif
inline hipError_t launch_convt_gemm(ConvtLaunch& a, int k, hipStream_t stream) {
    const int taps = convt_taps(k, a.u);
    a.n_idx = a.L_in + taps - 1;
    a.out_off = -(k - a.u) / 2;
    a.Gp = packed_groups(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    a.phase_bytes = (unsigned)(packed_convt_phase_floats(a.C_in, a.C_out, k, a.u) * sizeof(float));
    const int n_cols32 = a.u * a.n_ct;
    const int n_cu = device_cu_count();
    // Shape: the largest block that still gives every CU work -- 64 x 256 (each weight fragment feeds two row tiles, each
    // activation fragment two column tiles) from 1.5 blocks per CU on, 64 x 128 from two per CU on, else 32 x 128; narrow
    // outputs (u C_out = 64: the last upsampler) take 128 x 64.
    static const ConvtTile kB = {2, 2, 1, 4}, kC = {2, 1, 1, 4}, kA = {1, 1, 1, 4}, kD = {2, 1, 2, 2};
    ConvtTile t;
    if ((n_cols32 & 7) == 0 && 2 * convt_blocks(kB, a.n_idx, n_cols32, a.B) >= 3 * n_cu) t = kB;
    else if ((n_cols32 & 3) == 0 && convt_blocks(kC, a.n_idx, n_cols32, a.B) >= 2 * n_cu) t = kC;
    else if ((n_cols32 & 3) == 0) t = kA;
    else t = kD;
    const int rows = t.WR * t.MT * 32;
    a.n_row_tiles = (a.n_idx + rows - 1) / rows;
    a.n_col_blk = n_cols32 / (t.WC * t.NT);
    const long long n_tiles = (long long)a.n_row_tiles * a.n_col_blk * a.B;
    if (n_tiles > 0x3fffffffLL) return hipErrorInvalidValue;
    a.n_tiles = (int)n_tiles;
    // XCD order where column blocks share a window and the eight XCDs get the same number of row items (+-12.5 %)
    const long long n_items = (long long)a.n_row_tiles * a.B, items8 = ((n_items + 7) / 8) * 8;
    a.n_items = (int)n_items;
    a.xcd_order = IRIS_CONVT_XCD_ORDER && a.n_col_blk > 1 && (items8 - n_items) * 8 <= n_items;
    const int xm = a.xcd_order ? 8 : 1;
    a.jobs_per_xcd = a.xcd_order ? (int)(items8 / 8) * a.n_col_blk : (int)n_tiles;
    // persistent grid: at most per_cu blocks per CU, evened out so that every block walks the same number of tiles (+-1)
    // (what the registers allow: 64 x 256 blocks 161-200 VGPRs; one input 110-127; three-input staging 135-152, 200 at 128 x 64)
    const bool three_in = a.x1 != nullptr;
    const long long per_cu = t.NT == 2 ? 2 : (three_in ? (t.WR == 2 ? 2 : 3) : IRIS_CONVT_PER_CU);
    const long long cap = per_cu * (n_cu / xm);
    long long g = a.jobs_per_xcd < cap ? a.jobs_per_xcd : cap;
    if (g < 1) g = 1;
    const long long rounds = (a.jobs_per_xcd + g - 1) / g;
    g = (a.jobs_per_xcd + rounds - 1) / rounds;
    g *= xm;
    const size_t lds_bytes = ((size_t)(rows + taps - 1) * 68 + (size_t)a.C_out) * sizeof(float);
    dim3 grid((unsigned)g, 1u, 1u), block(256);
    if (taps != 2) return hipErrorInvalidValue;
    const bool three = three_in;
#define IRIS_CONVT_CASE(MT_, NT_, WR_, WC_)                                                                                   \
    if (t.MT == MT_ && t.NT == NT_ && t.WR == WR_) {                                                                          \
        if (three) return ::iris::launch_kernel_named("convt_mfma_f32_kernel<" #MT_ ", " #NT_ ", " #WR_ ", " #WC_ ", 2, 3>", \
                                           convt_mfma_f32_kernel<MT_, NT_, WR_, WC_, 2, 3>, grid, block, lds_bytes, stream, a); \
        return ::iris::launch_kernel_named("convt_mfma_f32_kernel<" #MT_ ", " #NT_ ", " #WR_ ", " #WC_ ", 2, 1>",            \
                                           convt_mfma_f32_kernel<MT_, NT_, WR_, WC_, 2, 1>, grid, block, lds_bytes, stream, a); \
    }
    IRIS_CONVT_CASE(2, 2, 1, 4)
    IRIS_CONVT_CASE(2, 1, 1, 4)
    IRIS_CONVT_CASE(1, 1, 1, 4)
    IRIS_CONVT_CASE(2, 1, 2, 2)
#undef IRIS_CONVT_CASE
    return hipErrorInvalidValue;
}
#endif  // IRIS_KERNELS_ONLY

}  // namespace iris
