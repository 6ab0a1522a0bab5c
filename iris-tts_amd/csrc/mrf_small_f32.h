// mrf_small_f32.h -- the MRF conv step for SMALL problems (short utterances, streaming windows): latency, not throughput.
//
// Same layer as mrf_conv_mfma_f32.h -- one conv step of the three ResBlock branches of a stage,
//     step 2m   : xt_j = Conv1d_dil(LeakyReLU(x_j))            step 2m+1 : x_j = Conv1d(LeakyReLU(xt_j)) + x_j
// (reference src/iris/hifigan_pretrained.py:64-71,130-136) -- and the SAME bits: every output element is the same
// k-ordered fp32 fmaf chain.  What changes is the shape of a job.
//
// Why: at 100 frames a stage-0 step has 25 row tiles of 32 rows: the persistent kernel's unit of work -- one wave = one
// 32 x 32 output tile with the whole K = 11 * 256 reduction -- is a 90,000-cycle dependent chain of v_mfma_f32_32x32x2_f32
// (64 cycles each, 2 k-steps per instruction), and a launch cannot be shorter than its longest job whatever the other 80 %
// of the SIMDs do.  v_mfma_f32_16x16x4_f32 advances a chain by 4 k-steps every 32-40 cycles on a quarter of the tile:
// a wave owns 16 rows x 16 channels, so the same step is cut into 4x as many jobs whose chains are 3-4x shorter.
// Bit-compatibility: the instruction sums its four k-steps as a sequential fmaf chain in k order (checked on the
// hardware, tools/mfma_order_check.hip); the 32 x 32 kernels walk a group of 8 channels in the order 0,4,1,5,2,6,3,7, so
// here K-quarter kq is fed channel {0,4,1,5}[kq] by the first and {2,6,3,7}[kq] by the second instruction of a group.
//
// Work split: a 256-thread block = 64 rows x 16 output channels of one branch of one batch item; its four waves take
// 16 rows each and stream the SAME weight fragments (the first to ask brings them into the CU's L1).  (Blocks of 32 / 64
// channels -- 512 / 1024 threads sharing one staged window -- were measured: no gain where the kernel is used, stage 0
// at <= ~120 frames, and a loss of 18-58 % there because they leave CUs empty.)  The input
// window (64 + (k-1) d rows, 64 channels at a time, LeakyReLU applied on the way) goes through LDS as in the other
// kernels (row stride 68 floats, staged through registers one chunk ahead).  Weights come from a second packing in this
// instruction's fragment order (pack_conv1d_weights16: one 16-byte load per lane feeds four MFMAs -- gathering them
// from the 32 x 32 packing would need twice the L1 bandwidth, and the CU's 64 B/clk is what this kernel runs into first);
// activations are read with ds_read_b128, of which a K-quarter uses two floats.  The accumulator is 4 registers; weight
// fragments are requested kSmallRing steps ahead.  In the D layout a lane holds 4 consecutive channels of one row:
// bias, residual and the 16-byte store need no transpose.
// The branches of a step run in different blocks, so the MRF mean of a stage's last step is left to the consumer
// (as in the persistent kernel's one-branch-per-block mode).
#pragma once
#include "mrf_conv_mfma_f32.h"

namespace iris {

constexpr int kSmallRows = 64;     // rows per block (4 waves x 16)
constexpr int kSmallCic = 64;      // input channels staged per chunk
constexpr int kSmallRing = 8;      // weight fragments in flight per wave (steps of 16 channels = 4 MFMAs)

__global__ void __launch_bounds__(256) mrf_small_f32_kernel(const ConvLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = kSmallCic + 4;
    constexpr int QPR = kSmallCic / 4;
    constexpr int NTHR = 256;
    constexpr int NQ = ((kSmallRows + kMrfSpanMax) * QPR + NTHR - 1) / NTHR;   // staged 16-byte quads per thread and chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // wave: 16-row sub-tile
    const int t = lane & 15, kq = lane >> 4;
    const int L = a.L_in, C = a.C_in;
    // job = (row block, 16-channel tile, branch): branch slowest so that the heavy k = 11 jobs are dispatched first
    const int n_rb = (L + kSmallRows - 1) / kSmallRows;
    const int n_ct16 = a.C_out >> 4;
    const int per_branch = n_rb * n_ct16;
    const int zr = blockIdx.x / per_branch;
    const int rem = blockIdx.x - zr * per_branch;
    const int z = a.nz_serial - 1 - zr;
    const int rb = rem / n_ct16, ct16 = rem - rb * n_ct16;
    ConvProblem p = a.p[0];
    if (z == 1) p = a.p[1];
    if (z == 2) p = a.p[2];
    const int b = blockIdx.y;
    const int ks = p.ks, dil = p.dil;
    const int i0 = rb * kSmallRows;
    const int R = kSmallRows + (ks - 1) * dil;
    const int in_row0 = i0 - p.pad_left;
    const size_t item = (size_t)b * L * C;
    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + item, tensor_bytes);
    const float slope = a.slope;

    // weights in 16x16x4 fragment order (pack_conv1d_weights16): one 16-byte load per lane = the A operands of the four
    // MFMAs that cover 16 input channels.  [tap][16-channel step gp][16-wide C_out tile][lane]
    const int ngp = C >> 4;
    const unsigned gp_bytes = (unsigned)n_ct16 * 1024u;               // bytes per (tap, 16-channel step)
    const unsigned tap_bytes = (unsigned)ngp * gp_bytes;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp16, (unsigned)ks * tap_bytes);
    const unsigned wvoff = (unsigned)(ct16 * 64 + lane) * 16u;
    // activations: row (16 wave + t) of the window; K-quarter kq uses channels {0,4,1,5}[kq] and +2 of each group of 8:
    // the 16 bytes at channel 8g + 4 (kq & 1), components (kq >> 1) and (kq >> 1) + 2
    const float* b_lane = lds + (wave * 16 + t) * S + 4 * (kq & 1);
    const int e0 = kq >> 1;

    // window staging through registers: the quads of chunk c+1 are requested before the K loop of chunk c
    const int r_lane = tid / QPR, q_lane = tid - r_lane * QPR;
    constexpr int RPI = NTHR / QPR;
    f32x4 st[NQ];
    auto stage_request = [&](int c0) {
        const int ci = c0 + 4 * q_lane;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int r = r_lane + i * RPI, row = in_row0 + r;
            const bool ok = r < R && row >= 0 && row < L && ci < C;
            st[i] = buf_load4(xr, ok ? (unsigned)(row * C + ci) * 4u : kOobOffset, 0);
        }
    };
    auto stage_write = [&]() {             // LeakyReLU(x) as max(x, slope x) (0 <= slope <= 1 checked by the host)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int r = r_lane + i * RPI;
            if (r < R) {
                f32x4 v = st[i];
                v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
                v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
                *reinterpret_cast<f32x4*>(lds + r * S + 4 * q_lane) = v;
            }
        }
    };

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int n_chunks = (C + kSmallCic - 1) / kSmallCic;
    stage_request(0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int c0 = chunk * kSmallCic;
        if (chunk > 0) __syncthreads();                             // every wave is done with the previous window
        stage_write();
        // ---- K loop over (tap, 16-channel step): four 16x16x4 MFMAs per step; weights kSmallRing steps ahead ----
        // steps of this chunk: 4, or 2 when only 32 channels are left (C % 32 == 0 is required): shifts, no division
        const int ssh = (C - c0) >= kSmallCic ? 2 : 1, smask = (1 << ssh) - 1;
        const int NS = ks << ssh;
        const unsigned wsoff0 = (unsigned)(c0 >> 4) * gp_bytes;
        auto w_soff = [&](int n) -> unsigned {      // step n of this chunk: tap n >> ssh, step n & smask (past the end: out of range)
            return n < NS ? wsoff0 + (unsigned)(n >> ssh) * tap_bytes + (unsigned)(n & smask) * gp_bytes : kOobOffset;
        };
        f32x4 wv[kSmallRing];
#pragma unroll
        for (int i = 0; i < kSmallRing; ++i) wv[i] = buf_load4(wr, wvoff, w_soff(i));
        // (vmcnt retires in order: the next chunk's window is requested BEHIND the first weight fragments, so only the
        //  waits of fragments requested inside the loop -- kSmallRing steps later -- can be held up by it)
        if (chunk + 1 < n_chunks) stage_request(c0 + kSmallCic);
        __syncthreads();                                             // the window is in place
        f32x4 bv[2][2];
        bv[0][0] = *reinterpret_cast<const f32x4*>(b_lane);
        bv[0][1] = *reinterpret_cast<const f32x4*>(b_lane + 8);
        for (int n0 = 0; n0 < NS; n0 += kSmallRing) {
#pragma unroll
            for (int i = 0; i < kSmallRing; ++i) {
                const int n = n0 + i;
                if (n < NS) {                                   // wave-uniform
                    const int nn = n + 1 < NS ? n + 1 : n;
                    const float* bp = b_lane + (nn >> ssh) * dil * S + 16 * (nn & smask);
                    bv[(i + 1) & 1][0] = *reinterpret_cast<const f32x4*>(bp);
                    bv[(i + 1) & 1][1] = *reinterpret_cast<const f32x4*>(bp + 8);
                    const f32x4 w4 = wv[i], x0 = bv[i & 1][0], x1 = bv[i & 1][1];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, e0 ? x0.y : x0.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, e0 ? x0.w : x0.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, e0 ? x1.y : x1.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, e0 ? x1.w : x1.z, acc, 0, 0, 0);
                    wv[i] = buf_load4(wr, wvoff, w_soff(n + kSmallRing));
                }
            }
        }
    }
    // ---- epilogue: D[row 4 kq + r][col t]: this lane has channels co0 .. co0+3 of row i0 + 16 wave + t ----
    const int row = i0 + wave * 16 + t;
    const int co0 = ct16 * 16 + 4 * kq;
    if (row < L) {
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + co0);
        const size_t off = item + (size_t)row * C + co0;
        f32x4 res4 = {0.f, 0.f, 0.f, 0.f};
        if (p.res) res4 = *reinterpret_cast<const f32x4*>(p.res + off);
        f32x4 o;
        o.x = (acc.x + bias4.x) + res4.x; o.y = (acc.y + bias4.y) + res4.y;
        o.z = (acc.z + bias4.z) + res4.z; o.w = (acc.w + bias4.w) + res4.w;
        *reinterpret_cast<f32x4*>(p.y + off) = o;
    }
}

// The small kernel takes the shapes the persistent kernel takes, with C a multiple of 32.
inline bool mrf_small_applicable(const ConvLaunch& a, int nz) {
    if (!mrf_kernel_applicable(a, nz) || a.sum_y) return false;
    if ((a.C_in & 31) || a.B > 65535) return false;
    for (int j = 0; j < nz; ++j)
        if (!a.p[j].wp16) return false;                 // the handle keeps this second packing only for the MRF layers
    return true;
}

// Estimated duration of a step in matrix-pipe cycles, for the choice against mrf_plan's modes: the MFMAs of all jobs
// spread over every SIMD at the 32-cycle issue rate, but never less than the longest single chain (40-cycle dependent
// latency), plus the window staging of a block per chunk.  In practice this selects the kernel for the C = 256 stage of
// inputs up to ~120 frames (measured: 68.8 -> 40.6 us per step at 100 frames; at 282 frames it would still be 12 % ahead
// on that stage, at C = 128 it never is: 58.6 vs 55.0 us at 100 frames) -- profiles/r02_notes.md.
inline double mrf_small_cycles(const ConvLaunch& a, int nz) {
    const int n_cu = device_cu_count();
    const double tiles16 = (double)((a.L_in + 15) / 16) * (a.C_out / 16) * a.B;
    double taps = 0, kmax = 0;
    for (int j = 0; j < nz; ++j) { taps += a.p[j].ks; if (a.p[j].ks > kmax) kmax = a.p[j].ks; }
    const double per_simd = tiles16 * taps * (a.C_in / 4.0) / (4.0 * n_cu);
    const double chain = kmax * (a.C_in / 4.0);
    const double chunks = (a.C_in + kSmallCic - 1) / kSmallCic;
    // blocks a CU walks through one after the other, each paying its chunks' staging (latency + conversion) partly exposed
    const double blocks = (double)((a.L_in + kSmallRows - 1) / kSmallRows) * (a.C_out >> 4) * nz * a.B;
    const double waves_per_cu = blocks * 4.0 / n_cu;
    const double rounds = waves_per_cu > 16.0 ? waves_per_cu / 16.0 : 1.0;
    return (per_simd * 32.0 > chain * 40.0 ? per_simd * 32.0 : chain * 40.0) * 1.15 + rounds * chunks * 4300.0;
}

inline hipError_t launch_mrf_small(ConvLaunch& a, int nz, hipStream_t stream) {
    a.n_co_blk = 1;
    a.Gp = packed_groups(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    a.z_serial = 0;
    a.nz_serial = nz;
    a.nz = nz;
    a.dyn_counter = nullptr;
    int span = 0;
    for (int j = 0; j < nz; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
    }
    const long long blocks = (long long)((a.L_in + kSmallRows - 1) / kSmallRows) * (a.C_out >> 4) * nz;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)(kSmallRows + span) * (kSmallCic + 4) * sizeof(float);
    dim3 grid((unsigned)blocks, (unsigned)a.B), block(256);
    return ::iris::launch_kernel(mrf_small_f32_kernel, grid, block, lds_bytes, stream, a);
}

}  // namespace iris
