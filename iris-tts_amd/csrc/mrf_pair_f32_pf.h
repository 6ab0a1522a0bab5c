// mrf_pair_f32_pf.h -- the fused fp32 conv pair (mrf_pair_f32.h) as PERSISTENT blocks that fetch the next job's window
// while the current job still computes, and -- as the LAST pair of a stage -- form the MRF mean themselves.
//
// Reference semantics: one iteration of ResBlock.forward's loop (src/iris/hifigan_pretrained.py:64-71)
//     xt = Conv1d_{k, dil d}(LeakyReLU(x));  y = Conv1d_{k, dil 1}(LeakyReLU(xt)) + x
// for the branches k = 3 / 7 / 11 of a stage (:130-136); SUM = true adds the MRF fusion of :131-137,
//     x = ((y_0 + y_1) + y_2) / num_kernels   (the reference's order and its true division),
// and stores only that.  Every element is the fmaf chain of mrf_conv_mfma_f32_kernel / mrf_pair_f32_kernel
// (v_mfma_f32_32x32x2_f32, taps ascending, groups of 8 channels ascending, channels 0,4,1,5,2,6,3,7 inside a group;
// xt = (acc + b1) + 0, y = (acc + b2) + x; the sum as in the persistent kernel's summing step): bit-identical to the separate
// launches, so which path a forward takes never changes a sample.
//
// What changes against mrf_pair_f32_kernel:
//   * blocks are persistent: a block starts with job blockIdx.x and draws every further job from a device counter (longest
//     jobs first, first come first served); the biases of all branches sit in LDS for the block's life;
//   * the next job's window is requested into registers during the last DB groups of conv2 (vmcnt retires in order: those
//     groups request no weights of this conv any more, so nothing younger is waited for before the window itself), flies
//     during the epilogue -- which needs no LDS here: 16-byte pieces straight from the D layout -- and is written to LDS
//     behind it; the weight ring's free slots take the first fragments of the conv that runs next;
//   * SUM: a block runs the three branches of ITS tile back to back in the reference's order (k = 3, 7, 11) with a common
//     output height M - (k_max - 1), keeps the running sum in registers and stores the mean: the stage's last pair no
//     longer needs the persistent kernel's two launches (conv1 step + summing step), and the per-branch outputs and xt of
//     that pair never reach HBM.
//
// RESULT (round 3, profiles/r03_notes.md; release and diagnostic builds, MI355X):
//   * non-summing pairs: 3-5 % SLOWER than one block per job at every size tried (batch 1 x 282 ... 32 x 500 frames, fixed
//     stride or drawn jobs, C = 32 at the same four blocks per CU, C = 64 at two instead of three: the window held in
//     registers does not fit 168 VGPRs there).  What the prefetch hides (a window wait that co-resident blocks already
//     cover) is less than what the job loop adds (a fourth barrier per job, the decode, the ring hand-over).  The forward
//     therefore keeps mrf_pair_f32_kernel for them; this path stays reachable through iris_hifigan_op_mrf_pair (modes 1 / 2,
//     parity tests) and the diagnostic build's IRIS_HIFIGAN_PAIR_PF_MODE=2.
//   * SUM: wins where its whole-tile jobs (21 tap-units against 11 / 7 / 3) fill several rounds of the chip: batch 32 x 500
//     frames, C = 32 stage +3 %, step -0.6 %; neutral at batch 1 x 1000; a loss at one or two rounds.  The forward takes it
//     from four rounds on (iris_hifigan.hip).
#pragma once
#include <type_traits>
#include "mrf_pair_f32.h"

namespace iris {

struct PairPfLaunchF32 {
    PairProblemF32 p[kMaxGroup];
    int B, L, C;
    float slope;
    int nz;               // branches
    int Gp, n_ct;         // packed-weight geometry
    int tiles[kMaxGroup]; // tiles per batch item of branch z: ceil(L / (M - (k_z - 1))); SUM: the same for all, from t_out
    int t_out;            // SUM: common output rows per tile, M - (k_max - 1); else 0 (per branch: M - (k - 1))
    int n_jobs;           // (tile, branch) jobs over the whole batch; SUM: tiles over the whole batch (< 2^30: pair_pf_f32_applicable)
    unsigned* next_job;   // zeroed device word: blocks draw their 2nd, 3rd, ... job from it (nullptr: fixed stride of gridDim.x)
    int bias_off;         // float offset of the bias table in LDS
    float* sum_y;         // SUM: [B, L, C] mean of the branch outputs
    float sum_div;        // SUM: num_kernels
};

// One conv over the LDS window, unrolled for KS taps: NG = KS * GPC groups of 8 channels.  Ring slot n % (DB+1) holds
// group n; group n + DB is requested while group n computes.  In the last DB groups the slots that fall free take the
// groups 0 .. DB-1 of the conv that runs NEXT (descriptor wr_next); afterwards the ring is rotated so that the next conv
// finds its group d in slot d.  tail(t), t = 0 .. DB-1, is called in those last groups (extra requests).
template <int KS, int MT, int GPC, int DB, int TAIL_VM, class Tail>
__device__ __forceinline__ void pair_pf_f32_mma(f32x16 (&acc)[MT], f32x4 (&bw)[DB + 1], const float* aptr, int dilS, int S,
                                                __amdgpu_buffer_rsrc_t wr, __amdgpu_buffer_rsrc_t wr_next, unsigned wvoff,
                                                unsigned wbytes_group, unsigned tap_bytes, Tail tail) {
    constexpr int NG = KS * GPC;
    static_assert(NG >= DB && DB <= GPC, "ring deeper than a tap");
    auto a_ptr = [&](int n) { return aptr + (n / GPC) * dilS + 8 * (n % GPC); };
    f32x4 av[2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(a_ptr(0) + m * 32 * S);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        if (n + DB < NG)
            bw[(n + DB) % (DB + 1)] = buf_load4(wr, wvoff, (unsigned)((n + DB) / GPC) * tap_bytes + (unsigned)((n + DB) % GPC) * wbytes_group);
        else {
            bw[(n + DB) % (DB + 1)] = buf_load4(wr_next, wvoff, (unsigned)(n + DB - NG) * wbytes_group);   // n + DB - NG < DB <= GPC: tap 0
            tail(n + DB - NG);
        }
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
            int ds_left = (n + 1 < NG) ? MT : 0, vm_left = 1 + (n + DB < NG ? 0 : TAIL_VM);   // (TAIL_VM: most requests a tail group adds)
#pragma unroll
            for (int k = 0; k < 4 * MT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // the next conv expects its groups 0 .. DB-1 in slots 0 .. DB-1: they were loaded into slots (NG + d) % (DB+1)
        f32x4 tmp[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d) tmp[d] = bw[(NG + d) % (DB + 1)];
#pragma unroll
        for (int d = 0; d < DB; ++d) bw[d] = tmp[d];
    }
}

template <class T>
__device__ __forceinline__ T* uniform_ptr_f32(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

template <int WT, int WC, int MT, int C, int MINB, bool SUM>
__global__ void __launch_bounds__(256, MINB) mrf_pair_f32_pf_kernel(const PairPfLaunchF32 a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(WT * WC == 4 && WC * 32 == C, "a block owns all C channels");
    constexpr int S = C + 4;
    constexpr int QPR = C / 4;                                            // 16-byte quads per window row
    constexpr int GPC = C / 8;
    constexpr int M = WT * MT * 32;
    constexpr int DB = 4;
    constexpr int RPI = 256 / QPR;                                        // window rows covered by one quad per thread
    constexpr int NQ = (M + kMrfSpanMax + RPI - 1) / RPI;                // staged quads per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int L = a.L;
    const float slope = a.slope;
    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 4u;
    const unsigned wbytes_group = (unsigned)a.n_ct * 64u * 16u;
    const unsigned tap_bytes = (unsigned)a.Gp * wbytes_group;
    const unsigned wvoff = (unsigned)(wc * 64 + lane) * 16u;              // this wave's 32-wide channel tile = wc
    const float* aptr = lds + (wt * MT * 32 + lo) * S + 4 * hi;
    const int co4 = wc * 32 + 4 * hi;                                     // this lane's channels: co4 + 8g + {0..3}

    // window staging: thread -> (row r_lane + u*RPI, quad q_lane); byte offset of quad u = vbase + u*row_step
    // (rows < 0 wrap to >= 2^31, rows >= L exceed num_records: both read 0 -- the convolution's zero padding)
    const int r_lane = tid / QPR, q_lane = tid - r_lane * QPR;
    constexpr unsigned row_step = (unsigned)(RPI * C) * 4u;
    float* const lds_wr = lds + r_lane * S + 4 * q_lane;

    // biases of every branch, both convs, in LDS for the block's life: [z][conv][C]
    float* const lds_bias = lds + a.bias_off;
#pragma unroll
    for (int z = 0; z < kMaxGroup; ++z)
        if (z < a.nz && tid < C / 2) {
            const float* src = tid < C / 4 ? a.p[z].b1 + 4 * tid : a.p[z].b2 + 4 * (tid - C / 4);
            *reinterpret_cast<f32x4*>(lds_bias + z * 2 * C + 4 * tid) = *reinterpret_cast<const f32x4*>(src);
        }

    // ---- jobs ---------------------------------------------------------------------------------------------------
    // !SUM: jobs are (tile, branch) pairs in the order ALL jobs of the heaviest branch (z = nz-1: k = 11) first, then k = 7,
    //       then k = 3, each branch's tiles over the whole batch.  SUM: job = a tile (b * tiles + tile); the block runs
    //       z = 0, 1, .., nz-1 on it (the reference's summation order).
    // A block starts with job blockIdx.x and draws every further job from a device counter (longest jobs first + first come
    // first served = the launch ends on short jobs, whatever speed each CU runs at); without a counter it walks a fixed stride.
    const int G = (int)gridDim.x;
    struct Step { const float* x; const f32x4* w1; const f32x4* w2; float* y; int ks, dil, b, o0, z, job; };
    // (always_inline: called from two places; left out of line, the closure's pointer to the kernel arguments would force
    //  hipcc to keep them -- and everything derived from them -- in scratch instead of SGPRs)
    auto load_step = [&](int job, int z_sum, Step& s) __attribute__((always_inline)) {
        int z = z_sum, t = job;
        if (!SUM) {
            // segments of a.tiles[z] * B jobs, heaviest branch (z = nz-1) first; constant indices keep the kernel arguments in SGPRs
            z = a.nz - 1;
#pragma unroll
            for (int zz = kMaxGroup - 1; zz >= 1; --zz)
                if (z == zz) {
                    const int seg = a.tiles[zz] * a.B;
                    if (t >= seg) { t -= seg; z = zz - 1; }
                }
        }
        z = __builtin_amdgcn_readfirstlane(z);
        t = __builtin_amdgcn_readfirstlane(t);
        // The branch's problem out of the kernel arguments.  Every field is read unconditionally and pinned to an SGPR
        // (readfirstlane) BEFORE the selection: hipcc turns a plain `p = a.p[0]; if (z == 1) p = a.p[1]; ...` into a lookup
        // in a private copy of the arguments, whose loads are per-lane values -- every buffer descriptor built from them
        // then needs a waterfall loop and the job loop becomes exec-masked control flow.
        auto sel_i = [&](int v0, int v1, int v2, int v3) __attribute__((always_inline)) {
            int v = __builtin_amdgcn_readfirstlane(v0);
            if (z == 1) v = __builtin_amdgcn_readfirstlane(v1);
            if (z == 2) v = __builtin_amdgcn_readfirstlane(v2);
            if (z == 3) v = __builtin_amdgcn_readfirstlane(v3);
            return v;
        };
        auto sel_p = [&](auto p0, auto p1, auto p2, auto p3) __attribute__((always_inline)) {
            auto v = uniform_ptr_f32(p0);
            if (z == 1) v = uniform_ptr_f32(p1);
            if (z == 2) v = uniform_ptr_f32(p2);
            if (z == 3) v = uniform_ptr_f32(p3);
            return v;
        };
        const int tiles_z = SUM ? __builtin_amdgcn_readfirstlane(a.tiles[0]) : sel_i(a.tiles[0], a.tiles[1], a.tiles[2], a.tiles[3]);
        const int bb = t / tiles_z, tile = t - bb * tiles_z;
        s.x = sel_p(a.p[0].x, a.p[1].x, a.p[2].x, a.p[3].x);
        s.w1 = sel_p(a.p[0].w1, a.p[1].w1, a.p[2].w1, a.p[3].w1);
        s.w2 = sel_p(a.p[0].w2, a.p[1].w2, a.p[2].w2, a.p[3].w2);
        s.y = sel_p(a.p[0].y, a.p[1].y, a.p[2].y, a.p[3].y);
        s.ks = sel_i(a.p[0].ks, a.p[1].ks, a.p[2].ks, a.p[3].ks);
        s.dil = sel_i(a.p[0].dil, a.p[1].dil, a.p[2].dil, a.p[3].dil);
        s.b = bb;
        s.o0 = tile * (SUM ? a.t_out : M - (s.ks - 1));
        s.z = z; s.job = job;
    };
    int* const next_slot = reinterpret_cast<int*>(lds_bias + a.nz * 2 * C);     // LDS word: the job drawn for this block's next step
    auto window_vbase = [&](const Step& s) -> unsigned {
        const int in_row0 = s.o0 - (s.ks - 1) / 2 - s.dil * (s.ks - 1) / 2;
        return (unsigned)((in_row0 + r_lane) * C + 4 * q_lane) * 4u;
    };

    f32x4 st[NQ];                                                         // the window in flight
    auto window_request_one = [&](int u, __amdgpu_buffer_rsrc_t xr, unsigned vbase, int R) {
        // rows past this branch's window are not requested: they would be real rows, i.e. HBM reads nobody uses
        st[u] = buf_load4(xr, r_lane + u * RPI < R ? vbase + (unsigned)u * row_step : kOobOffset, 0);
    };
    auto window_write = [&](int R) {                                      // LeakyReLU(x) = max(x, slope x), 0 <= slope <= 1
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (r_lane + u * RPI < R) {
                f32x4 v = st[u];
                v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
                v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
                *reinterpret_cast<f32x4*>(lds_wr + u * RPI * S) = v;
            }
    };

    f32x16 acc[MT];
    f32x16 sumv[MT];                                                      // SUM: running sum of the branch outputs of this tile
    f32x4 bw[DB + 1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    };

    Step J;
    if ((int)blockIdx.x >= a.n_jobs) return;
    load_step((int)blockIdx.x, 0, J);
    // ---- prologue: the only window wait a block exposes ---------------------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(J.x + (size_t)J.b * L * C, tensor_bytes);
        const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(J.w1, (unsigned)J.ks * tap_bytes);
#pragma unroll
        for (int d = 0; d < DB; ++d) bw[d] = buf_load4(wr1, wvoff, (unsigned)d * wbytes_group);
        const unsigned vb = window_vbase(J);
        const int R = M + (J.ks - 1) * J.dil;
#pragma unroll
        for (int u = 0; u < NQ; ++u) window_request_one(u, xr, vb, R);
        window_write(R);
    }
    __syncthreads();

    for (;;) {
        // the step after this one: the next branch of the same tile (SUM), or a new job -- known only after the draw below
        Step Jn = J;
        bool more = false;
        const bool new_job = !SUM || J.z + 1 >= a.nz;
        int drawn = J.job + G;                                            // fixed stride, unless a counter is given
        const int ks = J.ks, dil = J.dil;
        const int h2 = (ks - 1) / 2;
        const int T_OUT = SUM ? a.t_out : M - (ks - 1);
        const int o0 = J.o0;
        const size_t item = (size_t)J.b * L * C;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(J.x + item, tensor_bytes);
        const __amdgpu_buffer_rsrc_t yr = make_rsrc((SUM ? a.sum_y : J.y) + item, tensor_bytes);
        const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(J.w1, (unsigned)ks * tap_bytes);
        const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(J.w2, (unsigned)ks * tap_bytes);
        __amdgpu_buffer_rsrc_t xrn = xr, wr1n = wr1;
        unsigned vbn = 0;
        int Rn = 0;
        const float* const b1p = lds_bias + (J.z * 2 + 0) * C + co4;
        const float* const b2p = lds_bias + (J.z * 2 + 1) * C + co4;
        constexpr int PER = (NQ + DB - 1) / DB;                           // window requests per tail group

        auto iteration = [&](auto ks_tag) {
            constexpr int KS = decltype(ks_tag)::value;
            // (a distinct marker per kernel size: hipcc otherwise hoists the three paths' common head above the dispatch and
            //  spills it to scratch across the branch)
            asm volatile("; fp32 conv pair, %0 taps" :: "n"(KS) : "memory");
            // ---- conv1; its tail fetches conv2's first fragments ----------------------------------------------------------
            zero_acc();
            pair_pf_f32_mma<KS, MT, GPC, DB, 0>(acc, bw, aptr, dil * S, S, wr1, wr2, wvoff, wbytes_group, tap_bytes, [](int) {});
            // draw the next job (one lane; the answer is picked up behind the xt step)
            if (a.next_job && new_job && tid == 0) drawn = G + (int)atomicAdd(a.next_job, 1u);
            // the residual pieces travel during step 3 and conv2
            unsigned ovoff[MT];                                           // piece (m, g = 0) of this lane, or out of range
            f32x4 resv[MT * 4];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int im = (wt * MT + m) * 32 + lo;
                const int o = o0 + im;
                ovoff[m] = (im < T_OUT && o < L) ? (unsigned)(o * C + co4) * 4u : kOobOffset;
#pragma unroll
                for (int g = 0; g < 4; ++g) resv[m * 4 + g] = buf_load4(xr, ovoff[m], (unsigned)(8 * g) * 4u);
            }
            __syncthreads();                                              // every wave is done with the x window
            // ---- xt -> LDS: LeakyReLU((acc + bias1) + 0), zero outside [0, L) (conv2's zero padding) -------------------------
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row_l = (wt * MT + m) * 32 + lo;
                const int row_g = o0 - h2 + row_l;
                const bool inside = row_g >= 0 && row_g < L;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1p + 8 * g);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (acc[m][4 * g + e] + b4[e]) + 0.f;    // what the separate launch stores (residual-free: + 0)
                        v[e] = inside ? fmaxf(t, t * slope) : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(lds + row_l * S + co4 + 8 * g) = v;
                }
            }
            if (a.next_job && new_job && tid == 0) *next_slot = drawn;
            __syncthreads();
            // the step that follows (zero-length descriptors when there is none: such loads return 0 without touching memory)
            {
                int nj = J.job;
                if (new_job) nj = __builtin_amdgcn_readfirstlane(a.next_job ? *next_slot : drawn);
                more = nj < a.n_jobs;
                if (more) load_step(nj, new_job ? 0 : J.z + 1, Jn);
                xrn = make_rsrc(Jn.x + (size_t)Jn.b * L * C, more ? tensor_bytes : 0u);
                wr1n = make_rsrc(Jn.w1, more ? (unsigned)Jn.ks * tap_bytes : 0u);
                vbn = window_vbase(Jn);
                Rn = M + (Jn.ks - 1) * Jn.dil;
            }
            // ---- conv2 (dilation 1); its tail requests the NEXT window and (ring) the next step's first conv1 fragments -------
            zero_acc();
            pair_pf_f32_mma<KS, MT, GPC, DB, PER>(acc, bw, aptr, S, S, wr2, wr1n, wvoff, wbytes_group, tap_bytes, [&](int t) {
#pragma unroll
                for (int u = 0; u < NQ; ++u)
                    if (u / PER == t) window_request_one(u, xrn, vbn, Rn);
            });
            // ---- epilogue: (acc + bias2) + x, 16-byte pieces straight from the D layout (no LDS) -----------------------------
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(b2p + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[m][4 * g + e] = (acc[m][4 * g + e] + b4[e]) + resv[m * 4 + g][e];
                }
            if constexpr (SUM) {
                // the MRF fusion (hifigan_pretrained.py:131-137) in the reference's order: xs = rb0; xs += rb1; xs += rb2; x = xs / 3
                if (J.z == 0) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) sumv[m] = acc[m];
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sumv[m][r] = sumv[m][r] + acc[m][r];
                }
                if (J.z == a.nz - 1) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sumv[m][r] = sumv[m][r] / a.sum_div;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!SUM || J.z == a.nz - 1) {
                f32x4 outv[MT * 4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& src = SUM ? sumv[m] : acc[m];
                        outv[m * 4 + g] = f32x4{src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int idx = 0; idx < MT * 4; ++idx) buf_store4(outv[idx], yr, ovoff[idx / 4], (unsigned)(8 * (idx % 4)) * 4u);
                asm volatile("s_nop 1");       // explicit wait states behind the dwordx4 store group (see mrf_conv_mfma_f32.h)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int idx = 0; idx < MT * 4; ++idx) asm volatile("" :: "v"(outv[idx]));
            }
        };
        if (ks == 3)      iteration(std::integral_constant<int, 3>{});
        else if (ks == 7) iteration(std::integral_constant<int, 7>{});
        else              iteration(std::integral_constant<int, 11>{});
        if (!more) break;
        __syncthreads();                                                  // every wave is done reading xt
        window_write(Rn);
        __syncthreads();
        J = Jn;
    }
}

// ---- launch ----------------------------------------------------------------------------------------
#ifndef IRIS_PAIR_F32_PF_DEFAULT
#define IRIS_PAIR_F32_PF_DEFAULT 1
#endif

struct PairPfTileF32 { int WT, WC, MT, MINB, M; };

inline PairPfTileF32 pair_pf_f32_tile(int C) {
    if (C == 32) return PairPfTileF32{4, 1, 1, 4, 128};
    return PairPfTileF32{2, 2, 2, 2, 128};      // (three blocks per CU would spill: 48 staging + 32 residual + 64 accumulator/sum registers)
}

// Blocks of a persistent launch of `jobs` jobs.  Jobs of a SUM launch are equal and indivisible (a tile's three branches),
// so the launch takes whole rounds: with b resident blocks per CU it lasts ceil(jobs / (n_cu * b)) rounds of b jobs sharing a
// CU's matrix pipes, i.e. ~ ceil(jobs / (n_cu * b)) * b job times.  The block count per CU (<= MINB) with the fewest
// job times is taken (a lone block per CU hides latencies worse: penalised); `efficiency` = jobs / n_cu over that.
struct PairPfPlanF32 { long long blocks; int per_cu; double efficiency; };
inline PairPfPlanF32 pair_pf_f32_plan(long long jobs, int n_cu, int minb, bool sum) {
    PairPfPlanF32 pl{1, 1, 1.0};
    if (jobs < 1) jobs = 1;
    if (!sum) {                       // mixed job sizes, heaviest first: every slot, the tail is made of short jobs
        const long long slots = (long long)n_cu * minb;
        pl.blocks = jobs < slots ? jobs : slots;
        pl.per_cu = minb;
        return pl;
    }
    static const double penalty[5] = {0, 1.35, 1.1, 1.0, 1.0};
    double best = 1e30;
    for (int b = 1; b <= minb && b <= 4; ++b) {
        const long long slots = (long long)n_cu * b;
        const long long rounds = (jobs + slots - 1) / slots;
        const double cost = (double)rounds * b * penalty[b];
        if (cost < best) { best = cost; pl.per_cu = b; pl.blocks = jobs < slots ? jobs : (jobs + rounds - 1) / rounds; }
    }
    pl.efficiency = ((double)jobs / n_cu) / best;
    return pl;
}

// True when the pair launch `a` (nz branches) can take the persistent kernel (sum: as the stage's last pair, forming the mean).
inline bool pair_pf_f32_applicable(const PairLaunchF32& a, int nz, bool sum) {
    if (!pair_f32_applicable(a, nz)) return false;
    if (sum && nz != 3) return false;
    {   // 32-bit job indices (the blocks' stride walk adds up to one grid length beyond the last job)
        const PairPfTileF32 t = pair_pf_f32_tile(a.C);
        int kmax = 1;
        for (int j = 0; j < nz; ++j) if (a.p[j].ks > kmax) kmax = a.p[j].ks;
        const long long tiles = (a.L + (t.M - (kmax - 1)) - 1) / (t.M - (kmax - 1));
        if (tiles * a.B * nz > 0x3fffffffLL) return false;
    }
#ifndef IRIS_MRF_DIAG
    if (!sum) return false;        // release build: only the summing form exists (the plain persistent pairs measured 3-5 % slower)
#endif
    return IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_PF", IRIS_PAIR_F32_PF_DEFAULT) != 0;
}

// next_job: a ZEROED device word of this launch's own (blocks draw jobs from it), or nullptr for a fixed stride.
inline hipError_t launch_pair_f32_pf(const PairLaunchF32& src, int nz, float* sum_y, unsigned* next_job, hipStream_t stream) {
    if (src.C != 32 && src.C != 64) return hipErrorInvalidValue;
    PairPfLaunchF32 a;
    memset(&a, 0, sizeof(a));
    for (int j = 0; j < nz; ++j) {
        a.p[j] = src.p[j];
        for (int i = 0; i < nz; ++i)
            if (!sum_y && src.p[j].x == src.p[i].y) return hipErrorInvalidValue;             // never in place
        if (sum_y && src.p[j].x == sum_y) return hipErrorInvalidValue;
    }
    a.B = src.B; a.L = src.L; a.C = src.C; a.slope = src.slope; a.nz = nz;
    a.Gp = packed_groups(a.C);
    a.n_ct = packed_cotiles(a.C);
    a.sum_y = sum_y; a.sum_div = (float)nz;
    int span = 0, kmax = 1;
    for (int j = 0; j < nz; ++j) {
        const int s = (a.p[j].ks - 1) * a.p[j].dil;
        if (s > span) span = s;
        if (a.p[j].ks > kmax) kmax = a.p[j].ks;
    }
    const PairPfTileF32 t = pair_pf_f32_tile(a.C);
    const int t_out_min = t.M - (kmax - 1);
    a.t_out = sum_y ? t_out_min : 0;
    long long n_jobs = 0;
    for (int j = 0; j < nz; ++j) {
        const int t_out = sum_y ? t_out_min : t.M - (a.p[j].ks - 1);
        a.tiles[j] = (a.L + t_out - 1) / t_out;
        n_jobs += (long long)a.tiles[j] * a.B;
    }
    if (sum_y) n_jobs = (long long)a.tiles[0] * a.B;
    if (n_jobs > 0x3fffffffLL) return hipErrorInvalidValue;                                  // (pair_pf_f32_applicable)
    a.n_jobs = (int)n_jobs;
    a.next_job = next_job;
    const PairPfPlanF32 pl = pair_pf_f32_plan(n_jobs, device_cu_count(), t.MINB, sum_y != nullptr);
    const long long G = IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_PF_GRID", 0) ? n_jobs : pl.blocks;        // (A/B: one job per block)
    const size_t window_floats = (size_t)(t.M + span) * (a.C + 4);
    a.bias_off = (int)((window_floats + 3) & ~(size_t)3);
    const size_t lds_bytes = ((size_t)a.bias_off + (size_t)nz * 2 * a.C + 4) * sizeof(float);      // + the next-job word
    dim3 grid((unsigned)G, 1u, 1u), block(256);
    // The non-summing instantiations (plain persistent pairs: 3-5 % slower than one block per job at every size, profiles/r03_notes.md)
    // exist in the diagnostic build only; the release library carries the summing form alone.
#ifdef IRIS_MRF_DIAG
#define IRIS_PAIR_PF_F32_PLAIN(WT_, WC_, MT_, C_, MINB_)                                                                 \
        return ::iris::launch_kernel_named("mrf_pair_f32_pf_kernel", mrf_pair_f32_pf_kernel<WT_, WC_, MT_, C_, MINB_, false>,                 \
                                           grid, block, lds_bytes, stream, a);
#else
#define IRIS_PAIR_PF_F32_PLAIN(WT_, WC_, MT_, C_, MINB_) return hipErrorNotSupported;
#endif
#define IRIS_PAIR_PF_F32_CASE(WT_, WC_, MT_, C_, MINB_)                                                                  \
    if (a.C == C_ && t.WT == WT_ && t.MT == MT_) {                                                                       \
        if (sum_y) return ::iris::launch_kernel_named("mrf_pair_f32_pf_kernel<sum>", mrf_pair_f32_pf_kernel<WT_, WC_, MT_, C_, MINB_, true>,  \
                                                      grid, block, lds_bytes, stream, a);                                \
        IRIS_PAIR_PF_F32_PLAIN(WT_, WC_, MT_, C_, MINB_)                                                                 \
    }
    IRIS_PAIR_PF_F32_CASE(4, 1, 1, 32, 4)
    IRIS_PAIR_PF_F32_CASE(2, 2, 2, 64, 2)
#undef IRIS_PAIR_PF_F32_CASE
#undef IRIS_PAIR_PF_F32_PLAIN
    return hipErrorInvalidValue;
}

}  // namespace iris
