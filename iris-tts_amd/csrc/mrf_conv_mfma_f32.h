// mrf_conv_mfma_f32.h -- the hot kernel: one conv step of ALL MRF ResBlock branches of a stage.
//
// Reference semantics: ResBlock.forward, src/iris/hifigan_pretrained.py:64-71, for the three
// branches of a stage (kernel sizes 3/7/11, hifigan_pretrained.py:130-136):
//     step 2m   : xt_j = Conv1d_dil(LeakyReLU(x_j))            (convs1[m], dilation 1/3/5)
//     step 2m+1 : x_j  = Conv1d(LeakyReLU(xt_j)) + x_j          (convs2[m] + residual)
//
// Same MFMA formulation, LDS image and packed weights as conv_mfma_f32.h; what differs is the
// schedule, which is what the first profile asked for (profiles/r01*: MFMA pipe 45-74 % busy, and an
// ablation with every memory access removed still at 108-131 TFLOP/s):
//   * equal-cost blocks: a block owns one (time tile, C_out block) and runs the branches
//     k = 11, 7, 3 back to back ("phases" = branch x C_in chunk), so block cost no longer depends
//     on the kernel size and blocks live ~70 us instead of ~3-20 us;
//   * the global loads that stage phase p+1's input window are issued DURING the MFMAs of phase p,
//     one 16-byte load per MFMA group, and written to LDS after the phase (register staging,
//     "issue early / write late"); one LDS buffer, two barriers per phase;
//   * weight fragments are requested DB groups (DB * 4*MT MFMAs) ahead.  vmcnt retires in order, so a
//     wait for weight fragment n also waits for every older staging load: spreading the staging loads
//     one per group gives each of them DB groups (~DB*512 cycles) of flight time before any wave
//     can stall on it.
#pragma once
#include <stdio.h>
#include <type_traits>
#include <string.h>
#include "conv_mfma_f32.h"

namespace iris {

#ifdef IRIS_MRF_DIAG
#define IRIS_MRF_ABLATE(a) ((a).ablate)
#else
#define IRIS_MRF_ABLATE(a) 0
#endif

#ifndef IRIS_MRF_RING_MT1
#define IRIS_MRF_RING_MT1 4              // weight fragments in flight (groups ahead), half-height / full-height tiles (C >= 64)
#endif
#ifndef IRIS_MRF_RING_MT2
#define IRIS_MRF_RING_MT2 4
#endif
#ifndef IRIS_MRF_FORCE_PLAN
#define IRIS_MRF_FORCE_PLAN (-1)         // calibration builds (make relvariant EXTRA=-DIRIS_MRF_FORCE_PLAN=n): every launch in plan n
#endif
#ifndef IRIS_MRF_ZDYN_DEFAULT
#define IRIS_MRF_ZDYN_DEFAULT 1          // snake-ordered (tile, branch) jobs where the timing model favours them (A/B builds: 0)
#endif
#ifndef IRIS_MRF_ZDYN_MIN_CHUNKS
#define IRIS_MRF_ZDYN_MIN_CHUNKS 2       // ... on stages with at least this many C_in chunks (C >= 128: what the model was calibrated on)
#endif
#ifndef IRIS_ZPAR_ONE_PER_CU_DEFAULT
#define IRIS_ZPAR_ONE_PER_CU_DEFAULT 1   // (A/B builds: 0 = one-branch-per-block mode always spreads over every block slot)
#endif

#ifndef IRIS_MRF_MINWAVES
#define IRIS_MRF_MINWAVES 2      // waves per SIMD the register allocator must leave room for
#endif
#ifndef IRIS_MRF_TALL
#define IRIS_MRF_TALL 1          // wide stages: tile-serial launches with enough tiles run 128-row tiles (MT = 4, LEAN registers); A/B builds: 0
#endif

constexpr int kMrfSpanMax = 50;  // (ks-1)*dil of the widest supported conv: k=11, d=5

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer loads (T8/T20 of the CDNA guide): the 128-bit descriptor is built from kernel arguments
// and blockIdx only (provably wave-uniform -> SGPRs, no waterfall loop); every load then needs ONE
// 32-bit VGPR offset instead of a live 64-bit per-lane address, and the hardware range check
// (offset >= num_records -> 0) implements the zero padding of the convolution for free.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, (int)soff, 0);
}
constexpr unsigned kOobOffset = 0x80000000u;  // >= any num_records used here
template <class T>
__device__ __forceinline__ T* uniform_ptr_mrf(T* p) {     // a block-uniform pointer, pinned to SGPRs
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

// MINW = waves per SIMD the register allocator must leave room for (= blocks per CU: a block is one wave per SIMD).
// LEANP selects the LEAN register diet (round 4; 64-row tiles: 167 VGPRs instead of 223-240 -- which bought NOTHING as a third
// block per CU, profiles/r04_notes.md -- and what makes the 128-row tile, MT = 4, fit in 251):
//   * the biases of the three branches live in an LDS table behind the window (written once per block) and are read in
//     the epilogue, instead of 16 registers held over a whole branch;
//   * the epilogue stores straight from the accumulators (or the running MRF sum) -- nothing rewrites those registers
//     before the next branch zeroes them, behind the LDS write and two barriers -- instead of from a 32-register copy;
//   * the activation fragments are single-buffered: the MFMAs of a group are issued m-major (all four k-pairs of row tile 0,
//     then of row tile 1), and row tile m's fragment of the NEXT group is requested as soon as its four MFMAs have issued
//     (256 cycles ahead of its use).  v_mfma_f32_32x32x2_f32 chains on one accumulator back to back (SrcC forwarding).
// Every output element is the same fmaf chain in the same order as in the MINW = 2 form: bit-identical.
template <int WT, int WC, int MT, int CIC, int DB, int KA, int KB, int KC, bool SUM, int ZPAR, int MINW = IRIS_MRF_MINWAVES, bool LEANP = (MINW >= 3)>
__global__ void __launch_bounds__(256, MINW) mrf_conv_mfma_f32_kernel(const ConvLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool LEAN = LEANP;
    constexpr int S = CIC + 4;
    constexpr int QPR = CIC / 4;
    constexpr int GPC = CIC / 8;
    constexpr int T_BLK = WT * MT * 32;
    constexpr int NQ = ((T_BLK + kMrfSpanMax) * QPR + 255) / 256;  // staged 16-byte quads per thread
    constexpr int RPI = 256 / QPR;                                   // rows advanced per staged quad
    // (Two window buffers with the next phase's quads written behind the last MFMA groups -- one barrier per phase instead of
    //  barrier + LDS write + barrier -- were measured in round 3: bit-identical and no faster, the CU's other block covers that
    //  time already; removed in round 4, profiles/r03_notes.md.)
    constexpr int BUF_ROWS = T_BLK + kMrfSpanMax;
    constexpr int BUF_FLOATS = BUF_ROWS * S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WC, wc = wave - wt * WC;
    const int lo = lane & 31, hi = lane >> 5;
    const int L = a.L_in, C = a.C_in;
    const int n_chunks = (C + CIC - 1) / CIC;
    const int n_t = (L + T_BLK - 1) / T_BLK;
    const int tiles_per_item = n_t * a.n_co_blk;
    const int n_tiles = tiles_per_item * a.B;
    const unsigned wbytes_group = (IRIS_MRF_ABLATE(a) & 2) ? 0u : (unsigned)a.n_ct * 64u * 16u;  // bytes per (tap, group)
    const unsigned tap_bytes = (unsigned)a.Gp * wbytes_group;
    const float* aptr = lds + (wt * MT * 32 + lo) * S + 4 * hi;
    const float slope = a.slope;
#ifdef IRIS_MRF_DIAG
    const int ablate = a.ablate;          // diagnostic builds: runtime ablation switches
#else
    constexpr int ablate = 0;
#endif
    const unsigned tensor_bytes = (unsigned)L * (unsigned)C * 4u;
    // LEAN: [3][C_out] biases behind the window(s) and the next-tile word (launch_mrf_conv sizes the dynamic LDS for it)
    float* const lds_bias = lds + BUF_FLOATS + 4;
    if constexpr (LEAN) {
        const int q = a.C_out >> 2;                           // C_out % 4 == 0 (mrf_kernel_applicable)
        for (int i = tid; i < 3 * q; i += 256) {
            const int z = i / q, c4 = i - z * q;
            const float* bp = z == 0 ? a.p[0].bias : (z == 1 ? a.p[1].bias : a.p[2].bias);
            *reinterpret_cast<f32x4*>(lds_bias + z * a.C_out + 4 * c4) = *reinterpret_cast<const f32x4*>(bp + 4 * c4);
        }
        // (visible to every wave after the prologue's barrier; the table is never rewritten)
    }
#ifdef IRIS_MRF_BLOCKLOG
    // diagnostic build only (make variant NAME=blocklog EXTRA=-DIRIS_MRF_BLOCKLOG): every block leaves its start / end time on the
    // constant 100 MHz clock and the CU it ran on, for a per-CU timeline of the launch
    const unsigned long long blk_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef IRIS_MRF_DIAG
    // A/B (profiles/r03_notes.md): the two blocks of a CU run the same equal-cost sequence and would stay in lock-step; delay
    // the second residency generation by a.stagger x 1,024 cycles
    if (a.stagger > 0 && a.stagger_mod > 0 && (((int)blockIdx.x / a.stagger_mod) & 1))
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
#endif

    // A tile = (batch item, time tile, C_out block).  Everything a phase needs to know about it:
    struct Tile {
        size_t batch_off;   // elements to this batch item's [L, C] tensor
        int i0;             // first output row
        unsigned wvoff;     // this lane's byte offset inside a (tap, group) of packed weights
        unsigned ovoff4;    // byte offset of this lane's 16-byte output piece for (m=0, g=0): row
                            // i0 + wave rows + lo, channels co4..co4+3 (co4 = tile base + 4*hi); out of range if unused
        int co4;
    };
    auto make_tile = [&](int tile) {
        Tile t;
        const int b = tile / tiles_per_item, rem = tile - b * tiles_per_item;
        const int tile_co = rem % a.n_co_blk, tile_t = rem / a.n_co_blk;
        const int ct_raw = tile_co * WC + wc;
        const bool active = ct_raw < a.n_ct;                   // wave-uniform
        const int ct = active ? ct_raw : a.n_ct - 1;           // idle waves compute a duplicate, store nothing
        t.batch_off = (size_t)b * L * C;
        t.i0 = tile_t * T_BLK;
        t.wvoff = (unsigned)(ct * 64 + lane) * 16u;
        t.co4 = ct * 32 + 4 * hi;
        t.ovoff4 = (active && t.co4 < a.C_out)
                       ? (unsigned)((t.i0 + wt * MT * 32 + lo) * C + t.co4) * 4u : kOobOffset;
        return t;
    };

    // One staged quad = 16 bytes of row (in_row0 + r_lane + i*RPI), channels [c0+4q, c0+4q+4).
    // Its byte offset inside the batch item's tensor is vbase + i*row_stride; rows < 0 wrap to
    // >= 2^31 and rows >= L exceed num_records: both read 0.
    const int r_lane = tid / QPR, q_lane = tid - r_lane * QPR;
    const unsigned row_stride = (unsigned)(RPI * C) * 4u;
    float* const lds_wr = lds + r_lane * S + 4 * q_lane;     // + i * RPI * S for quad i
    f32x4 st[NQ];
    auto stage_vbase = [&](int in_row0, int c0) -> unsigned {
        const int ci = c0 + 4 * q_lane;
        if (ci >= C || (ablate & 1)) return kOobOffset;
        return (unsigned)((in_row0 + r_lane) * C + ci) * 4u;
    };
    auto stage_load_one = [&](int i, __amdgpu_buffer_rsrc_t xr, unsigned vbase, int R) {
        // rows past this branch's window (R < T_BLK + kMrfSpanMax for the short kernels) are not requested:
        // they would be real rows of the tensor, i.e. HBM reads that nobody uses
        st[i] = buf_load4(xr, r_lane + i * RPI < R ? vbase + (unsigned)i * row_stride : kOobOffset, 0);
    };
    auto stage_write_all = [&](int R) {      // LeakyReLU on the way in (hifigan_pretrained.py:66,68)
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (r_lane + i * RPI < R) {
                // 0 <= slope <= 1 (checked by the host): LeakyReLU(x) = max(x, slope*x), 2 VALU ops per value
                f32x4 v = st[i];
                v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
                v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
                *reinterpret_cast<f32x4*>(lds_wr + i * RPI * S) = v;
            }
    };

#ifdef IRIS_MRF_STAMPS
    // diagnostic build only: per-wave cycle totals of [0] MFMA loops, [1] epilogues, [2] barrier before
    // the LDS write, [3] LDS write, [4] barrier after it, [5] everything (kernel entry to exit)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [6] ring copy + bias/residual adds, [7] SUM + stores (parts of [1])
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    const unsigned long long t_entry = stamp();
#define IRIS_STAMP(var) const unsigned long long var = stamp()
#define IRIS_SEG(i, a_, b_) seg[i] += (b_) - (a_)
#else
#define IRIS_STAMP(var)
#define IRIS_SEG(i, a_, b_)
#endif
    f32x16 acc[MT];
    f32x16 sumv[MT];      // SUM kernels only: running MRF sum of the branch outputs of this tile
    f32x4 bw[DB + 1];     // ring of weight fragments; groups 0..DB-1 of a phase are requested by the
                          // phase before it (or by the prologue)

    // What follows a branch's last chunk: the next branch of the same tile, a branch of the block's next tile or job -- or
    // nothing (valid == false) at the very end.
    struct NextJob { bool valid; size_t batch_off; int i0; unsigned wvoff; const float* x; const f32x4* wp; int ks, dil, pad_left; };

    // One branch (problem p, KS taps) of tile `t`: all C_in chunks, then its epilogue.  get_next() is called once, at the
    // start of the branch's LAST chunk, and says which
    // window and weights the last chunk prefetches.
    auto run_branch = [&](auto ks_tag, auto pi_tag, const Tile& t, auto get_next) {
        constexpr int KS = decltype(ks_tag)::value;
        constexpr int PI = decltype(pi_tag)::value;   // index of this branch's problem in a.p[]
        constexpr int NG = KS * GPC;
        const ConvProblem& p = a.p[PI];               // constant index: stays in the kernarg segment
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp, (unsigned)(KS * a.Gp) * wbytes_group);
        const int dilS = p.dil * S;
        // Output / residual piece (m, g) of this lane sits at ovoff4 + m*32 rows + 8g channels: that part
        // goes into the scalar offset, rows >= L fall outside num_records (store dropped, load 0), lanes
        // with channels >= C_out get an out-of-range ovoff4.  A branch without residual uses a
        // zero-length descriptor, whose loads return 0.
        const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + t.batch_off, tensor_bytes);
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res + t.batch_off : p.y, (p.res && !(ablate & 8)) ? tensor_bytes : 0u);
        // loaded here, not in the epilogue: vmcnt retires in order, so a load issued in the epilogue
        // would have to wait for every prefetch issued by the last MFMA groups
        f32x4 bias4[4];                                  // channels co4 + 8g + {0..3}
        if constexpr (!LEAN) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias4[g] = *reinterpret_cast<const f32x4*>(p.bias + (t.co4 + 8 * g < a.C_out ? t.co4 + 8 * g : 0));
        }
        constexpr int NRES = MT * 4;                     // 16-byte residual pieces per lane: (m, g)
        constexpr int RPG = (NRES + NG - 1) / NG;        // residual loads issued per MFMA group
        f32x4 resv[NRES];
        auto res_load = [&](int idx, unsigned voff) {    // piece (m, g): row m*32 + lo, channels co4 + 8g..+3
            const int m = idx / 4, g = idx % 4;
            resv[idx] = buf_load4(rr, voff, (unsigned)(m * 32 * C + 8 * g) * 4u);
        };
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool last = chunk + 1 == n_chunks;
            NextJob nj;
            nj.valid = false; nj.batch_off = t.batch_off; nj.i0 = t.i0; nj.wvoff = t.wvoff;
            nj.x = p.x; nj.wp = p.wp; nj.ks = KS; nj.dil = p.dil; nj.pad_left = p.pad_left;
            if (last) nj = get_next();
            const bool next_valid = nj.valid;
            const bool has_next = !last || next_valid;
            f32x4 outv[LEAN ? 1 : MT * 4];        // store data of this branch's epilogue (see keep-alive below); LEAN: stores read acc / sumv
            bool stored = false;
            // the phase that follows: next chunk of this branch, or chunk 0 of the next job
            const bool cross = last && next_valid;
            const float* xq = cross ? nj.x + nj.batch_off : p.x + t.batch_off;
            const f32x4* wq = cross ? nj.wp : p.wp;
            const int ksq = cross ? nj.ks : KS;
            const int dilq = cross ? nj.dil : p.dil;
            const int padq = cross ? nj.pad_left : p.pad_left;
            const int i0q = cross ? nj.i0 : t.i0;
            const __amdgpu_buffer_rsrc_t xrn = make_rsrc(xq, tensor_bytes);
            const __amdgpu_buffer_rsrc_t wrn = make_rsrc(wq, (unsigned)(ksq * a.Gp) * wbytes_group);
            const int Rn = T_BLK + (ksq - 1) * dilq;
            const unsigned vbn = stage_vbase(i0q - padq, last ? 0 : (chunk + 1) * CIC);
            const unsigned wsoff0 = (unsigned)(chunk * GPC) * wbytes_group;
            const unsigned wsoffn = last ? 0u : (unsigned)((chunk + 1) * GPC) * wbytes_group;
            const unsigned wvoffn = cross ? nj.wvoff : t.wvoff;

            auto a_ptr = [&](int n) { return aptr + (n / GPC) * dilS + 8 * (n % GPC); };
            auto b_load = [&](int n, unsigned voff_next) {   // group n of this phase, or group n-NG of the next one
                if (n < NG)
                    return buf_load4(wr, t.wvoff, wsoff0 + (unsigned)(n / GPC) * tap_bytes + (unsigned)(n % GPC) * wbytes_group);
                return buf_load4(wrn, voff_next, wsoffn + (unsigned)(n - NG) * wbytes_group);   // n-NG < DB <= GPC: tap 0
            };
            f32x4 av[LEAN ? 1 : 2][MT];
            IRIS_STAMP(ts0);
#pragma unroll
            for (int m = 0; m < MT; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(a_ptr(0) + m * 32 * S);
            // Per group: 4*MT MFMAs of group n, and -- independent of them -- the requests for later
            // groups: MT LDS reads (A of n+1), one weight fragment (n+DB), one staging quad, RPG
            // residual words.  All of them are issued unconditionally (a request that is not needed
            // carries an out-of-range offset and returns 0 without touching memory), so the group is
            // one basic block and sched_group_barrier can slot ONE request behind each MFMA: every
            // request then issues in the 64-cycle shadow of an MFMA instead of piling up at the
            // group boundary.
            const unsigned vbn_eff = has_next ? vbn : kOobOffset;
            const unsigned res_voff = last ? t.ovoff4 : kOobOffset;
            const unsigned wvoffn_eff = has_next ? wvoffn : kOobOffset;
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                if (!(ablate & 16)) {
                    if (n < NQ) stage_load_one(n, xrn, vbn_eff, Rn);
#pragma unroll
                    for (int j = 0; j < RPG; ++j)
                        if (n * RPG + j < NRES) res_load(n * RPG + j, res_voff);
                }
                if (!(ablate & 32)) bw[(n + DB) % (DB + 1)] = b_load(n + DB, wvoffn_eff);
                if constexpr (LEAN) {
                    // m-major: row tile m's four MFMAs, then its fragment of group n + 1 into the same registers
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[n % (DB + 1)][e], av[0][m][e], acc[m], 0, 0, 0);
                        if (n + 1 < NG) av[0][m] = *reinterpret_cast<const f32x4*>(a_ptr(n + 1) + m * 32 * S);
                    }
                    // schedule: per row tile 4 MFMAs with one memory request behind each of the first ones, then the DS read
                    const int n_vm = 1 + (n < NQ ? 1 : 0) + ((n * RPG < NRES) ? ((NRES - n * RPG) < RPG ? (NRES - n * RPG) : RPG) : 0);
                    int vm_left = n_vm;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
                        }
                        if (n + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                if (n + 1 < NG) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        av[(n + 1) & 1][m] = *reinterpret_cast<const f32x4*>(a_ptr(n + 1) + m * 32 * S);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[n % (DB + 1)][e], av[n & 1][m][e], acc[m], 0, 0, 0);
                // schedule: MFMA, request, MFMA, request, ...  (masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read)
                {
                    constexpr int n_ds = MT;
                    const int n_vm = 1 + (n < NQ ? 1 : 0) + ((n * RPG < NRES) ? ((NRES - n * RPG) < RPG ? (NRES - n * RPG) : RPG) : 0);
                    int ds_left = (n + 1 < NG) ? n_ds : 0, vm_left = n_vm;
#pragma unroll
                    for (int k = 0; k < 4 * MT; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                        else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = NG; i < NQ; ++i) stage_load_one(i, xrn, vbn_eff, Rn);
            IRIS_STAMP(ts1);
            IRIS_SEG(0, ts0, ts1);
            // the next phase expects its groups 0..DB-1 in ring slots 0..DB-1: they were loaded into
            // slots (NG + d) % (DB+1)
            {
                f32x4 tmp[DB];
#pragma unroll
                for (int d = 0; d < DB; ++d) tmp[d] = bw[(NG + d) % (DB + 1)];
#pragma unroll
                for (int d = 0; d < DB; ++d) bw[d] = tmp[d];
            }

            if (last) {
                // Epilogue of this branch.  The MFMA was issued as D = W_frag x X_frag, i.e. D[co][t]: a
                // lane holds ONE time step (col = lane&31) and, in registers 4g..4g+3, the 4 consecutive
                // channels 8g + 4*(lane>>5) + {0..3} -- one 16-byte piece of the channels-last row, so the
                // 16-byte stores need no cross-lane transpose (the store tail is issue-bound: 8 dwordx4
                // stores per branch instead of 32 dword stores).
                // Bias and residual are added IN PLACE and all stores are then issued straight from
                // registers that nothing rewrites before the next branch starts.  Storing from
                // short-lived temporaries is NOT safe here: hipcc reuses a buffer_store_dwordx4's data
                // registers two instructions later, and with the store path busy (8 stores per wave, two
                // blocks per CU) stores were observed to pick up the NEXT piece's values in lanes 12-15
                // of each 16-lane row -- wrong results only when two blocks shared a CU.
                if constexpr (LEAN) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bias4[g] = *reinterpret_cast<const f32x4*>(lds_bias + PI * a.C_out + (t.co4 + 8 * g < a.C_out ? t.co4 + 8 * g : 0));
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[m][4 * g + e] = (acc[m][4 * g + e] + bias4[g][e]) + resv[m * 4 + g][e];
#ifdef IRIS_MRF_STAMPS
                const unsigned long long ts1b = stamp();
                seg[6] += ts1b - ts1;
#endif
                if constexpr (SUM) {
                    // Last conv step of the stage: the block has all branch outputs of its tile, so the MRF
                    // sum and the division by num_kernels (hifigan_pretrained.py:131-137) happen here, in
                    // the reference's order -- the host passes the branches reversed, so the first branch
                    // processed (PI == 2) is resblock 0: xs = rb0; xs += rb1; xs += rb2; x = xs / 3.
                    // Only the mean is stored (into a.sum_y); the per-branch outputs are never written.
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if constexpr (PI == 2) sumv[m][r] = acc[m][r];
                            else                   sumv[m][r] = sumv[m][r] + acc[m][r];
                            if constexpr (PI == 0) sumv[m][r] = sumv[m][r] / a.sum_div;
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (LEAN && (!SUM || PI == 0)) {
                    // straight from the accumulators / the running sum: these registers are next written by the zero-init of
                    // the following branch, i.e. behind the LDS write and both barriers below (keep-alive there)
                    const __amdgpu_buffer_rsrc_t yo = SUM ? make_rsrc(a.sum_y + t.batch_off, tensor_bytes) : yr;
#pragma unroll
                    for (int idx = 0; idx < MT * 4; ++idx) {
                        const f32x16& src = SUM ? sumv[idx / 4] : acc[idx / 4];
                        const int g = idx % 4;
                        const f32x4 v = {src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                        if (!(ablate & 4) || v.x == 1.2345e-30f)
                            buf_store4(v, yo, t.ovoff4, (unsigned)((idx / 4) * 32 * C + 8 * g) * 4u);
                    }
                    asm volatile("s_nop 1");
                    stored = true;
                } else if constexpr (!SUM || PI == 0) {
                    const __amdgpu_buffer_rsrc_t yo = SUM ? make_rsrc(a.sum_y + t.batch_off, tensor_bytes) : yr;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& src = SUM ? sumv[m] : acc[m];
                            outv[m * 4 + g] = f32x4{src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int idx = 0; idx < MT * 4; ++idx)
                        if (!(ablate & 4) || outv[idx].x == 1.2345e-30f)
                            buf_store4(outv[idx], yo, t.ovoff4, (unsigned)((idx / 4) * 32 * C + 8 * (idx % 4)) * 4u);
                    // explicit wait states behind the store group as well: LLVM's store-data hazard handling skips MUBUF
                    // stores wider than 64 bits whose soffset is an SGPR -- exactly this form (checked in the ISA: the data
                    // registers of the group are next written only by the following branch, see the keep-alive below)
                    asm volatile("s_nop 1");
                    stored = true;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            IRIS_STAMP(ts2);
            IRIS_SEG(1, ts1, ts2);
#ifdef IRIS_MRF_STAMPS
            if (last) seg[7] += ts2 - ts1;
#endif
            if (has_next) {
                __syncthreads();          // every wave is done reading this chunk's window
                IRIS_STAMP(ts3);
                stage_write_all(Rn);
                IRIS_STAMP(ts4);
                if (stored) {
                    // keep the epilogue's store-data registers allocated until here: nothing may be
                    // written into them right behind the buffer_store_dwordx4s that read them
                    if constexpr (LEAN) {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {      // (one register per statement: a 64-byte "v" operand is not valid in the host pass)
                                if constexpr (SUM) asm volatile("" :: "v"(sumv[m][r])); else asm volatile("" :: "v"(acc[m][r]));
                            }
                    } else {
#pragma unroll
                        for (int idx = 0; idx < MT * 4; ++idx) asm volatile("" :: "v"(outv[idx]));
                    }
                }
                __syncthreads();
                IRIS_STAMP(ts5);
                IRIS_SEG(2, ts2, ts3); IRIS_SEG(3, ts3, ts4); IRIS_SEG(4, ts4, ts5);
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // prologue: the first window and the first weight fragments of problem PI (the only latencies a
    // block exposes)
    auto prologue = [&](const float* x0, const f32x4* wp0, int ks0, int dil0, int pad0, const Tile& t) {
        const int R0 = T_BLK + (ks0 - 1) * dil0;
        const __amdgpu_buffer_rsrc_t xr0 = make_rsrc(x0 + t.batch_off, tensor_bytes);
        const __amdgpu_buffer_rsrc_t wr0 = make_rsrc(wp0, (unsigned)(ks0 * a.Gp) * wbytes_group);
        const unsigned vb0 = stage_vbase(t.i0 - pad0, 0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) stage_load_one(i, xr0, vb0, R0);
#pragma unroll
        for (int d = 0; d < DB; ++d) bw[d] = buf_load4(wr0, t.wvoff, (unsigned)d * wbytes_group);
        stage_write_all(R0);
        __syncthreads();
    };
    // the next job when it is problem a.p[PN] (constant index: the kernel arguments stay in SGPRs) of tile tn
    auto next_of = [&](auto pn_tag, bool valid, const Tile& tn) {
        constexpr int PN = decltype(pn_tag)::value;
        // (pinned to SGPRs here: a select between fields of the kernel-argument struct is otherwise turned into a load from a
        //  selected ADDRESS, i.e. a table in private memory)
        return NextJob{valid, tn.batch_off, tn.i0, tn.wvoff, uniform_ptr_mrf(a.p[PN].x), uniform_ptr_mrf(a.p[PN].wp),
                       __builtin_amdgcn_readfirstlane(a.p[PN].ks), __builtin_amdgcn_readfirstlane(a.p[PN].dil),
                       __builtin_amdgcn_readfirstlane(a.p[PN].pad_left)};
    };
    unsigned* const next_slot = reinterpret_cast<unsigned*>(lds + BUF_FLOATS);   // LDS word: a drawn tile / job index
    if constexpr (ZPAR == 0) {
        // equal-cost blocks: every block runs all three branches of its tiles, heaviest first:
        // p[2] (KC taps), p[1] (KB), p[0] (KA); persistent over tiles
        int tile = blockIdx.x;
        if (tile >= n_tiles) return;
        Tile t = make_tile(tile);
        prologue(a.p[2].x, a.p[2].wp, KC, a.p[2].dil, a.p[2].pad_left, t);        // (the host orders the branches: p[2] has KC taps)
        // dynamic mode: the next tile index comes from a global counter (first tile = blockIdx.x).  Thread 0 fetches
        // it at the start of a tile and leaves it in an LDS word behind the window; everybody reads it after the
        // barriers that end the first branch -- long before the last branch needs it for its prefetch.
        for (;;) {
            if (a.dyn_counter && tid == 0) *next_slot = gridDim.x + atomicAdd(a.dyn_counter, 1u);
            run_branch(std::integral_constant<int, KC>{}, I2{}, t, [&] { return next_of(I1{}, true, t); });
            const int tile_next = a.dyn_counter ? (int)*next_slot : tile + (int)gridDim.x;
            const bool more = (unsigned)tile_next < (unsigned)n_tiles;
            const Tile tn = make_tile(more ? tile_next : tile);
            run_branch(std::integral_constant<int, KB>{}, I1{}, t, [&] { return next_of(I0{}, true, t); });
            run_branch(std::integral_constant<int, KA>{}, I0{}, t, [&] { return next_of(I2{}, more, tn); });
            if (!more) break;
            tile = tile_next;
            t = tn;
        }
    } else if constexpr (ZPAR == 1) {
        // One branch per block: a stage step is spread over 3x the jobs, which fills the chip when there are fewer
        // tiles than block slots and softens the round quantisation of the tile-serial mode in between.  Each
        // branch has its own range of blocks, sized by the host so that the three finish together (mrf_plan):
        // [0, zb1) branch 2 (k = 11, first in dispatch order), [zb1, zb2) branch 1, [zb2, gridDim.x) branch 0;
        // a block walks the tiles  first, first + (blocks of its branch), ...
        const int bx = (int)blockIdx.x;
        const int zb = bx < a.zb1 ? 2 : (bx < a.zb2 ? 1 : 0);
        const int step = zb == 2 ? a.zb1 : (zb == 1 ? a.zb2 - a.zb1 : (int)gridDim.x - a.zb2);
        int tile = zb == 2 ? bx : (zb == 1 ? bx - a.zb1 : bx - a.zb2);
        if (tile >= n_tiles) return;
        Tile t = make_tile(tile);
        auto walk = [&](auto ks_tag, auto pi_tag) {
            constexpr int PI = decltype(pi_tag)::value;
            prologue(a.p[PI].x, a.p[PI].wp, decltype(ks_tag)::value, a.p[PI].dil, a.p[PI].pad_left, t);
            for (;;) {
                const int tile_next = tile + step;
                const bool more = tile_next < n_tiles;
                const Tile tn = make_tile(more ? tile_next : tile);
                run_branch(ks_tag, pi_tag, t, [&] { return next_of(pi_tag, more, tn); });
                if (!more) break;
                tile = tile_next;
                t = tn;
            }
        };
        if (zb == 2)      walk(std::integral_constant<int, KC>{}, I2{});
        else if (zb == 1) walk(std::integral_constant<int, KB>{}, I1{});
        else              walk(std::integral_constant<int, KA>{}, I0{});
    } else {
        // (tile, branch) JOBS in a fixed "snake" order (round 3).  Job j = branch 2 - j / n_tiles (all k = 11 jobs first, then
        // k = 7, then k = 3), tile j % n_tiles; in round r the block takes job r * G + blockIdx.x (r even) or
        // r * G + G - 1 - blockIdx.x (r odd), G = gridDim.x: the blocks that got the long jobs of one round get the short ones of
        // the next.  The launch then takes about (all units) / G + one short job whatever its number of tiles is, instead of whole
        // rounds of three-branch tiles (700 frames at batch 1 took as long as 1000).
        // (Jobs DRAWN from a counter, longest first, were tried first: a returning atomic is waited for by the first vmcnt wait
        //  behind it -- the next weight fragment, memory operations retire in order -- so every job paid the atomic's round
        //  trip, 2-4 us on jobs of 12-50 us; profiles/r03_notes.md.)
        const int n_jobs = 3 * n_tiles;
        const int G = (int)gridDim.x, bx = (int)blockIdx.x;
        if (bx >= n_jobs) return;
        // the problem of branch z, every field pinned to an SGPR before the selection (a plain select over kernel-argument
        // structs becomes a per-lane table lookup: mrf_pair_f32_pf.h)
        auto problem_of = [&](int z, const Tile& tn, bool valid) __attribute__((always_inline)) {
            auto sel_i = [&](int v0, int v1, int v2) { int v = __builtin_amdgcn_readfirstlane(v0);
                                                       if (z == 1) v = __builtin_amdgcn_readfirstlane(v1);
                                                       if (z == 2) v = __builtin_amdgcn_readfirstlane(v2); return v; };
            auto sel_p = [&](auto p0, auto p1, auto p2) { auto v = uniform_ptr_mrf(p0);
                                                          if (z == 1) v = uniform_ptr_mrf(p1);
                                                          if (z == 2) v = uniform_ptr_mrf(p2); return v; };
            return NextJob{valid, tn.batch_off, tn.i0, tn.wvoff, sel_p(a.p[0].x, a.p[1].x, a.p[2].x), sel_p(a.p[0].wp, a.p[1].wp, a.p[2].wp),
                           sel_i(a.p[0].ks, a.p[1].ks, a.p[2].ks), sel_i(a.p[0].dil, a.p[1].dil, a.p[2].dil),
                           sel_i(a.p[0].pad_left, a.p[1].pad_left, a.p[2].pad_left)};
        };
        int round = 0;
        int z = 2 - bx / n_tiles;
        Tile t = make_tile(bx - (2 - z) * n_tiles);
        {
            const NextJob first = problem_of(z, t, true);
            prologue(first.x, first.wp, first.ks, first.dil, first.pad_left, t);
        }
        for (;;) {
            ++round;
            const int job_next = round * G + ((round & 1) ? G - 1 - bx : bx);
            const bool more = job_next < n_jobs;
            const int zn = more ? 2 - job_next / n_tiles : z;
            const Tile tn = make_tile(more ? job_next - (2 - zn) * n_tiles : 0);
            const NextJob nxt = problem_of(zn, tn, more);
            auto get_next = [&]() { return nxt; };
            if (z == 2)      run_branch(std::integral_constant<int, KC>{}, I2{}, t, get_next);
            else if (z == 1) run_branch(std::integral_constant<int, KB>{}, I1{}, t, get_next);
            else             run_branch(std::integral_constant<int, KA>{}, I0{}, t, get_next);
            if (!more) break;
            z = zn;
            t = tn;
        }
    }
#ifdef IRIS_MRF_BLOCKLOG
    if (tid == 0 && a.dbg) {
        unsigned long long* rec = a.dbg + 4 * (size_t)blockIdx.x;
        rec[0] = blk_t0;
        rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID: cu / sh / se
        rec[3] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
    }
#endif
#ifdef IRIS_MRF_STAMPS
    seg[5] = stamp() - t_entry;
    if (lane == 0 && a.dbg) {
        for (int i = 0; i < 6; ++i) atomicAdd(a.dbg + i, seg[i]);
        atomicAdd(a.dbg + 6, 1ull);
        atomicAdd(a.dbg + 8, seg[6]); atomicAdd(a.dbg + 9, seg[7]);
    }
#endif
}

#ifndef IRIS_KERNELS_ONLY     // (register-count probes instantiate single kernels: tools/kernel_probe.sh)
// True when the grouped launch `a` (nz problems) can take the MRF kernel.
inline bool mrf_kernel_applicable(const ConvLaunch& a, int nz) {
    if (a.z_is_phase || a.x_channels_first || a.in_act == IN_ACT_MRF_LRELU) return false;
    if (a.C_in != a.C_out || (a.C_in & 3) || a.L_in != a.L_out || a.out_stride != 1 || a.out_off != 0) return false;
    if (a.n_idx != a.L_out) return false;
    if ((uint64_t)a.L_in * a.C_in * 4u >= 0x7fffffffull) return false;       // 32-bit buffer offsets
    if (nz != 3) return false;
    const bool fwd = a.p[0].ks == 3 && a.p[1].ks == 7 && a.p[2].ks == 11;     // the V1 MRF
    const bool rev = a.p[0].ks == 11 && a.p[1].ks == 7 && a.p[2].ks == 3;     // ... reversed, for the summing step
    if (a.sum_y ? !rev : !fwd) return false;
    if (a.in_act != IN_ACT_LRELU || !(a.slope >= 0.f && a.slope <= 1.f)) return false;
    { const ConvTile tt = pick_tile(a.C_in, a.C_out); if (tt.WT == 4 && tt.CIC != 32) return false; }
    for (int j = 0; j < nz; ++j) {
        if (packed_conv1d_floats(a.C_in, a.C_out, a.p[j].ks) * 4u >= 0x7fffffffull) return false;
        const int ks = a.p[j].ks;
        if (ks != 3 && ks != 7 && ks != 11) return false;
        if ((ks - 1) * a.p[j].dil > kMrfSpanMax) return false;
    }
    return true;
}

// How a grouped MRF step is spread over the chip.
struct MrfPlan { int MT; bool zpar; bool zdyn; bool small; long long n_tiles; long long grid; int zb1, zb2; };

// the small-problem kernel (mrf_small_f32.h): 16 x 16 jobs on v_mfma_f32_16x16x4_f32, same bits
inline bool mrf_small_applicable(const ConvLaunch& a, int nz);
inline double mrf_small_cycles(const ConvLaunch& a, int nz);
inline hipError_t launch_mrf_small(ConvLaunch& a, int nz, hipStream_t stream);

inline int mrf_cu_count() { return device_cu_count(); }

// The modes, and how one is chosen (round 3: one timing model for all of them, calibrated on forced-plan sweeps over
// 100 ... 2000 frames and batches 1 ... 32 -- profiles/r03_notes.md, tools/plan_sweep.py):
//  * tile-serial (MT = 2 or 1): a persistent grid of at most `per_cu` blocks per CU; every block runs the three branches
//    of its tiles (21 x MT tap-units each) and the grid is evened out so that every block walks the same number of tiles
//    (+-1): 1000 tiles on 256 CUs x 2 -> 2 rounds -> 500 blocks of 2 tiles.
//  * one branch per block, fixed ranges (zpar, MT = 1): branch c gets nb_c blocks which each walk ceil(tiles / nb_c) tiles;
//    the block counts minimise max_c ceil(tiles / nb_c) x cost_c under nb_11 + nb_7 + nb_3 <= slots, or <= CUs (one block
//    per CU) when the model prefers that.
//  * (tile, branch) jobs in snake order (zdyn, MT = 2 or 1; two or more C_in chunks): min(3 x tiles, slots) blocks, see the
//    kernel.  Takes about (all units) / slots + one short job whatever the number of tiles is, where the tile-serial mode pays
//    whole rounds -- 700 frames at batch 1 took as long as 1000.
//  * the small-problem kernel (mrf_small_f32.h: 16 x 16 jobs on v_mfma_f32_16x16x4_f32) when its estimate (MFMA work spread
//    over all SIMDs, bounded below by its longest chain) is clearly below the best of the modes above.
// The model: a block's chain is the tap-units it runs (x MT); blocks are dispatched in index order, so CU i hosts blocks
// i, i + CUs, ...; two blocks on a CU share its matrix pipes -- while both run each advances one unit per time unit, a block
// alone on its CU advances kLoneSpeed units (1.8: a lone wave per SIMD leaves its pipe idle between dependent MFMAs).  A CU
// takes  shorter + (longer - shorter) / kLoneSpeed,  the launch the slowest CU.  Half-height tiles cost 6 % per unit (more
// window per output row), the snake order 4 % (blocks end unevenly), fixed ranges 2 %.  Over the 66 measured (shape, stage)
// cases the model's choice is within 2.5 % of the best forced plan, 0.1 % on average.
// `force` (>= 0: the single-step test entry point, the diagnostic build's IRIS_HIFIGAN_MRFPLAN, or a calibration build's
// IRIS_MRF_FORCE_PLAN) pins the mode: 0 full-height tiles, 1 half-height tiles, 2 half-height + one branch per block,
// 3 the round-1 rule, 4 the small-problem kernel, 5 / 6 snake-ordered jobs at half / full tile height.
constexpr double kLoneSpeed = 1.8, kHalfHeightCost = 1.06, kSnakeCost = 1.04, kFixedRangeCost = 1.02;

inline MrfPlan mrf_plan_uncached(const ConvLaunch& a, bool allow_zpar, int plan_env, int per_cu) {
    const ConvTile t = pick_tile(a.C_in, a.C_out);
    const int n_cu = mrf_cu_count();
    const long long slots = (long long)n_cu * per_cu;
    const int n_co_blk = (a.C_out + t.CO_BLK - 1) / t.CO_BLK;
    auto tiles = [&](int MT) { return (long long)((a.L_out + t.WT * MT * 32 - 1) / (t.WT * MT * 32)) * n_co_blk * a.B; };
    static const int cost[3] = {3, 7, 11};            // branch 0, 1, 2
    const long long n1 = tiles(1);
    // time of the slowest CU; chain(i) = units of block i, CU c hosts blocks c, c + n_cu, ...
    auto cu_time = [&](long long blocks, auto chain) -> double {
        double worst = 0;
        double c[8];
        for (long long i = 0; i < blocks && i < n_cu; ++i) {
            int k = 0;
            for (long long j = i; j < blocks && k < 8; j += n_cu) c[k++] = chain(j);
            for (int x = 1; x < k; ++x) for (int y = x; y > 0 && c[y] < c[y - 1]; --y) { const double tmp = c[y]; c[y] = c[y - 1]; c[y - 1] = tmp; }
            double tcu = 0, prev = 0;
            for (int x = 0; x < k; ++x) {                 // k - x blocks still running
                const int m = k - x;
                tcu += (c[x] - prev) * (m == 1 ? 1.0 / kLoneSpeed : 0.5 * m);
                prev = c[x];
            }
            if (tcu > worst) worst = tcu;
        }
        return worst;
    };
    // tile-serial
    auto serial_grid = [&](long long n) { long long g = n < slots ? n : slots; if (g < 1) g = 1; const long long r = (n + g - 1) / g; return (n + r - 1) / r; };
    auto serial = [&](int MT) -> double {
        const long long n = tiles(MT), g = serial_grid(n), rounds = (n + g - 1) / g, full = n - (rounds - 1) * g;
        return cu_time(g, [&](long long i) { return (double)((i < full ? rounds : rounds - 1) * 21 * MT); }) * (MT == 1 ? kHalfHeightCost : 1.0);
    };
    // fixed ranges: smallest M with sum_c ceil(n / floor(M / cost_c)) <= limit
    auto split = [&](long long limit, long long (&out)[3]) -> double {
        double bestM = 1e30;
        for (int c = 0; c < 3; ++c)
            for (long long r = 1; r <= 4096; r = r < 64 ? r + 1 : r * 2) {
                const long long M = r * cost[c];
                if ((double)M >= bestM) break;
                long long sum = 0, cand[3];
                bool ok = true;
                for (int d = 0; d < 3; ++d) {
                    const long long rd = M / cost[d];
                    if (rd < 1) { ok = false; break; }
                    cand[d] = (n1 + rd - 1) / rd;
                    sum += cand[d];
                }
                if (ok && sum <= limit) { bestM = (double)M; out[0] = cand[0]; out[1] = cand[1]; out[2] = cand[2]; }
            }
        return bestM;
    };
    auto ranges_time = [&](const long long (&x)[3]) -> double {      // blocks in dispatch order: branch 2, 1, 0
        return cu_time(x[0] + x[1] + x[2], [&](long long i) {
            const int c = i < x[2] ? 2 : (i < x[2] + x[1] ? 1 : 0);
            return (double)(((n1 + x[c] - 1) / x[c]) * cost[c]);
        }) * kHalfHeightCost * kFixedRangeCost;
    };
    long long nb[3] = {1, 1, 1};
    double zpar_units = 1e30, zpar_chain = 1e30;
    if (allow_zpar && n1 > 0 && (zpar_chain = split(slots, nb)) < 1e29) {
        zpar_units = ranges_time(nb);
        long long nb1[3] = {1, 1, 1};
        if (nb[0] + nb[1] + nb[2] > n_cu && IRIS_DIAG_ENV("IRIS_HIFIGAN_ZPAR_ONE_PER_CU", IRIS_ZPAR_ONE_PER_CU_DEFAULT) && split(n_cu, nb1) < 1e29) {
            const double u1 = ranges_time(nb1);
            if (u1 < zpar_units) { zpar_units = u1; nb[0] = nb1[0]; nb[1] = nb1[1]; nb[2] = nb1[2]; }
        }
    }
    // snake-ordered jobs: block b runs jobs r * G + (r even ? b : G - 1 - b); N(X) = how many of them have an index below X
    auto snake = [&](int MT) -> double {
        const long long n = tiles(MT), J = 3 * n, G = J < slots ? J : slots;
        auto below = [&](long long X, long long b) {
            long long cnt = 0;
            for (int par = 0; par < 2; ++par) {
                const long long off = par ? G - 1 - b : b;
                const long long r = X > off ? (X - off + G - 1) / G : 0;       // rounds r' in [0, r) have r' * G + off < X
                cnt += par ? r / 2 : (r + 1) / 2;
            }
            return cnt;
        };
        return cu_time(G, [&](long long b) {
            const long long c11 = below(n, b), c7 = below(2 * n, b) - c11, c3 = below(J, b) - c11 - c7;
            return (double)((11 * c11 + 7 * c7 + 3 * c3) * MT);
        }) * (MT == 1 ? kHalfHeightCost : 1.0) * kSnakeCost;
    };
    MrfPlan pl;
    pl.MT = 2; pl.zpar = false; pl.zdyn = false; pl.small = false; pl.zb1 = pl.zb2 = 0;
    const bool small_ok = allow_zpar && mrf_small_applicable(a, 3);
    const bool zdyn_ok = allow_zpar && a.C_in >= IRIS_MRF_ZDYN_MIN_CHUNKS * t.CIC && n1 > 0;
    if (plan_env >= 0 && plan_env <= 2) { pl.MT = plan_env == 0 ? 2 : 1; pl.zpar = plan_env == 2 && allow_zpar && zpar_units < 1e29; }
    else if (plan_env == 5 || plan_env == 6) { pl.MT = plan_env == 5 ? 1 : 2; pl.zpar = pl.zdyn = zdyn_ok; }
    else if (plan_env == 3) {
        if (4 * tiles(2) < 3 * slots) { pl.MT = 1; pl.zpar = allow_zpar && tiles(1) < n_cu && zpar_units < 1e29; }
    } else if (plan_env == 4) {
        pl.small = small_ok;
    } else {
        double best = serial(2);
        { const double u = serial(1); if (u < 0.999 * best) { best = u; pl.MT = 1; } }
        if (zpar_units < 0.999 * best) { best = zpar_units; pl.MT = 1; pl.zpar = true; }
        if (zdyn_ok && IRIS_DIAG_ENV("IRIS_HIFIGAN_ZDYN", IRIS_MRF_ZDYN_DEFAULT)) {
            for (int MT = 2; MT >= 1; --MT) {
                const double u = snake(MT);
                if (u < 0.999 * best) { best = u; pl.MT = MT; pl.zpar = pl.zdyn = true; }
            }
        }
        // The small-problem kernel is weighed as in round 2, against the longest CHAIN of the modes above (units of one tap
        // of a 32-row x 32-channel wave tile = C/8 groups x 4 MFMAs x 64 cycles): that comparison was calibrated on
        // 40 ... 282 frames and is kept as it was.
        // (A rule that also took the kernel for the C = 256 stage up to 2,600 rows was tried and removed: it rested on numbers
        // from the diagnostic build, whose persistent kernel is ~10 % slower; in the release build it won 5 % at 282 frames
        // and lost 10 % at 200 -- profiles/r02_notes.md.)
        auto chain = [&](int MT) { return (double)((tiles(MT) + slots - 1) / slots) * 21.0 * MT * (MT == 1 ? 1.03 : 1.0); };
        double longest = chain(2);
        if (chain(1) < longest) longest = chain(1);
        if (zpar_chain * 1.03 * 1.06 < longest) longest = zpar_chain * 1.03 * 1.06;
        if (small_ok && IRIS_DIAG_ENV("IRIS_HIFIGAN_MRFSMALL", 1) &&
            mrf_small_cycles(a, 3) < 0.9 * longest * (double)a.C_in * 32.0) {
            pl.small = true; pl.zpar = pl.zdyn = false; pl.MT = 1;
        }
    }
    pl.n_tiles = tiles(pl.MT);
    if (pl.zdyn) {
        pl.grid = 3 * pl.n_tiles < slots ? 3 * pl.n_tiles : slots;
        return pl;
    }
    if (pl.zpar) {
        pl.zb1 = (int)nb[2]; pl.zb2 = (int)(nb[2] + nb[1]); pl.grid = nb[2] + nb[1] + nb[0];
        return pl;
    }
    pl.grid = serial_grid(pl.n_tiles);
    return pl;
}

// The plan of a (shape, mode) is the same in every forward: a few entries are remembered per thread.
inline MrfPlan mrf_plan(const ConvLaunch& a, bool allow_zpar, int force = -1) {
    const int per_cu_env = IRIS_DIAG_ENV("IRIS_HIFIGAN_PERCU", 0);
    const int plan_env = force >= 0 ? force : IRIS_DIAG_ENV("IRIS_HIFIGAN_MRFPLAN", IRIS_MRF_FORCE_PLAN);
    const int per_cu = per_cu_env > 0 ? per_cu_env : IRIS_MRF_MINWAVES;
#ifdef IRIS_MRF_DIAG
    return mrf_plan_uncached(a, allow_zpar, plan_env, per_cu);       // (environment switches may change between calls)
#else
    struct Entry { int key[8]; MrfPlan pl; bool used; };
    static thread_local Entry cache[16] = {};
    static thread_local int next = 0;
    const bool has16 = a.p[0].wp16 && a.p[1].wp16 && a.p[2].wp16;      // (the small-problem kernel's packing: part of its applicability)
    const int key[8] = {a.C_in, a.C_out, a.L_out, a.B, (allow_zpar ? 1 : 0) | (has16 ? 2 : 0) | (a.sum_y ? 4 : 0), plan_env, mrf_cu_count(), per_cu};
    for (const Entry& e : cache)
        if (e.used && memcmp(e.key, key, sizeof(key)) == 0) return e.pl;
    Entry& e = cache[next];
    next = (next + 1) % 16;
    memcpy(e.key, key, sizeof(key));
    e.pl = mrf_plan_uncached(a, allow_zpar, plan_env, per_cu);
    e.used = true;
    return e.pl;
#endif
}

inline hipError_t launch_mrf_conv(ConvLaunch& a, int nz, hipStream_t stream, int force_plan = -1) {
    const ConvTile t = pick_tile(a.C_in, a.C_out);
    a.n_co_blk = (a.C_out + t.CO_BLK - 1) / t.CO_BLK;
    a.Gp = packed_groups(a.C_in);
    a.n_ct = packed_cotiles(a.C_out);
    a.z_serial = 1;
    a.nz_serial = nz;
    a.nz = 1;
    a.ablate = IRIS_DIAG_ENV("IRIS_HIFIGAN_ABLATE", 0);
    a.stagger = IRIS_DIAG_ENV("IRIS_HIFIGAN_STAGGER", 0);
    a.stagger_mod = mrf_cu_count();
    const MrfPlan pl = mrf_plan(a, a.sum_y == nullptr, force_plan);
    if (pl.small) return launch_mrf_small(a, nz, stream);
    // 128-row tiles (round 4) for the tile-serial, non-summing launches of the wide stages: each weight fragment feeds four row
    // tiles instead of two and a tile has half the phase transitions per MFMA (the kernel's LEAN register form: 251 VGPRs).
    // Taken when it leaves no block slot emptier than the 64-row plan does: batch 1 x 1000 frames, C = 128: 500 blocks of one
    // 128-row tile instead of two 64-row tiles, 335 -> 328 us per step; 4 x 1000: -3 %; neutral from ~16,000 frames on.
    MrfPlan plt = pl;
    bool tall = false;
    if (IRIS_DIAG_ENV("IRIS_HIFIGAN_MRFTALL", IRIS_MRF_TALL) && a.sum_y == nullptr && t.WT == 1 && pl.MT == 2 && !pl.zpar && !pl.zdyn && !pl.small) {
        const long long n4 = (long long)((a.L_out + 127) / 128) * a.n_co_blk * a.B;
        const long long slots = 2LL * mrf_cu_count();
        long long g4 = n4 < slots ? n4 : slots; if (g4 < 1) g4 = 1;
        const long long r4 = (n4 + g4 - 1) / g4; g4 = (n4 + r4 - 1) / r4;
        if (g4 >= pl.grid) { tall = true; plt.MT = 4; plt.n_tiles = n4; plt.grid = g4; }
    }
    const MrfPlan& plx = plt;
    const int T_BLK = t.WT * plx.MT * 32;
    const bool lean = tall;
    size_t lds_bytes = (size_t)(T_BLK + kMrfSpanMax) * (t.CIC + 4) * sizeof(float) + 16;         // + next-tile word
    if (lean) lds_bytes += (size_t)3 * a.C_out * sizeof(float);                                   // + the bias table of the LEAN form
    // the counter only pays when a block walks several tiles (each fetch delays one wave by an atomic round trip)
    const int dyn_env = IRIS_DIAG_ENV("IRIS_HIFIGAN_DYNTILES", 1);
    if (!dyn_env || pl.zpar || plx.n_tiles < 4 * plx.grid) a.dyn_counter = nullptr;
    a.zb1 = pl.zb1; a.zb2 = pl.zb2;
    const long long n_tiles = plx.n_tiles, g = plx.grid;
    if (n_tiles > 0x7fffffffLL / 3) return hipErrorInvalidValue;
    dim3 grid((unsigned)g, 1u, 1u), block(256);
#define IRIS_MRF_LAUNCH_K(...)                                                                    \
    do {                                                                                          \
        auto kfn = __VA_ARGS__;                                                                   \
        { const hipError_t e__ = ::iris::launch_kernel_named(#__VA_ARGS__, kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
    } while (0)
    // (the 128-row form exists for the wide tile only: IRIS_MRF_LAUNCH_TALL refuses the others)
#define IRIS_MRF_LAUNCH_TALL(WT_, WC_, CIC_) IRIS_MRF_LAUNCH_TALL_##WT_(WC_, CIC_)
#define IRIS_MRF_LAUNCH_TALL_1(WC_, CIC_) IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<1, WC_, 4, CIC_, 2, 3, 7, 11, false, 0, 2, true>)
#define IRIS_MRF_LAUNCH_TALL_2(WC_, CIC_) return hipErrorInvalidValue
#define IRIS_MRF_LAUNCH_TALL_4(WC_, CIC_) return hipErrorInvalidValue
    // (snake-ordered jobs need two or more C_in chunks -- mrf_plan's zdyn_ok: the wide tile only; the narrow tiles are not instantiated)
#define IRIS_MRF_LAUNCH_SNAKE(WT_, WC_, MT_, CIC_, D_) IRIS_MRF_LAUNCH_SNAKE_##WT_(WC_, MT_, CIC_, D_)
#define IRIS_MRF_LAUNCH_SNAKE_1(WC_, MT_, CIC_, D_) IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<1, WC_, MT_, CIC_, D_, 3, 7, 11, false, 2>)
#define IRIS_MRF_LAUNCH_SNAKE_2(WC_, MT_, CIC_, D_) return hipErrorInvalidValue
#define IRIS_MRF_LAUNCH_SNAKE_4(WC_, MT_, CIC_, D_) return hipErrorInvalidValue
#define IRIS_MRF_LAUNCH_DB(WT_, WC_, CIC_, D1_, D2_)                                                       \
    do {                                                                                          \
        if (tall) IRIS_MRF_LAUNCH_TALL(WT_, WC_, CIC_);                                            \
        else if (pl.MT == 2) {                                                                    \
            if (pl.zdyn) IRIS_MRF_LAUNCH_SNAKE(WT_, WC_, 2, CIC_, D2_);                           \
            else if (a.sum_y) IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<WT_, WC_, 2, CIC_, D2_, 11, 7, 3, true, 0>);   \
            else         IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<WT_, WC_, 2, CIC_, D2_, 3, 7, 11, false, 0>);  \
        } else if (a.sum_y)  IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<WT_, WC_, 1, CIC_, D1_, 11, 7, 3, true, 0>);   \
        else if (pl.zdyn)    IRIS_MRF_LAUNCH_SNAKE(WT_, WC_, 1, CIC_, D1_);                       \
        else if (pl.zpar)    IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<WT_, WC_, 1, CIC_, D1_, 3, 7, 11, false, 1>);   \
        else                 IRIS_MRF_LAUNCH_K(mrf_conv_mfma_f32_kernel<WT_, WC_, 1, CIC_, D1_, 3, 7, 11, false, 0>);  \
    } while (0)
#ifdef IRIS_MRF_BLOCKLOG
    static unsigned long long* blk_dev = nullptr;
    if (!blk_dev) { if (hipMalloc(&blk_dev, 4096 * 4 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory; }
    (void)hipMemsetAsync(blk_dev, 0, 4096 * 4 * sizeof(unsigned long long), stream);
    a.dbg = g <= 4096 ? blk_dev : nullptr;
#endif
#ifdef IRIS_MRF_STAMPS
    static unsigned long long* dbg_dev = nullptr;
    if (!dbg_dev) { if (hipMalloc(&dbg_dev, 16 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory; }
    (void)hipMemsetAsync(dbg_dev, 0, 16 * sizeof(unsigned long long), stream);
    a.dbg = dbg_dev;
#endif
    if (t.WT == 4)          IRIS_MRF_LAUNCH_DB(4, 1, 32, 4, 4);     // C <= 32 (mrf_kernel_applicable: CIC == 32 there)
    else if (t.WT == 2)     IRIS_MRF_LAUNCH_DB(2, 2, 64, IRIS_MRF_RING_MT1, 4);
    else                    IRIS_MRF_LAUNCH_DB(1, 4, 64, IRIS_MRF_RING_MT1, IRIS_MRF_RING_MT2);
#ifdef IRIS_MRF_STAMPS
    {   // diagnostic build: synchronous read-back of the per-wave cycle shares
        unsigned long long h[16];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
        const double tot = (double)h[5], nw = (double)h[6];
        double mfma_cyc = 0;   // ideal MFMA cycles per wave: 64 per MFMA
        for (int j = 0; j < nz; ++j) mfma_cyc += 2.0 * a.p[j].ks * (a.C_in / 2.0);
        mfma_cyc *= 64.0 * (double)n_tiles / (double)g;
        fprintf(stderr, "[stamps] C=%d L=%d grid=%lld waves=%.0f cyc/wave=%.0f ideal_mfma=%.0f (%.3f) | mfma_loop %.3f epilogue %.3f bar1 %.3f ldswrite %.3f bar2 %.3f other %.3f\n",
                a.C_in, a.L_in, g, nw, tot / nw, mfma_cyc, mfma_cyc / (tot / nw), h[0] / tot, h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot,
                1.0 - (h[0] + h[1] + h[2] + h[3] + h[4]) / tot);
        fprintf(stderr, "[stamps]   of the epilogue: adds (incl. wait for residual) %.3f, whole last-chunk epilogue %.3f\n", h[8] / tot, h[9] / tot);
    }
#endif
#ifdef IRIS_MRF_BLOCKLOG
    if (a.dbg && IRIS_DIAG_ENV("IRIS_HIFIGAN_BLOCKLOG", 0)) {   // synchronous read-back: one line per launch + per-CU detail of the slowest CUs
        static unsigned long long hrec[4096 * 4];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(hrec, blk_dev, (size_t)g * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0; double busy = 0; int n = 0;
        static int per_cu[8 * 64 * 16]; memset(per_cu, 0, sizeof(per_cu));
        static unsigned long long cu_end[8 * 64 * 16]; memset(cu_end, 0, sizeof(cu_end));
        for (long long i = 0; i < g; ++i) {
            const unsigned long long t0 = hrec[4 * i], t1 = hrec[4 * i + 1];
            if (!t1) continue;
            const unsigned hw = (unsigned)hrec[4 * i + 2], xcc = (unsigned)hrec[4 * i + 3] & 15u;
            const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            const unsigned key = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
            if (key < sizeof(per_cu) / sizeof(per_cu[0])) { per_cu[key]++; if (t1 > cu_end[key]) cu_end[key] = t1; }
            if (t0 < tmin) tmin = t0;
            if (t1 > tmax) tmax = t1;
            busy += (double)(t1 - t0); ++n;
        }
        int cus = 0, hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double end_sum = 0, end_min = 1e30;
        for (size_t k = 0; k < sizeof(per_cu) / sizeof(per_cu[0]); ++k)
            if (per_cu[k]) { ++cus; hist[per_cu[k] < 7 ? per_cu[k] : 7]++; const double e = (double)(cu_end[k] - tmin); end_sum += e; if (e < end_min) end_min = e; }
        const double span = (double)(tmax - tmin);
        fprintf(stderr, "[blocklog] C=%d L=%d grid=%lld blocks=%d CUs=%d blocks/CU hist 1:%d 2:%d 3:%d 4:%d 5+:%d | span %.1f us, mean block %.1f us, "
                        "block-residency %.3f of 2 per CU over the span | CU finish: earliest %.3f mean %.3f of the span\n",
                a.C_in, a.L_in, g, n, cus, hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], span / 100.0, busy / n / 100.0,
                busy / (span * cus), end_min / span, end_sum / cus / span);
    }
#endif
#undef IRIS_MRF_LAUNCH_DB
#undef IRIS_MRF_LAUNCH_SNAKE
#undef IRIS_MRF_LAUNCH_SNAKE_1
#undef IRIS_MRF_LAUNCH_SNAKE_2
#undef IRIS_MRF_LAUNCH_SNAKE_4
#undef IRIS_MRF_LAUNCH_TALL
#undef IRIS_MRF_LAUNCH_TALL_1
#undef IRIS_MRF_LAUNCH_TALL_2
#undef IRIS_MRF_LAUNCH_TALL_4
#undef IRIS_MRF_LAUNCH_K
    return hipSuccess;       // (every launch above has reported its own status)
}
#endif  // IRIS_KERNELS_ONLY

}  // namespace iris
