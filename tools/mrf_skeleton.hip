// Calibration microbenchmark (not product code): the wide fp32 MRF kernel's SKELETON, built up feature by feature.
//
// mfma_vmem_cost.hip showed that the kernel's inner instruction mix runs at 0.985 of the matrix peak as a bare loop, while the real
// kernel's steady state is 0.88 (profiles/r04_notes.md section 8).  This program adds the kernel's other ingredients one at a time to
// the same loop -- each a template flag -- and prints the rate after every step, so that the step where the 10 % goes can be read off:
//   W  weight fragments streamed from a real 1.4 MB array (an L2 hit, a new address per group) instead of one L1-resident kilobyte
//   P  phases: every NG groups a barrier, the LDS write of the staged quads (LeakyReLU on the way), a barrier, the ring hand-over
//   S  staging loads: in the first 8 groups of a phase one 16-byte load per thread from a real 32 MB tensor (written beforehand)
//   E  epilogue behind every second phase: 8 residual loads requested in its first 8 groups, bias + residual adds, 8 dwordx4 stores
//   V  the three branch lengths (88 / 56 / 24 groups per phase) instead of 56 everywhere
//   NOST / NORES  the epilogue without its stores / without its residual loads (what each costs)
//   SPREAD [UNI]  (a candidate) the stores leave one per group in groups 8 .. 15 of the NEXT phase (UNI: evenly over the whole phase), from a copy
//   FULL / RFULL  stores / residual loads addressed as 8 rows x 128 contiguous bytes per instruction instead of 32 rows x 2 x 16 bytes
//   LAT  an s_waitcnt vmcnt(0) right behind the store burst, timed with s_memtime: how long the stores take to be acknowledged
//   (-DSKEL_DB=8: weight ring of 8; -DSKEL_STORE_AUX=1 / 2 / 17: sc0 / nt / sc0 sc1 stores; every run prints the clock the chip held)
//   AHEAD (a candidate) the weight fragments of groups DB .. DB+7 of the phase BEHIND an epilogue are requested before that epilogue's
//      stores, into the residual registers (free once the residual is added): the first request behind the stores is then waited for
//      12 groups later instead of 4 (vmcnt counts loads and stores together, in issue order)
//   LATE  (a candidate) the stores leave from a copy BEHIND the phase transition (barrier, LDS write, barrier) instead of in front of it:
//      the other waves of the block then do not wait at the barrier for the slowest wave's stores to issue
//   D  (a candidate, not what the kernel does) two LDS images: the staged quad of group n is written to the OTHER image in group n + 6,
//      behind an MFMA like every other request -- no write burst between the phases, one barrier per phase instead of two
// A block = 4 waves = 64 rows x 128 channels like the kernel (C = 128: two chunks per branch, six phases per tile), persistent over
// `tiles` tiles; 1 or 2 blocks per CU.  The arithmetic is meaningless (operands are whatever the buffers hold); only the time counts.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/mrf_skeleton tools/mrf_skeleton.hip && tools/mrf_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { W = 1, P = 2, S = 4, E = 8, V = 16, D = 32, NOST = 64, NORES = 128, SPREAD = 256, FULL = 512, RFULL = 1024, LAT = 2048, UNI = 4096, AHEAD = 8192, LATE = 16384 };
#ifndef SKEL_DB
#define SKEL_DB 4
#endif
#ifndef SKEL_STORE_AUX
#define SKEL_STORE_AUX 0
#endif
constexpr int S_ROW = 68, ROWS = 114, DB = SKEL_DB, NQ = 8;

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int F, int NG, bool AH>
__device__ __forceinline__ void phase(f32x16 (&acc)[2], f32x4 (&bw)[DB + 1], f32x4 (&st)[NQ], f32x4 (&resv)[NQ], const float* aptr,
                                      __amdgpu_buffer_rsrc_t wr, unsigned wvoff, unsigned wsoff, __amdgpu_buffer_rsrc_t xr, unsigned xvoff,
                                      __amdgpu_buffer_rsrc_t rr, unsigned rvoff, bool last_chunk, float* wr_ptr, int tid,
                                      f32x4 (&outv)[NQ], __amdgpu_buffer_rsrc_t yr, unsigned out_voff) {
    f32x4 av[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(aptr + m * 32 * S_ROW);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        int n_vm = 1, n_st = 0;
        if constexpr (F & S) { if (n < NQ) { st[n] = bload(xr, xvoff + (unsigned)n * 16u * 512u, 0); ++n_vm; } }
        constexpr bool AHD = AH && (F & AHEAD);
        if constexpr ((F & E) && !(F & NORES) && !AHD) { if (n < NQ) { resv[n] = bload(rr, last_chunk ? rvoff + (unsigned)n * 8u * 512u : 0x80000000u, 0); ++n_vm; } }
        if (!AHD || n >= NQ)
            bw[(n + DB) % (DB + 1)] = bload(wr, wvoff, (F & W) ? wsoff + (unsigned)(n + DB) * 4096u : (unsigned)((n + DB) & 3) * 1024u);
        else --n_vm;
        const f32x4& frag = (AHD && n >= DB && n < DB + NQ) ? resv[n - DB] : bw[n % (DB + 1)];
        if constexpr (F & SPREAD) {     // the previous branch's stores, one per group behind the requests (groups 8 .. 15)
            constexpr int STRIDE = (F & UNI) ? (NG - NQ) / NQ : 1;      // UNI: evenly over the whole phase instead of groups 8 .. 15
            if (n >= NQ && n < NQ + NQ * STRIDE && (n - NQ) % STRIDE == 0) {
                const int idx = (n - NQ) / STRIDE;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[idx]), yr, (int)out_voff, (int)((idx / 4) * 32 * 512 + 32 * (idx % 4)), 0);
                n_st = 1;
            }
        }
        bool ds_wr = false;
        if constexpr (F & D) {
            if (n >= 6 && n < 6 + NQ) {
                const int i = n - 6;
                f32x4 v = st[i];
                v.x = fmaxf(v.x, v.x * 0.1f); v.y = fmaxf(v.y, v.y * 0.1f); v.z = fmaxf(v.z, v.z * 0.1f); v.w = fmaxf(v.w, v.w * 0.1f);
                if (tid / 16 + i * 16 < ROWS) *reinterpret_cast<f32x4*>(wr_ptr + i * 16 * S_ROW) = v;
                ds_wr = true;
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
            av[(n + 1) & 1][m] = *reinterpret_cast<const f32x4*>(aptr + ((n + 1) / 8) * S_ROW + 8 * ((n + 1) % 8) + m * 32 * S_ROW);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[e], av[n & 1][m][e], acc[m], 0, 0, 0);
        {
            int ds_left = 2, vm_left = n_vm;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
                else if (n_st > 0) { __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); --n_st; }
                else if (ds_wr && k < 7) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
                else if (ds_wr) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // ring hand-over: the next phase finds its first DB fragments in slots 0 .. DB-1
        f32x4 tmp[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d) tmp[d] = bw[(NG + d) % (DB + 1)];
#pragma unroll
        for (int d = 0; d < DB; ++d) bw[d] = tmp[d];
    }
}

template <int F>
__global__ void __launch_bounds__(256, 2) skeleton(const float* w, const float* x, const float* res, float* y, float* sink, int tiles, unsigned tbytes, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[((F & D) ? 2 : 1) * ROWS * S_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROWS * S_ROW; i += 256) lds[i] = w[(i * 7 + blockIdx.x) & 0x3ffff];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 4 << 20, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, tbytes, 0x00020000);
    const float* aptr0 = lds + (lane & 31) * S_ROW + 4 * (lane >> 5);
    float* const lds_wr = lds + (tid / 16) * S_ROW + 4 * (tid % 16);
    if constexpr (F & D) { for (int i = tid; i < ROWS * S_ROW; i += 256) lds[ROWS * S_ROW + i] = lds[i]; __syncthreads(); }
    const unsigned wvoff = (unsigned)(wave * 64 + lane) * 16u;
    f32x16 acc[2];
    f32x4 bw[DB + 1], st[NQ], resv[NQ], outv[NQ];
    bool out_pending = false; unsigned out_voff = 0x80000000u;
    for (int i = 0; i < NQ; ++i) outv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d <= DB; ++d) bw[d] = bload(wr, wvoff, (unsigned)d * 4096u);
    for (int i = 0; i < NQ; ++i) { st[i] = f32x4{0.f, 0.f, 0.f, 0.f}; resv[i] = st[i]; }
    if constexpr (F & AHEAD) { for (int j = 0; j < NQ; ++j) resv[j] = bload(wr, wvoff, (unsigned)(DB + j) * 4096u); }
    float keep = 0.f;
    unsigned long long lat_sum = 0; unsigned lat_n = 0;
    unsigned long long c0, r0;        // shader clock / constant 100 MHz clock at the start: their ratio is the clock the chip held
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    for (int t = 0; t < tiles; ++t) {
        const unsigned tile = (unsigned)(blockIdx.x + t * gridDim.x);      // 512 tiles of 64 rows x 128 channels = 32,768 rows
        const unsigned xvoff = (tile * 64u * 512u + (unsigned)(tid / 16) * 512u + (unsigned)(tid % 16) * 16u) & (tbytes - 1u);
        const unsigned fvoff = (tile * 64u * 512u + (unsigned)(lane >> 3) * 512u + (unsigned)wave * 128u + (unsigned)(lane & 7) * 16u) & (tbytes - 1u);
        const unsigned rvoff = (tile * 64u * 512u + (unsigned)(lane & 31) * 512u + (unsigned)wave * 128u + (unsigned)(lane >> 5) * 16u) & (tbytes - 1u);
#pragma unroll
        for (int ph = 0; ph < 6; ++ph) {
            const bool last_chunk = ph & 1;
            if (!(ph & 1)) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
            }
            const unsigned wsoff = (unsigned)ph * 88u * 4096u;
            const float* aptr = aptr0 + ((F & D) ? (ph & 1) * ROWS * S_ROW : 0);
            float* const wr_ptr = lds_wr + ((F & D) ? ((ph + 1) & 1) * ROWS * S_ROW : 0);
            if (ph & 1) {
                if ((F & V) && ph < 2)       phase<F, 88, false>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
                else if ((F & V) && ph >= 4) phase<F, 24, false>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
                else                         phase<F, 56, false>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
            } else {
                if ((F & V) && ph < 2)       phase<F, 88, true>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
                else if ((F & V) && ph >= 4) phase<F, 24, true>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
                else                         phase<F, 56, true>(acc, bw, st, resv, aptr, wr, wvoff, wsoff, xr, xvoff, rr, (F & RFULL) ? fvoff : rvoff, last_chunk, wr_ptr, tid, outv, yr, out_pending ? out_voff : 0x80000000u);
            }
            bool stored = false;
            if constexpr (F & SPREAD) out_pending = false;     // (the phase just run has issued them)
            if ((F & E) && last_chunk) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m][4 * g + e] = (acc[m][4 * g + e] + 0.5f) + resv[m * 4 + g][e];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F & AHEAD) {
                    const unsigned wsn = (unsigned)((ph + 1) % 6) * 88u * 4096u;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) resv[j] = bload(wr, wvoff, wsn + (unsigned)(DB + j) * 4096u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (F & NOST) {
                    keep += acc[0][0] + acc[1][5] + acc[0][9] + acc[1][14];
                } else if constexpr ((F & SPREAD) || (F & LATE)) {
#pragma unroll
                    for (int idx = 0; idx < 8; ++idx) {
                        const f32x16& src = acc[idx / 4];
                        const int g = idx % 4;
                        outv[idx] = f32x4{src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                    }
                    out_pending = true; out_voff = rvoff;
                } else {
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) {
                    const f32x16& src = acc[idx / 4];
                    const int g = idx % 4;
                    const f32x4 v = {src[4 * g + 0], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
                    if constexpr (F & FULL)    // same bytes per instruction, but 8 rows x 128 contiguous bytes instead of 32 rows x 2 x 16 bytes
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (int)fvoff, idx * 8 * 512, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (int)rvoff, (int)((idx / 4) * 32 * 512 + 32 * g), SKEL_STORE_AUX);
                }
                asm volatile("s_nop 1");
                if constexpr (F & LAT) {      // how long until the 8 stores are acknowledged (everything older is long complete)
                    unsigned long long t0, t1;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
                    lat_sum += t1 - t0; ++lat_n;
                }
                stored = true;
                }
            } else if (last_chunk) {
                keep += acc[0][0] + acc[1][5];
            }
            if constexpr ((F & P) && (F & D)) {
                if (stored) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < 16; ++q) asm volatile("" :: "v"(acc[m][q]));
                }
                __syncthreads();
            } else if constexpr (F & P) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    f32x4 v = st[i];
                    v.x = fmaxf(v.x, v.x * 0.1f); v.y = fmaxf(v.y, v.y * 0.1f); v.z = fmaxf(v.z, v.z * 0.1f); v.w = fmaxf(v.w, v.w * 0.1f);
                    if (tid / 16 + i * 16 < ROWS) *reinterpret_cast<f32x4*>(lds_wr + i * 16 * S_ROW) = v;
                }
                if (stored) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < 16; ++q) asm volatile("" :: "v"(acc[m][q]));
                }
                __syncthreads();
                if constexpr (F & LATE) {
                    if (out_pending) {
#pragma unroll
                        for (int idx = 0; idx < 8; ++idx)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[idx]), yr, (int)out_voff, (int)((idx / 4) * 32 * 512 + 32 * (idx % 4)), 0);
                        asm volatile("s_nop 1");
                        out_pending = false;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    sink[blockIdx.x * 256 + tid] = (F & LAT) ? (float)lat_sum / (float)(lat_n ? lat_n : 1) : keep + st[0][0] + resv[0][0];
}

static unsigned long long* g_clk;
static unsigned g_tbytes = 32u << 20;
template <int F>
void run(const char* what, int blocks, int tiles, const float* w, const float* x, const float* res, float* y, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 1e30f, all[9];
    for (int rep = 0; rep < 9; ++rep) {      // best of the last 8 (the first one pays the code load)
        hipEventRecord(e0);
        hipLaunchKernelGGL((skeleton<F>), dim3(blocks), dim3(256), 0, 0, w, x, res, y, sink, tiles, g_tbytes, g_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&all[rep], e0, e1);
        if (rep > 0 && all[rep] < ms) ms = all[rep];
    }
    float worst = 0.f; for (int rep = 1; rep < 9; ++rep) worst = all[rep] > worst ? all[rep] : worst;
    const double groups = (F & V) ? 2.0 * (88 + 56 + 24) : 6.0 * 56;
    const double flops = (double)blocks * 4 * tiles * groups * 8 * 4096.0;
    {
        static unsigned long long hc[2048];
        hipMemcpy(hc, g_clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += (double)hc[2 * i]; r += (double)hc[2 * i + 1]; }
        printf("    s_memtime / s_memrealtime over the blocks' lives: %.4f (x 100 MHz if s_memrealtime is the 100 MHz clock)\n", c / r);
    }
    if (F & LAT) {
        static float host[1024 * 256];
        hipMemcpy(host, sink, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost);
        double sum = 0, mx = 0; for (int i = 0; i < blocks * 256; i += 64) { sum += host[i]; mx = host[i] > mx ? host[i] : mx; }
        printf("    store burst -> vmcnt(0): mean %.0f, max-of-wave-means %.0f s_memtime ticks (100 MHz ticks: x 10 ns)\n", sum / (blocks * 4), mx);
    }
    printf("%-66s %d block(s)/CU: %8.3f ms (worst of 8: %.3f)  %6.1f TFLOP/s = %.3f of 157.3\n", what, blocks / 256, ms, worst, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

// argv[1] = "big": activations, residual and output in 1 GB buffers each, every tile its own rows (HBM misses) instead of 32 MB ones that stay in the caches
int main(int argc, char** argv) {
    float *w, *x, *res, *y, *sink;
    if (argc > 1 && argv[1][0] == 'b') g_tbytes = 1u << 30;
    printf("activation / residual / output buffers: %u MB each\n", g_tbytes >> 20);
    hipMalloc(&w, 4 << 20); hipMalloc(&x, g_tbytes); hipMalloc(&res, g_tbytes); hipMalloc(&y, g_tbytes); hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&g_clk, 1024 * 16);
    {
        static float host[1 << 20];
        unsigned s = 12345u;
        for (int i = 0; i < (1 << 20); ++i) { s = s * 1664525u + 1013904223u; host[i] = (float)(int)(s >> 8) / 8388608.0f - 1.0f; }
        hipMemcpy(w, host, sizeof(host), hipMemcpyHostToDevice);
        for (unsigned k = 0; k < g_tbytes / sizeof(host); ++k) { hipMemcpy(x + (size_t)k * (1 << 20), host, sizeof(host), hipMemcpyHostToDevice); hipMemcpy(res + (size_t)k * (1 << 20), host, sizeof(host), hipMemcpyHostToDevice); }
    }
    const int T = 48;     // tiles per block: ~7 ms launches at two blocks per CU
    for (int blocks : {512, 256, 512}) {
        run<0>("bare loop (ring 4, 2 ds_read, 1 L1-hit load per group)", blocks, T, w, x, res, y, sink);
        run<W>("W: weights streamed from L2", blocks, T, w, x, res, y, sink);
        run<W | P>("W P: + phase barriers, LDS write, ring hand-over", blocks, T, w, x, res, y, sink);
        run<W | P | S>("W P S: + staging loads", blocks, T, w, x, res, y, sink);
        run<W | P | S | E>("W P S E: + epilogue (residual loads, adds, stores)", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V>("W P S E V: + branch lengths 88 / 56 / 24", blocks, T, w, x, res, y, sink);
        run<P | S | E | V>("  P S E V (weights from L1 again)", blocks, T, w, x, res, y, sink);
        run<W | P | E | V>("W P   E V (no staging loads)", blocks, T, w, x, res, y, sink);
        run<W | P | S | V>("W P S   V (no epilogue)", blocks, T, w, x, res, y, sink);
        run<W | S | E | V>("W   S E V (no barriers / LDS write)", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | NOST>("W P S E V, epilogue without the stores", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | NORES>("W P S E V, epilogue without the residual loads", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | SPREAD>("W P S E V, stores spread over groups 8..15 of the next phase", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | SPREAD | UNI>("W P S E V, stores spread evenly over the whole next phase", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | AHEAD>("W P S E V, 8 weight fragments requested ahead of the stores", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | AHEAD | SPREAD>("W P S E V, fragments ahead AND stores spread", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | LATE>("W P S E V, stores behind the phase transition (from a copy)", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | FULL>("W P S E V, stores as 8 rows x 128 B per instruction", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | FULL | RFULL>("W P S E V, stores AND residual loads as 8 rows x 128 B", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | LAT>("W P S E V + a vmcnt(0) right behind the stores, timed", blocks, T, w, x, res, y, sink);
        run<W | P | S | D>("W P S D: two LDS images, in-loop writes, one barrier (no E, V)", blocks, T, w, x, res, y, sink);
        run<W | P | S | E | V | D>("W P S E V D: two LDS images, in-loop writes, one barrier", blocks, T, w, x, res, y, sink);
    }
    return 0;
}
