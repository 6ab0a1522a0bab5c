// iris_hifigan.hip -- C-ABI (include/iris_hifigan.h) over the gfx950 kernels.
//
// Generator dataflow follows HiFiGANModel.forward (reference src/iris/hifigan_pretrained.py:123-143,
// Keras twin src/iris/vocoder.py:103-130):
//   conv_pre -> for each stage { LeakyReLU -> ConvTranspose1d -> MRF(3 ResBlocks) / 3 } -> LeakyReLU
//   -> conv_post -> tanh.
// Launch plan for one forward (V1 config: 30 launches):
//   1            conv_pre, reading the channels-first mel directly
//   per stage:   1 upsample launch (all u phases as blockIdx.z; input = LeakyReLU of conv_pre, or
//                LeakyReLU(mean of the previous stage's branch outputs) fused into the LDS staging)
//                2*num_dilations grouped launches: launch s runs conv (s even: convs1[s/2], dilated;
//                s odd: convs2[s/2] + residual) of ALL MRF branches at once (blockIdx.z = branch)
//   1            conv_post + tanh, reading the mean of the last stage's branch outputs.
// The MRF sum and the division by num_kernels (hifigan_pretrained.py:131-137) are never
// materialised: the consumer of a stage reads the branch outputs and forms ((b0+b1)+b2)/3 itself.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <new>
#include <utility>
#include <vector>

#include "generator_internal.h"
#include "host_parallel.h"
#include "conv_mfma_f32.h"
#include "mrf_conv_mfma_f32.h"
#include "convt_mfma_f32.h"
#include "mrf_small_f32.h"
#include "mrf_pair_f32.h"
#include "mrf_pair_f32_pf.h"
#include "conv_post.h"
#include "postnet.h"

using namespace iris;

namespace iris {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace iris

namespace {

int validate(const iris_hifigan_config* c) {
    if (!c) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "config is NULL");
    if (c->in_channels < 1 || c->upsample_initial_channel < 1)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "channel counts must be positive");
    if (c->num_upsamples < 1 || c->num_upsamples > IRIS_HIFIGAN_MAX_STAGES)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "num_upsamples %d out of range", c->num_upsamples);
    if (c->num_kernels < 1 || c->num_kernels > IRIS_HIFIGAN_MAX_KERNELS)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "num_kernels %d out of range", c->num_kernels);
    if ((c->upsample_initial_channel >> c->num_upsamples) < 1)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "upsample_initial_channel too small for %d stages",
                    c->num_upsamples);
    for (int i = 0; i < c->num_upsamples; ++i) {
        const int u = c->upsample_rates[i], k = c->upsample_kernel_sizes[i];
        if (u < 1 || k < u || ((k - u) & 1))
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT,
                        "stage %d: need kernel >= rate and (kernel - rate) even, got k=%d u=%d", i, k, u);
        if (u > 65535) return fail(IRIS_HIFIGAN_UNSUPPORTED, "upsample rate too large");
    }
    for (int j = 0; j < c->num_kernels; ++j) {
        const int k = c->resblock_kernel_sizes[j];
        if (k < 1 || !(k & 1))
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "resblock kernel size %d must be odd", k);
        if (c->num_dilations[j] < 1 || c->num_dilations[j] > IRIS_HIFIGAN_MAX_DILATIONS)
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "num_dilations[%d] out of range", j);
        if (c->num_dilations[j] != c->num_dilations[0])
            return fail(IRIS_HIFIGAN_UNSUPPORTED, "MRF branches must have the same number of dilations");
        for (int m = 0; m < c->num_dilations[j]; ++m)
            if (c->resblock_dilations[j][m] < 1)
                return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "dilations must be >= 1");
    }
    if (c->pre_kernel_size < 1 || !(c->pre_kernel_size & 1) || c->post_kernel_size < 1 ||
        !(c->post_kernel_size & 1))
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "pre/post kernel sizes must be odd");
    return IRIS_HIFIGAN_OK;
}

void build_layers(iris_hifigan_handle* h) {
    const iris_hifigan_config& c = h->cfg;
    h->pre.kind = 0;
    h->pre.C_in = c.in_channels; h->pre.C_out = c.upsample_initial_channel; h->pre.k = c.pre_kernel_size;
    h->stages.resize(c.num_upsamples);
    h->hop = 1;
    int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.num_upsamples; ++i) {
        Stage& st = h->stages[i];
        st.up.kind = 1;
        st.up.C_in = ch; st.up.C_out = ch / 2; st.up.k = c.upsample_kernel_sizes[i];
        st.up.u = c.upsample_rates[i];
        st.rate = st.up.u;
        ch /= 2;
        st.C = ch;
        h->hop *= st.rate;
        st.c1.resize(c.num_kernels); st.c2.resize(c.num_kernels);
        for (int j = 0; j < c.num_kernels; ++j) {
            for (int m = 0; m < c.num_dilations[j]; ++m) {
                ConvLayer l1; l1.C_in = ch; l1.C_out = ch; l1.k = c.resblock_kernel_sizes[j];
                l1.dil = c.resblock_dilations[j][m];
                ConvLayer l2 = l1; l2.dil = 1;
                st.c1[j].push_back(l1); st.c2[j].push_back(l2);
            }
        }
    }
    h->post.kind = 2;
    h->post.C_in = ch; h->post.C_out = 1; h->post.k = c.post_kernel_size;
    // device blob layout
    size_t off = 0;
    for_each_layer(h, [&](ConvLayer& l) {
        l.ref_w_floats = (size_t)l.C_in * l.C_out * l.k;
        if (l.kind == 2)      l.w_floats = (size_t)l.k * l.C_in;
        else if (l.kind == 1) l.w_floats = packed_convt_phase_floats(l.C_in, l.C_out, l.k, l.u) * l.u;
        else                  l.w_floats = packed_conv1d_floats(l.C_in, l.C_out, l.k);
    });
    for_each_layer(h, [&](ConvLayer& l) {
        l.w_off = off; off += (l.w_floats + 3) & ~(size_t)3;
        l.b_off = off; off += ((size_t)l.C_out + 3) & ~(size_t)3;
    });
    h->blob_floats = off;
}

uint64_t ref_weight_count(iris_hifigan_handle* h) {
    uint64_t n = 0;
    for_each_layer(h, [&](ConvLayer& l) { n += l.ref_w_floats + l.C_out; });
    return n;
}

// ---- workspace layout (floats per mel frame, times B*T) ----
struct WsLayout {
    size_t pre;   // conv_pre output                [B, T, C0]
    size_t up;    // upsample output of a stage     [B, L, C]     (max over stages)
    size_t y[IRIS_HIFIGAN_MAX_KERNELS];   // running x of branch j
    size_t xt[IRIS_HIFIGAN_MAX_KERNELS];  // conv1 output of branch j
    size_t total; // floats
};

WsLayout ws_layout(const iris_hifigan_handle* h, int B, int T) {
    WsLayout w;
    const size_t frames = (size_t)B * T;
    size_t per_frame_max = 0;
    size_t L = 1;
    for (const auto& st : h->stages) {
        L *= st.rate;
        const size_t e = L * st.C;
        if (e > per_frame_max) per_frame_max = e;
    }
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += (floats + 63) & ~(size_t)63; return o; };
    w.pre = take(frames * h->pre.C_out);
    w.up = take(frames * per_frame_max);
    for (int j = 0; j < h->cfg.num_kernels; ++j) {
        w.y[j] = take(frames * per_frame_max);
        w.xt[j] = take(frames * per_frame_max);
    }
    w.total = off;
    return w;
}

void init_launch(ConvLaunch& a) { memset(&a, 0, sizeof(a)); a.out_stride = 1; }

// second packing of the ResBlock conv weights for the small-problem kernel (mrf_small_f32.h): float offsets of each layer's
// 16 x 16 fragments in blob_w16 (layers whose channel counts are not multiples of 16 keep -1: they never take that kernel)
size_t assign_w16_offsets(iris_hifigan_handle* h) {
    size_t off16 = 0;
    for (auto& st : h->stages)
        for (size_t j = 0; j < st.c1.size(); ++j)
            for (int half = 0; half < 2; ++half)
                for (auto& l : (half == 0 ? st.c1[j] : st.c2[j]))
                    if ((l.C_in & 15) == 0 && (l.C_out & 15) == 0) { l.w16f_off = off16; off16 += packed16_conv1d_floats(l.C_in, l.C_out, l.k); }
    return off16;
}


}  // namespace

namespace iris {
// Batch items one pass of a forward processes.  Batch items are independent, so a forward of a large batch runs as
// consecutive passes over sub-batches that share ONE workspace: the workspace is bounded by kPassFrames mel frames
// (229 KB per frame in fp32: 15 GB) instead of growing with the batch (58 GB at 256 x 1000 frames), at no cost in
// throughput -- the kernels are at their large-batch efficiency from ~16,000 frames on.
int pass_items(int B, int T) {
    if (B <= 1 || T <= 0) return B;
    const long long fit = kPassFrames / T;
    return (int)(fit < 1 ? 1 : (fit < B ? fit : B));
}
}  // namespace iris

extern "C" {

int32_t iris_hifigan_abi_version(void) { return IRIS_HIFIGAN_ABI_VERSION; }
const char* iris_hifigan_last_error(void) { return iris::g_err; }

int32_t iris_hifigan_weight_count(const iris_hifigan_config* cfg, uint64_t* count) {
    IRIS_ABI_BEGIN
    TRY(validate(cfg));
    if (!count) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "count is NULL");
    iris_hifigan_handle tmp;
    tmp.cfg = *cfg;
    build_layers(&tmp);
    *count = ref_weight_count(&tmp);
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_create(const iris_hifigan_config* cfg, const float* weights_host,
                            uint64_t n_weights, iris_hifigan_handle** out) {
    IRIS_ABI_BEGIN
    TRY(validate(cfg));
    if (!weights_host || !out) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    struct Owner {                      // frees a half-built generator on every early return, exceptions included
        iris_hifigan_handle* h = nullptr;
        ~Owner() { if (h) (void)iris_hifigan_destroy(h); }
    } owner;
    iris_hifigan_handle* h = new (std::nothrow) iris_hifigan_handle;
    if (!h) return fail(IRIS_HIFIGAN_OUT_OF_MEMORY, "host allocation failed");
    owner.h = h;
    h->cfg = *cfg;
    build_layers(h);
    const uint64_t expect = ref_weight_count(h);
    if (n_weights != expect)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "weight blob has %llu values, config needs %llu",
                    (unsigned long long)n_weights, (unsigned long long)expect);
    // Repacking: the MFMA fragment order of the persistent kernels (32 x 32 tiles) and, for the ResBlock convs, the 16 x 16
    // fragments of the short-input kernel (mrf_small_f32.h) -- two gathers over 13.9 M weights.  (layer, tap) pieces write
    // disjoint ranges, so they run on a few host threads (host_parallel.h): the reference's caller loads a model to vocode ONE
    // utterance (scripts/synthesize.py:197-198), so this is time that caller waits for.
    std::vector<float> host(h->blob_floats, 0.f);
    const size_t off16 = assign_w16_offsets(h);
    std::vector<float> host16(off16);
    std::vector<std::function<void()>> jobs;
    {
        const float* src = weights_host;
        for_each_layer(h, [&](ConvLayer& l) {
            float* dst = host.data() + l.w_off;
            const ConvLayer* lp = &l;
            if (l.kind == 2) {
                jobs.push_back([=] {            // [1][C][k] -> [k][C]
                    for (int c = 0; c < lp->C_in; ++c)
                        for (int kap = 0; kap < lp->k; ++kap) dst[(size_t)kap * lp->C_in + c] = src[(size_t)c * lp->k + kap];
                });
            } else if (l.kind == 1) {
                for (int ph = 0; ph < l.u; ++ph)
                    jobs.push_back([=] { pack_convt_weights(src, lp->C_in, lp->C_out, lp->k, lp->u, dst, ph, ph + 1); });
            } else {
                for (int kap = 0; kap < l.k; ++kap)
                    jobs.push_back([=] { pack_conv1d_weights(src, lp->C_in, lp->C_out, lp->k, dst, kap, kap + 1); });
                if (l.w16f_off != (size_t)-1) {
                    float* dst16 = host16.data() + l.w16f_off;
                    for (int kap = 0; kap < l.k; ++kap)
                        jobs.push_back([=] { pack_conv1d_weights16(src, lp->C_in, lp->C_out, lp->k, dst16, kap, kap + 1); });
                }
            }
            src += l.ref_w_floats;
            memcpy(host.data() + l.b_off, src, sizeof(float) * l.C_out);
            src += l.C_out;
        });
    }
    // the reference-layout weights stay on the host for the packings of the other dtypes (bf16 fragments, split-bf16 planes),
    // which are built by iris_hifigan_prepare / the first forward of that dtype; iris_hifigan_release_host_weights drops them
    jobs.push_back([=] { h->ref_weights.assign(weights_host, weights_host + n_weights); });
    run_host_jobs(jobs);
    // the generator lives on the device that is current now; every later call runs under that device
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess) e = hipMalloc(&h->blob, h->blob_floats * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(h->blob, host.data(), h->blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && off16 > 0) e = hipMalloc(&h->blob_w16, off16 * sizeof(float));
    if (e == hipSuccess && off16 > 0)
        e = hipMemcpy(h->blob_w16, host16.data(), off16 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? IRIS_HIFIGAN_OUT_OF_MEMORY : IRIS_HIFIGAN_HIP_ERROR,
                    "weight upload failed: %s", hipGetErrorString(e));
    if (hipMalloc(&h->tile_counters, kTileCounterWords * sizeof(unsigned)) != hipSuccess) h->tile_counters = nullptr;   // optional
    owner.h = nullptr;
    *out = h;
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_destroy(iris_hifigan_handle* h) {
    if (!h) return IRIS_HIFIGAN_OK;
    if (h->host_only) { delete h; return IRIS_HIFIGAN_OK; }
    DeviceGuard guard(h->device);       // the allocations belong to the handle's device, whatever is current now
    for (hipEvent_t ev : h->ev) (void)hipEventDestroy(ev);
    if (h->blob) (void)hipFree(h->blob);
    if (h->blob16) (void)hipFree(h->blob16);
    if (h->blob_s3) (void)hipFree(h->blob_s3);
    if (h->blob_w16) (void)hipFree(h->blob_w16);
    if (h->tile_counters) (void)hipFree(h->tile_counters);
    delete h;
    return IRIS_HIFIGAN_OK;
}

}  // extern "C"

namespace {

// Builds the weight packing `dtype` needs beyond what create uploaded (bf16 fragments / split-bf16 planes), once.
// Synchronous (packs on the host, allocates, uploads).  A packing counts as built only when its build SUCCEEDED or the
// configuration genuinely cannot have it (then the forward of that dtype reports UNSUPPORTED): a transient failure -- out
// of memory -- is reported and retried by the next call.  `stream` non-null-checked: inside a stream capture nothing may
// allocate or synchronise, so a forward that still needs a packing there is refused with NOT_PREPARED.
int ensure_prepared(iris_hifigan_handle* h, int32_t dtype, hipStream_t stream, bool from_forward) {
    const bool want_bf16 = dtype == IRIS_HIFIGAN_BF16 && !h->built_bf16;
    const bool want_s3 = dtype == IRIS_HIFIGAN_F32_SPLIT && !h->built_s3;
    if (!want_bf16 && !want_s3) return IRIS_HIFIGAN_OK;
    if (from_forward) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(IRIS_HIFIGAN_NOT_PREPARED, "the weight packing of dtype %d is not built yet and the stream is being captured: "
                        "call iris_hifigan_prepare(h, %d) before capturing forwards of this dtype", dtype, dtype);
    }
    if (h->ref_weights.empty())
        return fail(IRIS_HIFIGAN_NOT_PREPARED, "the host copy of the weights was released (iris_hifigan_release_host_weights) before "
                    "dtype %d was prepared", dtype);
    if (want_bf16) { TRY(bf16_build_blob(h, h->ref_weights.data())); h->built_bf16 = true; }
    if (want_s3)   { TRY(f32s_build_blob(h, h->ref_weights.data())); h->built_s3 = true; }
    if (h->built_bf16 && h->built_s3) std::vector<float>().swap(h->ref_weights);   // every packing exists
    return IRIS_HIFIGAN_OK;
}

}  // namespace

extern "C" {

int32_t iris_hifigan_prepare(iris_hifigan_handle* h, int32_t dtype) {
    IRIS_ABI_BEGIN
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    if (dtype != IRIS_HIFIGAN_F32 && dtype != IRIS_HIFIGAN_BF16 && dtype != IRIS_HIFIGAN_F32_SPLIT)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "dtype %d not supported", dtype);
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail(IRIS_HIFIGAN_HIP_ERROR, "cannot select device %d: %s", h->device, hipGetErrorString(guard.err));
    return ensure_prepared(h, dtype, nullptr, false);
    IRIS_ABI_END
}

int32_t iris_hifigan_release_host_weights(iris_hifigan_handle* h) {
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    std::vector<float>().swap(h->ref_weights);
    return IRIS_HIFIGAN_OK;
}

int32_t iris_hifigan_hop_length(const iris_hifigan_handle* h, int32_t* hop) {
    if (!h || !hop) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    *hop = h->hop;
    return IRIS_HIFIGAN_OK;
}

int32_t iris_hifigan_workspace_bytes(const iris_hifigan_handle* h, int32_t B, int32_t T,
                                     int32_t dtype, uint64_t* bytes) {
    if (!h || !bytes) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 0 || T < 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "negative shape");
    B = pass_items(B, T);                                   // a large batch runs as passes over sub-batches sharing the workspace
    if (dtype == IRIS_HIFIGAN_BF16) { *bytes = bf16_workspace_bytes(h, B, T); return IRIS_HIFIGAN_OK; }
    if (dtype != IRIS_HIFIGAN_F32 && dtype != IRIS_HIFIGAN_F32_SPLIT)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "dtype %d not supported", dtype);
    *bytes = ws_layout(h, B, T).total * sizeof(float);
    return IRIS_HIFIGAN_OK;
}

int32_t iris_hifigan_set_profiling(iris_hifigan_handle* h, int32_t enabled) {
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    h->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    h->profiling_paused = 0;
    h->n_rec = 0;
    h->n_ev = 0;
    return IRIS_HIFIGAN_OK;
}

int32_t iris_hifigan_pause_profiling(iris_hifigan_handle* h, int32_t paused) {
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    h->profiling_paused = paused != 0;
    return IRIS_HIFIGAN_OK;
}

int32_t iris_hifigan_read_profile(iris_hifigan_handle* h, iris_hifigan_launch_record* out,
                                  int32_t capacity, int32_t* n_launches) {
    IRIS_ABI_BEGIN
    if (!h || !n_launches) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    *n_launches = h->n_rec;
    for (int i = 0; i < h->n_rec; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->rec_ev[i].first, h->rec_ev[i].second));
        h->recs[i].ms = ms;
        if (out && i < capacity) out[i] = h->recs[i];
    }
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

}  // extern "C"

namespace {

// Checks shared by forward and forward_until.
int check_forward_args(const iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T, const void* workspace_dev,
                       int32_t dtype) {
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    if (dtype != IRIS_HIFIGAN_F32 && dtype != IRIS_HIFIGAN_BF16 && dtype != IRIS_HIFIGAN_F32_SPLIT)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "dtype %d not supported", dtype);
    if (B < 0 || T < 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "negative shape");
    if (B == 0 || T == 0) return IRIS_HIFIGAN_OK;
    if (!mel_dev || !workspace_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL device pointer");
    if (B > 65535) return fail(IRIS_HIFIGAN_UNSUPPORTED, "batch %d exceeds 65535 (grid.y)", B);
    if ((int64_t)T * h->hop > (int64_t)1 << 30)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "T*hop = %lld exceeds 2^30 rows", (long long)T * h->hop);
    return IRIS_HIFIGAN_OK;
}

// The fp32 / split-product forward.  `stop` (forward_until only): return after MRF step stop.step of stage
// stop.stage has been queued; *mean_in_y0 then says where that stage's result lies (forward_until's contract).
int forward_f32(iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T, void* wav_dev, void* workspace_dev,
                uint64_t workspace_bytes, int32_t dtype, hipStream_t stream, const ForwardStop& stop, int32_t* until_flags) {
    const WsLayout w = ws_layout(h, B, T);
    if (workspace_bytes < w.total * sizeof(float))
        return fail(IRIS_HIFIGAN_WORKSPACE_TOO_SMALL, "workspace has %llu bytes, need %llu",
                    (unsigned long long)workspace_bytes, (unsigned long long)(w.total * sizeof(float)));
    float* ws = (float*)workspace_dev;
    const float* blob = h->blob;
    const float slope = h->cfg.lrelu_slope;
    const int nk = h->cfg.num_kernels;
    Prof prof{h, stream, (h->profiling && !h->profiling_paused) ? h->n_rec : 0};
    const double fB = (double)B;
    // large batches: the MRF kernel's blocks draw tiles from per-launch counters (mrf_conv_mfma_f32.h); one
    // memset per forward zeroes them.  Below ~2000 frames no launch has enough tiles per block to use them.
    const bool dyn_tiles = h->tile_counters && (long long)B * T >= 2000 &&
                           (int)h->stages.size() * 2 * h->cfg.num_dilations[0] <= kTileCounterWords / 2;
    // the persistent pair kernels draw jobs from one counter word per launch (upper half of the array); a forward too short
    // to use either kind of counter skips the memset
    const bool pf_counters = h->tile_counters && (long long)B * T >= 100 &&
                             (int)h->stages.size() * h->cfg.num_dilations[0] <= kTileCounterWords / 2;
    // (zeroed where the first launch that reads them is about to be issued: a batch-1 forward of a few hundred frames has none)
    bool counters_zeroed = false;
    auto zero_counters = [&]() -> hipError_t {
        if (counters_zeroed || h->host_only) return hipSuccess;
        counters_zeroed = true;
        return hipMemsetAsync(h->tile_counters, 0, kTileCounterWords * sizeof(unsigned), stream);
    };
    if (dyn_tiles) HIP_TRY(zero_counters());

    // ---- conv_pre (hifigan_pretrained.py:124) ----
    {
        ConvLaunch a; init_launch(a);
        const ConvLayer& l = h->pre;
        a.p[0].x = (const float*)mel_dev; a.p[0].wp = (const f32x4*)(blob + l.w_off);
        a.p[0].bias = blob + l.b_off; a.p[0].res = nullptr; a.p[0].y = ws + w.pre;
        a.p[0].ks = l.k; a.p[0].dil = 1; a.p[0].pad_left = (l.k - 1) / 2;
        a.B = B; a.L_in = T; a.L_out = T; a.C_in = l.C_in; a.C_out = l.C_out; a.n_idx = T;
        a.in_act = IN_ACT_NONE; a.x_channels_first = 1; a.slope = slope;
        TRY(prof.begin(0, -1, 0, 2.0 * fB * T * l.C_in * l.C_out * l.k,
                       4.0 * (fB * T * (l.C_in + l.C_out) + (double)l.ref_w_floats + l.C_out)));
        HIP_TRY(launch_conv(a, 1, stream));
        TRY(prof.end());
    }

    int L = T;
    bool prev_summed = false;   // the previous stage left mean(branches) in y[0] (MRF kernel's summing step)
    for (size_t i = 0; i < h->stages.size(); ++i) {
        const Stage& st = h->stages[i];
        const int L_out = L * st.rate;
        // ---- LeakyReLU + ConvTranspose1d (hifigan_pretrained.py:127-128) ----
        {
            ConvLaunch a; init_launch(a);
            const ConvLayer& l = st.up;
            const int taps = convt_taps(l.k, l.u);
            a.p[0].wp = (const f32x4*)(blob + l.w_off); a.p[0].bias = blob + l.b_off;
            a.p[0].res = nullptr; a.p[0].y = ws + w.up;
            a.p[0].ks = taps; a.p[0].dil = 1; a.p[0].pad_left = taps - 1;
            // bytes are reported in accounting L (the MRF accumulation costs one extra read
            // per additional branch) whether or not the summing step already folded the mean
            const int n_in = i == 0 ? 1 : nk;
            if (i == 0) { a.p[0].x = ws + w.pre; a.in_act = IN_ACT_LRELU; }
            else if (prev_summed) { a.p[0].x = ws + w.y[0]; a.in_act = IN_ACT_LRELU; }
            else {
                a.in_act = IN_ACT_MRF_LRELU; a.n_mrf = nk;
                for (int j = 0; j < nk; ++j) a.xmrf[j] = ws + w.y[j];
                a.p[0].x = a.xmrf[0];
            }
            a.B = B; a.L_in = L; a.L_out = L_out; a.C_in = l.C_in; a.C_out = l.C_out;
            a.n_idx = L + taps - 1; a.out_stride = l.u; a.out_off = -(l.k - l.u) / 2;
            a.z_is_phase = 1;
            a.phase_wp_stride = (int64_t)(packed_convt_phase_floats(l.C_in, l.C_out, l.k, l.u) / 4);
            a.slope = slope;
            TRY(prof.begin(1, (int)i, 0, 2.0 * fB * L * l.C_in * l.C_out * l.k,
                           4.0 * (fB * L * l.C_in * n_in + fB * L_out * l.C_out +
                                  (double)l.ref_w_floats + l.C_out)));
            // split-product mode: single-input upsamplers (conv_pre output, or the folded branch mean) on split products.
            // Off in the release library: with the upsamplers split as well the worst observed waveform error grows
            // from 2e-5 to 5e-5 -- still inside 1e-4, but the mode keeps the wider margin (diagnostic builds: S3UPS=1).
            const int s3_ups = IRIS_DIAG_ENV("IRIS_HIFIGAN_S3UPS", 0);
            if (s3_ups && dtype == IRIS_HIFIGAN_F32_SPLIT && a.in_act == IN_ACT_LRELU && f32s_ups_applicable(h, l, L))
                TRY(f32s_launch_ups(h, l, a.p[0].x, a.p[0].y, B, L, stream));
            else if ((a.in_act == IN_ACT_LRELU || (a.in_act == IN_ACT_MRF_LRELU && nk == 3)) &&
                     convt_gemm_applicable(l.C_in, l.C_out, l.k, l.u, L, L_out, slope)) {
                // the whole layer as ONE GEMM [L + 1, 2 C_in] x [2 C_in, u C_out] (convt_mfma_f32.h), bit for bit the polyphase
                // launches below; its input is one tensor (conv_pre's output, or the MRF mean the previous stage's last step
                // stored) or the previous stage's three branch outputs, whose mean is then formed while the window is staged
                ConvtLaunch c; memset(&c, 0, sizeof(c));
                c.x = a.p[0].x; c.wp = a.p[0].wp; c.bias = a.p[0].bias; c.y = a.p[0].y;
                if (a.in_act == IN_ACT_MRF_LRELU) { c.x = a.xmrf[0]; c.x1 = a.xmrf[1]; c.x2 = a.xmrf[2]; }
                c.B = B; c.L_in = L; c.L_out = L_out; c.C_in = l.C_in; c.C_out = l.C_out; c.u = l.u; c.slope = slope;
                HIP_TRY(launch_convt_gemm(c, l.k, stream));
            } else
                HIP_TRY(launch_conv(a, l.u, stream));
            TRY(prof.end());
        }
        // ---- MRF: num_kernels ResBlocks advance together (hifigan_pretrained.py:64-71,131-136) ----
        const int nd = h->cfg.num_dilations[0];
        const double n_el = fB * L_out * st.C;
        const int use_mrf = IRIS_DIAG_ENV("IRIS_HIFIGAN_MRF", 1);
        const int use_sum = IRIS_DIAG_ENV("IRIS_HIFIGAN_MRFSUM", 1);
        // one conv step (half 0: convs1[m], half 1: convs2[m] + residual) of all branches, separate launches
        auto fill_step = [&](ConvLaunch& a, int m, int half, double& flops, double& wbytes) {
            init_launch(a);
            flops = 0; wbytes = 0;
            for (int j = 0; j < nk; ++j) {
                const ConvLayer& l = half == 0 ? st.c1[j][m] : st.c2[j][m];
                ConvProblem& p = a.p[j];
                const float* cur = (m == 0) ? ws + w.up : ws + w.y[j];  // x entering this pair
                if (half == 0) { p.x = cur; p.res = nullptr; p.y = ws + w.xt[j]; }
                else           { p.x = ws + w.xt[j]; p.res = cur; p.y = ws + w.y[j]; }
                p.wp = (const f32x4*)(blob + l.w_off); p.bias = blob + l.b_off;
                p.wp16 = (h->blob_w16 && l.w16f_off != (size_t)-1) ? (const f32x4*)(h->blob_w16 + l.w16f_off) : nullptr;
                p.ks = l.k; p.dil = l.dil; p.pad_left = l.dil * (l.k - 1) / 2;
                flops += 2.0 * n_el * l.C_in * l.k;
                wbytes += 4.0 * ((double)l.ref_w_floats + l.C_out);
            }
            a.B = B; a.L_in = L_out; a.L_out = L_out; a.C_in = st.C; a.C_out = st.C;
            a.n_idx = L_out; a.in_act = IN_ACT_LRELU; a.slope = slope;
            a.dyn_counter = dyn_tiles ? h->tile_counters + ((int)i * 2 * nd + 2 * m + half) : nullptr;
        };
        // ---- fused conv pairs (mrf_pair_f32.h; C = 32 / 64, exact fp32): conv1 -> xt in LDS -> conv2 + residual in ONE launch,
        // bit for bit the two separate launches.  A fused pair cannot run in place, so the running x of a branch alternates
        // between its y and xt buffers, arranged so that the last fused pair leaves it in y (where the separate launches and
        // the next layer expect it).  The last pair of the stage stays separate when its second step is the persistent
        // kernel's summing launch (which forms the MRF mean; large problems).  forward_until asking for a state after a
        // conv1 gets the separate launches for that stage.
        auto fill_pair = [&](PairLaunchF32& pa, int m, double& flops, double& wbytes) -> bool {
            memset(&pa, 0, sizeof(pa));
            flops = 0; wbytes = 0;
            bool ok = nk <= kMaxGroup;
            for (int j = 0; j < nk && ok; ++j) {
                const ConvLayer& c1 = st.c1[j][m];
                const ConvLayer& c2 = st.c2[j][m];
                ok = c1.k == c2.k && c2.dil == 1 && c1.C_in == st.C && c1.C_out == st.C && c2.C_in == st.C && c2.C_out == st.C;
                PairProblemF32& p = pa.p[j];
                p.w1 = (const f32x4*)(blob + c1.w_off); p.b1 = blob + c1.b_off;
                p.w2 = (const f32x4*)(blob + c2.w_off); p.b2 = blob + c2.b_off;
                p.ks = c1.k; p.dil = c1.dil;
                flops += 2.0 * n_el * st.C * (c1.k + c2.k);
                wbytes += 4.0 * ((double)c1.ref_w_floats + c1.C_out + (double)c2.ref_w_floats + c2.C_out);
            }
            pa.B = B; pa.L = L_out; pa.C = st.C; pa.slope = slope;
            return ok;
        };
        int n_fused = 0;
        bool pf = false;            // the fused pairs run on the persistent kernel (mrf_pair_f32_pf.h)
        bool fused_sum = false;     // ... and the stage's last pair forms the MRF mean itself (no persistent summing launches)
        if (dtype == IRIS_HIFIGAN_F32 && use_mrf && !(stop.stage == (int)i && !(stop.step & 1))) {
            bool all_ok = true;
            PairLaunchF32 pa0; double f0, wb0;
            for (int m = 0; m < nd && all_ok; ++m) all_ok = fill_pair(pa0, m, f0, wb0) && pair_f32_applicable(pa0, nk);
            if (all_ok) {
                bool sums = false;          // (the same decision as at the last step below)
                if (use_sum && nk == 3) {
                    ConvLaunch a; double f, wb;
                    fill_step(a, nd - 1, 1, f, wb);
                    if (mrf_kernel_applicable(a, nk)) {
                        const MrfPlan pq = mrf_plan(a, true);
                        ConvLaunch b = a;
                        b.p[0] = a.p[2]; b.p[2] = a.p[0];
                        b.sum_y = ws + w.y[0]; b.sum_div = (float)nk;
                        sums = mrf_kernel_applicable(b, nk) && !pq.zpar && !pq.small;
                    }
                }
                // The stage's LAST pair on the persistent summing kernel (mrf_pair_f32_pf.h: a block runs the three branches of
                // its tile and stores only the MRF mean -- no xt, no per-branch outputs, one launch instead of the persistent
                // kernel's two): taken where its whole-tile jobs fill at least four rounds of the chip well.  Measured
                // (profiles/r03_notes.md): +3 % on the C = 32 stage and +0.6 % on the step at batch 32 x 500, neutral at
                // batch 1 x 1000, a loss where a launch is one or two rounds (its jobs are 21 tap-units against 11 / 7 / 3).
                // The non-summing pairs stay on the one-job-per-block kernel: as persistent, prefetching blocks they
                // were 3-5 % slower at every size (diagnostic builds: IRIS_HIFIGAN_PAIR_PF_MODE=2).
                const PairPfTileF32 pt = pair_pf_f32_tile(st.C);
                const long long tiles_pf = (long long)((L_out + (pt.M - 10) - 1) / (pt.M - 10)) * B;
                pf = IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_PF_MODE", 0) == 2 && pair_pf_f32_applicable(pa0, nk, false);
                const int sum_env = IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_SUM", 1);          // 0 never, 1 by size, 2 always
                if (use_sum && sum_env != 0 && pair_pf_f32_applicable(pa0, nk, true)) {
                    const PairPfPlanF32 sp = pair_pf_f32_plan(tiles_pf, device_cu_count(), pt.MINB, true);
                    fused_sum = sum_env == 2 || (sp.efficiency >= 0.85 && tiles_pf >= 4LL * device_cu_count() * sp.per_cu);
                }
                n_fused = (fused_sum || !sums) ? nd : nd - 1;
            }
        }
        const float* cur_x[kMaxGroup];
        for (int j = 0; j < nk && j < kMaxGroup; ++j) cur_x[j] = ws + w.up;
        for (int m = 0; m < nd; ++m) {
            if (m < n_fused) {
                PairLaunchF32 pa; double flops, wbytes;
                (void)fill_pair(pa, m, flops, wbytes);
                const bool is_sum = fused_sum && m == nd - 1;
                // never in place: the running x of a branch alternates between its y and xt buffers, arranged so that the last
                // fused pair ends in y -- or, in front of the summing pair (which writes the mean to y[0]), in xt
                const bool to_y = fused_sum ? (((nd - 2 - m) & 1) != 0) : (((n_fused - 1 - m) & 1) == 0);
                for (int j = 0; j < nk; ++j) { pa.p[j].x = cur_x[j]; pa.p[j].y = to_y ? ws + w.y[j] : ws + w.xt[j]; }
                // algorithmic FLOP / bytes (accounting L) are those of both steps; the record carries the second step's index
                TRY(prof.begin(2, (int)i, 2 * m + 1, flops, 4.0 * n_el * nk * 5 + wbytes));
                // (a zeroed counter word per persistent launch: the upper half of the per-forward counters)
                unsigned* const ctr = (pf_counters && IRIS_DIAG_ENV("IRIS_HIFIGAN_PAIR_PF_DYN", 1)) ? h->tile_counters + kTileCounterWords / 2 + ((int)i * nd + m) : nullptr;
                if ((is_sum || pf) && ctr) HIP_TRY(zero_counters());
                if (is_sum)  HIP_TRY(launch_pair_f32_pf(pa, nk, ws + w.y[0], ctr, stream));
                else if (pf) HIP_TRY(launch_pair_f32_pf(pa, nk, nullptr, ctr, stream));
                else         HIP_TRY(launch_pair_f32(pa, nk, stream));
                TRY(prof.end());
                for (int j = 0; j < nk; ++j) cur_x[j] = pa.p[j].y;
                if (m == nd - 1) prev_summed = is_sum;
                if (stop.stage == (int)i && stop.step == 2 * m + 1) {
                    if (until_flags) *until_flags = is_sum ? IRIS_HIFIGAN_UNTIL_MEAN_IN_Y0 : (to_y ? 0 : IRIS_HIFIGAN_UNTIL_X_IN_XT);
                    TRY(prof.finish());
                    return IRIS_HIFIGAN_OK;
                }
                continue;
            }
            for (int half = 0; half < 2; ++half) {
                ConvLaunch a;
                double flops = 0, wbytes = 0;
                fill_step(a, m, half, flops, wbytes);
                TRY(prof.begin(2, (int)i, 2 * m + half, flops,
                               4.0 * n_el * nk * (half == 0 ? 2 : 3) + wbytes));
                const bool last_step = m == nd - 1 && half == 1;
                bool launched = false;
                if (dtype == IRIS_HIFIGAN_F32_SPLIT && f32s_step_applicable(h, st.C, L_out, nk)) {
                    // fp32 storage, split-bf16 products (conv_mfma_f32s.h); the branch mean is left to the consumer
                    F32sStep step;
                    for (int j = 0; j < nk; ++j) {
                        step.x[j] = a.p[j].x; step.res[j] = a.p[j].res; step.y[j] = a.p[j].y;
                        step.layer[j] = half == 0 ? &st.c1[j][m] : &st.c2[j][m];
                    }
                    // last step of the stage: the kernel folds the branch mean (written to y[0]; a lane overwrites only
                    // elements it has read itself as branch 0's residual)
                    const bool fold = IRIS_DIAG_ENV("IRIS_HIFIGAN_S3SUM", 1) && last_step;
                    TRY(f32s_launch_step(h, step, nk, B, L_out, st.C, fold ? ws + w.y[0] : nullptr, stream));
                    launched = true;
                    if (last_step) prev_summed = fold;
                }
                if (!launched && use_mrf && use_sum && last_step && nk == 3) {
                    // last step of the stage: the MRF kernel can form mean_j(y_j) itself.  It processes
                    // p[2], p[1], p[0]; passing the branches reversed makes that resblock 0, 1, 2 -- the
                    // reference's summation order (hifigan_pretrained.py:131-137).  The mean goes to y[0]
                    // (in place: each lane overwrites only elements it read itself as branch 0's residual).
                    ConvLaunch b = a;
                    b.p[0] = a.p[2]; b.p[2] = a.p[0];
                    b.sum_y = ws + w.y[0]; b.sum_div = (float)nk;
                    // (small problems run one branch per block -- mrf_plan's latency modes -- and cannot sum)
                    const MrfPlan pq = mrf_kernel_applicable(a, nk) ? mrf_plan(a, true) : MrfPlan{};
                    if (mrf_kernel_applicable(b, nk) && !pq.zpar && !pq.small) {
                        HIP_TRY(launch_mrf_conv(b, nk, stream));
                        launched = true; prev_summed = true;
                    }
                }
                if (!launched) {
                    if (last_step) prev_summed = false;
                    if (use_mrf && mrf_kernel_applicable(a, nk)) HIP_TRY(launch_mrf_conv(a, nk, stream));
                    else                                         HIP_TRY(launch_conv(a, nk, stream));
                }
                TRY(prof.end());
                if (stop.stage == (int)i && stop.step == 2 * m + half) {
                    if (until_flags) *until_flags = (last_step && prev_summed) ? IRIS_HIFIGAN_UNTIL_MEAN_IN_Y0 : 0;
                    TRY(prof.finish());
                    return IRIS_HIFIGAN_OK;
                }
            }
        }
        L = L_out;
    }

    // ---- LeakyReLU + conv_post + tanh (hifigan_pretrained.py:139-141) ----
    {
        post::ConvPostLaunch a; memset(&a, 0, sizeof(a));
        const ConvLayer& l = h->post;
        if (prev_summed) { a.x[0] = ws + w.y[0]; a.n_in = 1; }
        else { for (int j = 0; j < nk; ++j) a.x[j] = ws + w.y[j]; a.n_in = nk; }
        a.w = blob + l.w_off; a.bias = blob + l.b_off; a.y = (float*)wav_dev;
        a.B = B; a.L = L; a.C = l.C_in; a.k = l.k; a.slope = slope; a.inv_n = 1.0f / (float)nk;
        TRY(prof.begin(3, -1, 0, 2.0 * fB * L * l.C_in * l.k,
                       4.0 * (fB * L * l.C_in * nk + fB * L + (double)l.ref_w_floats + 1)));
        HIP_TRY(post::launch_conv_post(a, stream));
        TRY(prof.end());
    }
    TRY(prof.finish());
    return IRIS_HIFIGAN_OK;
}

}  // namespace

extern "C" {

int32_t iris_hifigan_forward(iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T,
                             void* wav_dev, void* workspace_dev, uint64_t workspace_bytes,
                             int32_t dtype, void* stream_) {
    IRIS_ABI_BEGIN
    TRY(check_forward_args(h, mel_dev, B, T, workspace_dev, dtype));
    if (B == 0 || T == 0) return IRIS_HIFIGAN_OK;  // empty batch / empty mel -> empty waveform
    if (!wav_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL device pointer");
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail(IRIS_HIFIGAN_HIP_ERROR, "cannot select device %d: %s", h->device, hipGetErrorString(guard.err));
    TRY(ensure_prepared(h, dtype, (hipStream_t)stream_, true));
    if (dtype == IRIS_HIFIGAN_F32_SPLIT && !h->blob_s3)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "split-product mode needs ResBlock channel counts that are multiples of 32");
    const ForwardStop none{-1, -1};
    const int Bp = pass_items(B, T);
    for (int b0 = 0; b0 < B; b0 += Bp) {
        const int nb = B - b0 < Bp ? B - b0 : Bp;
        const float* mel_p = (const float*)mel_dev + (size_t)b0 * h->cfg.in_channels * T;
        float* wav_p = (float*)wav_dev + (size_t)b0 * h->hop * T;
        if (dtype == IRIS_HIFIGAN_BF16)
            TRY(bf16_forward(h, mel_p, nb, T, wav_p, workspace_dev, workspace_bytes, (hipStream_t)stream_, none, nullptr));
        else
            TRY(forward_f32(h, mel_p, nb, T, wav_p, workspace_dev, workspace_bytes, dtype, (hipStream_t)stream_, none, nullptr));
    }
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_forward_until(iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T,
                                   void* workspace_dev, uint64_t workspace_bytes, int32_t dtype,
                                   int32_t stop_stage, int32_t stop_step, int32_t* flags, void* stream_) {
    IRIS_ABI_BEGIN
    TRY(check_forward_args(h, mel_dev, B, T, workspace_dev, dtype));
    if (B == 0 || T == 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "forward_until needs a non-empty input");
    if (stop_stage < 0 || stop_stage >= (int)h->stages.size() || stop_step < 0 || stop_step >= 2 * h->cfg.num_dilations[0])
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "no MRF step %d in stage %d", stop_step, stop_stage);
    if (pass_items(B, T) != B)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "forward_until takes shapes that run in one pass (B * T <= %d frames)", kPassFrames);
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail(IRIS_HIFIGAN_HIP_ERROR, "cannot select device %d: %s", h->device, hipGetErrorString(guard.err));
    TRY(ensure_prepared(h, dtype, (hipStream_t)stream_, true));
    if (dtype == IRIS_HIFIGAN_F32_SPLIT && !h->blob_s3)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "split-product mode needs ResBlock channel counts that are multiples of 32");
    const ForwardStop stop{stop_stage, stop_step};
    if (dtype == IRIS_HIFIGAN_BF16)
        return bf16_forward(h, mel_dev, B, T, nullptr, workspace_dev, workspace_bytes, (hipStream_t)stream_, stop, flags);
    return forward_f32(h, mel_dev, B, T, nullptr, workspace_dev, workspace_bytes, dtype, (hipStream_t)stream_, stop, flags);
    IRIS_ABI_END
}

int32_t iris_hifigan_describe_plan(const iris_hifigan_config* cfg, int32_t B, int32_t T, int32_t dtype, int32_t cu_count,
                                   iris_hifigan_plan* out) {
    IRIS_ABI_BEGIN
    TRY(validate(cfg));
    if (!out) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "out is NULL");
    memset(out, 0, sizeof(*out));
    // a generator without a device: the layer table, the offsets of every weight packing and fake (never dereferenced)
    // base pointers -- then the forward itself, with every launch recorded instead of issued
    iris_hifigan_handle h;
    h.cfg = *cfg;
    h.host_only = true;
    build_layers(&h);
    h.blob = reinterpret_cast<float*>((uintptr_t)0x10000000);
    h.tile_counters = reinterpret_cast<unsigned*>((uintptr_t)0x08000000);
    if (assign_w16_offsets(&h) > 0) h.blob_w16 = reinterpret_cast<float*>((uintptr_t)0x18000000);
    TRY(bf16_build_blob(&h, nullptr)); h.built_bf16 = true;
    TRY(f32s_build_blob(&h, nullptr)); h.built_s3 = true;
    void* const mel = reinterpret_cast<void*>((uintptr_t)0x40000000);
    void* const wav = reinterpret_cast<void*>((uintptr_t)0x50000000);
    void* const ws = reinterpret_cast<void*>((uintptr_t)0x100000000ull);
    TRY(check_forward_args(&h, mel, B, T, ws, dtype));
    if (dtype == IRIS_HIFIGAN_F32_SPLIT && !h.blob_s3)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "split-product mode needs ResBlock channel counts that are multiples of 32");
    uint64_t need = 0;
    TRY(iris_hifigan_workspace_bytes(&h, B, T, dtype, &need));
    out->workspace_bytes = need;
    out->cu_count = cu_count > 0 ? cu_count : 256;
    if (B == 0 || T == 0) return IRIS_HIFIGAN_OK;
    DryRunLaunch recs[IRIS_HIFIGAN_MAX_PLAN_LAUNCHES];
    DryRun dry{recs, IRIS_HIFIGAN_MAX_PLAN_LAUNCHES, 0, out->cu_count};
    struct Scope { DryRun* prev; Scope(DryRun* d) : prev(dry_run_slot()) { dry_run_slot() = d; } ~Scope() { dry_run_slot() = prev; } } scope(&dry);
    const ForwardStop none{-1, -1};
    int rc = IRIS_HIFIGAN_OK;
    const int Bp = pass_items(B, T);
    out->passes = (B + Bp - 1) / Bp;
    for (int b0 = 0; b0 < B && rc == IRIS_HIFIGAN_OK; b0 += Bp) {          // (the launches of every pass are recorded)
        const int nb = B - b0 < Bp ? B - b0 : Bp;
        rc = dtype == IRIS_HIFIGAN_BF16 ? bf16_forward(&h, mel, nb, T, wav, ws, need, nullptr, none, nullptr)
                                        : forward_f32(&h, mel, nb, T, wav, ws, need, dtype, nullptr, none, nullptr);
    }
    out->n_launches = dry.n;
    for (int i = 0; i < dry.n && i < IRIS_HIFIGAN_MAX_PLAN_LAUNCHES; ++i) {
        iris_hifigan_plan_launch& o = out->launches[i];
        strncpy(o.kernel, recs[i].kernel ? recs[i].kernel : "", sizeof(o.kernel) - 1);
        o.grid[0] = recs[i].grid[0]; o.grid[1] = recs[i].grid[1]; o.grid[2] = recs[i].grid[2];
        o.block = recs[i].block; o.lds_bytes = recs[i].lds_bytes;
    }
    return rc;
    IRIS_ABI_END
}

int32_t iris_hifigan_workspace_layout(const iris_hifigan_handle* h, int32_t B, int32_t T, int32_t dtype,
                                      iris_hifigan_workspace_map* out) {
    if (!h || !out) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 0 || T < 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "negative shape");
    memset(out, 0, sizeof(*out));
    B = pass_items(B, T);                                   // (the layout of one pass; forward_until takes single-pass shapes only)
    if (dtype == IRIS_HIFIGAN_BF16) return bf16_workspace_map(h, B, T, out);
    if (dtype != IRIS_HIFIGAN_F32 && dtype != IRIS_HIFIGAN_F32_SPLIT)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "dtype %d not supported", dtype);
    const WsLayout w = ws_layout(h, B, T);
    out->element_bytes = 4;
    out->pre_offset = w.pre * 4; out->up_offset = w.up * 4; out->total_bytes = w.total * 4;
    for (int j = 0; j < h->cfg.num_kernels; ++j) { out->y_offset[j] = w.y[j] * 4; out->xt_offset[j] = w.xt[j] * 4; }
    return IRIS_HIFIGAN_OK;
}

// ------------------------------------------------------------------------------------------------
// single-layer entry points
// ------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    float* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t upload(const std::vector<float>& v) {
        hipError_t e = hipMalloc(&p, v.size() * sizeof(float));
        if (e != hipSuccess) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
    }
};
}  // namespace

int32_t iris_hifigan_op_conv1d(const float* x_dev, const float* w_host, const float* bias_host,
                               const float* res_dev, float* y_dev, int32_t B, int32_t L,
                               int32_t C_in, int32_t C_out, int32_t k, int32_t dilation,
                               int32_t in_act, float slope, int32_t x_channels_first, void* stream_) {
    IRIS_ABI_BEGIN
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || C_out < 1 || k < 1 || !(k & 1) || dilation < 1 || B > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv1d shape");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<float> packed(packed_conv1d_floats(C_in, C_out, k) + C_out);
    pack_conv1d_weights(w_host, C_in, C_out, k, packed.data());
    const size_t boff = packed.size() - C_out;
    memcpy(packed.data() + boff, bias_host, sizeof(float) * C_out);
    DevBuf wb;
    HIP_TRY(wb.upload(packed));
    ConvLaunch a; init_launch(a);
    a.p[0].x = x_dev; a.p[0].wp = (const f32x4*)wb.p; a.p[0].bias = wb.p + boff; a.p[0].res = res_dev;
    a.p[0].y = y_dev; a.p[0].ks = k; a.p[0].dil = dilation; a.p[0].pad_left = dilation * (k - 1) / 2;
    a.B = B; a.L_in = L; a.L_out = L; a.C_in = C_in; a.C_out = C_out; a.n_idx = L;
    a.in_act = in_act ? IN_ACT_LRELU : IN_ACT_NONE; a.x_channels_first = x_channels_first; a.slope = slope;
    HIP_TRY(launch_conv(a, 1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_conv_transpose1d(const float* x_dev, const float* w_host,
                                         const float* bias_host, float* y_dev, int32_t B, int32_t L,
                                         int32_t C_in, int32_t C_out, int32_t k, int32_t u,
                                         int32_t in_act, float slope, void* stream_) {
    IRIS_ABI_BEGIN
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || C_out < 1 || u < 1 || k < u || ((k - u) & 1) || B > 65535 || u > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv_transpose1d shape");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t phase_floats = packed_convt_phase_floats(C_in, C_out, k, u);
    std::vector<float> packed(phase_floats * u + C_out);
    pack_convt_weights(w_host, C_in, C_out, k, u, packed.data());
    const size_t boff = packed.size() - C_out;
    memcpy(packed.data() + boff, bias_host, sizeof(float) * C_out);
    DevBuf wb;
    HIP_TRY(wb.upload(packed));
    const int taps = convt_taps(k, u);
    ConvLaunch a; init_launch(a);
    a.p[0].x = x_dev; a.p[0].wp = (const f32x4*)wb.p; a.p[0].bias = wb.p + boff; a.p[0].res = nullptr;
    a.p[0].y = y_dev; a.p[0].ks = taps; a.p[0].dil = 1; a.p[0].pad_left = taps - 1;
    a.B = B; a.L_in = L; a.L_out = L * u; a.C_in = C_in; a.C_out = C_out; a.n_idx = L + taps - 1;
    a.out_stride = u; a.out_off = -(k - u) / 2; a.z_is_phase = 1;
    a.phase_wp_stride = (int64_t)(phase_floats / 4);
    a.in_act = in_act ? IN_ACT_LRELU : IN_ACT_NONE; a.slope = slope;
    if (in_act && convt_gemm_applicable(C_in, C_out, k, u, L, L * u, slope)) {       // (what the forward launches for this layer)
        ConvtLaunch c; memset(&c, 0, sizeof(c));
        c.x = x_dev; c.wp = a.p[0].wp; c.bias = a.p[0].bias; c.y = y_dev;
        c.B = B; c.L_in = L; c.L_out = L * u; c.C_in = C_in; c.C_out = C_out; c.u = u; c.slope = slope;
        HIP_TRY(launch_convt_gemm(c, k, stream));
    } else
        HIP_TRY(launch_conv(a, u, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_conv_post(const float* x0_dev, const float* x1_dev, const float* x2_dev,
                                  const float* w_host, const float* bias_host, float* y_dev,
                                  int32_t B, int32_t L, int32_t C_in, int32_t k, float slope,
                                  void* stream_) {
    IRIS_ABI_BEGIN
    if (!x0_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || k < 1 || !(k & 1) || B > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv_post shape");
    if ((x1_dev == nullptr) != (x2_dev == nullptr))
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "x1 and x2 must both be given or both be NULL");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<float> wv((size_t)k * C_in + 1);
    for (int c = 0; c < C_in; ++c)
        for (int kap = 0; kap < k; ++kap) wv[(size_t)kap * C_in + c] = w_host[(size_t)c * k + kap];
    wv[(size_t)k * C_in] = bias_host[0];
    DevBuf wb;
    HIP_TRY(wb.upload(wv));
    post::ConvPostLaunch a; memset(&a, 0, sizeof(a));
    a.x[0] = x0_dev; a.n_in = 1;
    if (x1_dev) { a.x[1] = x1_dev; a.x[2] = x2_dev; a.n_in = 3; }
    a.w = wb.p; a.bias = wb.p + (size_t)k * C_in; a.y = y_dev;
    a.B = B; a.L = L; a.C = C_in; a.k = k; a.slope = slope; a.inv_n = 1.0f / (float)a.n_in;
    HIP_TRY(post::launch_conv_post(a, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_mrf_step(const float* const* x_dev, const float* const* w_host, const float* const* bias_host,
                                 const float* const* res_dev, float* const* y_dev, float* mean_dev,
                                 int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                 float slope, int32_t plan, void* stream_) {
    IRIS_ABI_BEGIN
    if (!x_dev || !w_host || !bias_host || !k || !dil || (!y_dev && !mean_dev))
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C < 1 || plan < 0 || plan > 6) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad mrf_step shape or plan");
    const int nk = 3;
    for (int j = 0; j < nk; ++j) {
        if (!x_dev[j] || !w_host[j] || !bias_host[j] || (!mean_dev && !y_dev[j])) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL branch argument");
        if (k[j] < 1 || !(k[j] & 1) || dil[j] < 1) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad branch kernel size / dilation");
    }
    hipStream_t stream = (hipStream_t)stream_;
    DevBuf wb[3], wb16[3];
    size_t boff[3];
    ConvLaunch a; init_launch(a);
    for (int j = 0; j < nk; ++j) {
        std::vector<float> packed(packed_conv1d_floats(C, C, k[j]) + ((size_t)C + 3 & ~(size_t)3));
        pack_conv1d_weights(w_host[j], C, C, k[j], packed.data());
        boff[j] = packed_conv1d_floats(C, C, k[j]);
        memcpy(packed.data() + boff[j], bias_host[j], sizeof(float) * C);
        HIP_TRY(wb[j].upload(packed));
        if ((C & 15) == 0) {
            std::vector<float> p16(packed16_conv1d_floats(C, C, k[j]));
            pack_conv1d_weights16(w_host[j], C, C, k[j], p16.data());
            HIP_TRY(wb16[j].upload(p16));
        }
        ConvProblem& p = a.p[j];
        p.wp16 = (const f32x4*)wb16[j].p;
        p.x = x_dev[j]; p.res = res_dev ? res_dev[j] : nullptr; p.y = y_dev ? y_dev[j] : nullptr;
        p.wp = (const f32x4*)wb[j].p; p.bias = wb[j].p + boff[j];
        p.ks = k[j]; p.dil = dil[j]; p.pad_left = dil[j] * (k[j] - 1) / 2;
    }
    a.B = B; a.L_in = L; a.L_out = L; a.C_in = C; a.C_out = C; a.n_idx = L; a.in_act = IN_ACT_LRELU; a.slope = slope;
    if (mean_dev) {
        // the summing step takes the branches reversed so that they are processed as resblock 0, 1, 2 (see forward)
        std::swap(a.p[0], a.p[2]);
        a.sum_y = mean_dev; a.sum_div = (float)nk;
        for (int j = 0; j < nk; ++j) if (!a.p[j].y) a.p[j].y = mean_dev;     // never written; keeps descriptors valid
    }
    if (!mrf_kernel_applicable(a, nk)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "shape cannot take the MRF kernel");
    const int force = plan == 0 ? -1 : (plan >= 4 ? plan : plan - 1);
    if (mean_dev && (force == 2 || force >= 4)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "the one-branch-per-block modes cannot form the mean");
    if (force == 4 && !mrf_small_applicable(a, nk)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "the small-problem kernel needs C %% 32 == 0");
    if (force >= 5 && !mrf_plan(a, true, force).zdyn) return fail(IRIS_HIFIGAN_UNSUPPORTED, "the job mode needs two or more C_in chunks (C >= 128)");
    HIP_TRY(launch_mrf_conv(a, nk, stream, force));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_mrf_pair(const float* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                 const float* const* w2_host, const float* const* b2_host, float* const* y_dev,
                                 float* mean_dev, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                 float slope, int32_t mode, void* stream_) {
    IRIS_ABI_BEGIN
    if (!x_dev || !w1_host || !b1_host || !w2_host || !b2_host || (!y_dev && !mean_dev) || !k || !dil)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C < 1 || mode < 0 || mode > 2) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad mrf_pair shape or mode");
    if (mean_dev && mode == 0) return fail(IRIS_HIFIGAN_UNSUPPORTED, "only the persistent kernel (modes 1, 2) forms the mean");
    const int nk = 3;
    hipStream_t stream = (hipStream_t)stream_;
    DevBuf wb[3];
    PairLaunchF32 pa; memset(&pa, 0, sizeof(pa));
    for (int j = 0; j < nk; ++j) {
        if (!x_dev[j] || !w1_host[j] || !b1_host[j] || !w2_host[j] || !b2_host[j] || (!mean_dev && !y_dev[j]))
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL branch argument");
        if (k[j] < 1 || !(k[j] & 1) || dil[j] < 1) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad branch kernel size / dilation");
        const size_t wf = packed_conv1d_floats(C, C, k[j]), cpad = ((size_t)C + 3) & ~(size_t)3;
        std::vector<float> packed(2 * wf + 2 * cpad);
        pack_conv1d_weights(w1_host[j], C, C, k[j], packed.data());
        pack_conv1d_weights(w2_host[j], C, C, k[j], packed.data() + wf);
        memcpy(packed.data() + 2 * wf, b1_host[j], sizeof(float) * C);
        memcpy(packed.data() + 2 * wf + cpad, b2_host[j], sizeof(float) * C);
        HIP_TRY(wb[j].upload(packed));
        PairProblemF32& p = pa.p[j];
        p.x = x_dev[j]; p.y = mean_dev ? mean_dev : y_dev[j];      // (summing launch: the branch outputs are never written)
        p.w1 = (const f32x4*)wb[j].p; p.w2 = (const f32x4*)(wb[j].p + wf);
        p.b1 = wb[j].p + 2 * wf; p.b2 = wb[j].p + 2 * wf + cpad;
        p.ks = k[j]; p.dil = dil[j];
    }
    pa.B = B; pa.L = L; pa.C = C; pa.slope = slope;
    if (!pair_f32_applicable(pa, nk)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "shape cannot take the fused fp32 pair kernel");
    if (mode == 1 || mode == 2) {
        if (!pair_pf_f32_applicable(pa, nk, mean_dev != nullptr)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "shape cannot take the persistent pair kernel");
        struct Word { unsigned* p = nullptr; ~Word() { if (p) (void)hipFree(p); } } ctr;
        if (mode == 1) {                                    // blocks draw their jobs from a counter (mode 2: fixed stride)
            HIP_TRY(hipMalloc(&ctr.p, sizeof(unsigned)));
            HIP_TRY(hipMemsetAsync(ctr.p, 0, sizeof(unsigned), stream));
        }
        HIP_TRY(launch_pair_f32_pf(pa, nk, mean_dev, ctr.p, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else {
        HIP_TRY(launch_pair_f32(pa, nk, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

// ------------------------------------------------------------------------------------------------
// PostNet (src/iris/postnet.py:48-67)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

struct iris_postnet_handle {
    int n_mels = 0, num_layers = 0, channels = 0, k = 0;
    std::vector<ConvLayer> layers;
    float* blob = nullptr;
    size_t blob_floats = 0;
    int device = 0;
};

extern "C" {

int32_t iris_postnet_create(int32_t n_mels, int32_t num_layers, int32_t channels, int32_t kernel_size,
                            const float* weights_host, uint64_t n_weights, iris_postnet_handle** out) {
    IRIS_ABI_BEGIN
    if (!weights_host || !out) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (n_mels < 1 || channels < 1 || num_layers < 2 || num_layers > 64 || kernel_size < 1 || !(kernel_size & 1))
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "PostNet needs n_mels, channels >= 1, 2 <= num_layers <= 64, odd kernel_size");
    iris_postnet_handle* h = new (std::nothrow) iris_postnet_handle;
    if (!h) return fail(IRIS_HIFIGAN_OUT_OF_MEMORY, "host allocation failed");
    h->n_mels = n_mels; h->num_layers = num_layers; h->channels = channels; h->k = kernel_size;
    uint64_t expect = 0;
    size_t off = 0;
    for (int i = 0; i < num_layers; ++i) {
        ConvLayer l;
        l.C_in = i == 0 ? n_mels : channels;
        l.C_out = i == num_layers - 1 ? n_mels : channels;
        l.k = kernel_size;
        l.ref_w_floats = (size_t)l.C_in * l.C_out * l.k;
        l.w_floats = packed_conv1d_floats(l.C_in, l.C_out, l.k);
        l.w_off = off; off += (l.w_floats + 3) & ~(size_t)3;
        l.b_off = off; off += ((size_t)l.C_out + 3) & ~(size_t)3;
        expect += l.ref_w_floats + l.C_out;
        h->layers.push_back(l);
    }
    h->blob_floats = off;
    if (n_weights != expect) {
        delete h;
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "weight blob has %llu values, PostNet needs %llu",
                    (unsigned long long)n_weights, (unsigned long long)expect);
    }
    std::vector<float> host(h->blob_floats, 0.f);
    const float* src = weights_host;
    for (const ConvLayer& l : h->layers) {
        pack_conv1d_weights(src, l.C_in, l.C_out, l.k, host.data() + l.w_off);
        src += l.ref_w_floats;
        memcpy(host.data() + l.b_off, src, sizeof(float) * l.C_out);
        src += l.C_out;
    }
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess) e = hipMalloc(&h->blob, h->blob_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->blob, host.data(), h->blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->blob) (void)hipFree(h->blob);
        delete h;
        return fail(IRIS_HIFIGAN_HIP_ERROR, "PostNet weight upload failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_postnet_destroy(iris_postnet_handle* h) {
    if (!h) return IRIS_HIFIGAN_OK;
    if (h->blob) (void)hipFree(h->blob);
    delete h;
    return IRIS_HIFIGAN_OK;
}

// workspace: two ping-pong hidden buffers [B, T, channels] and the residual [B, T, n_mels]
int32_t iris_postnet_workspace_bytes(const iris_postnet_handle* h, int32_t B, int32_t T, uint64_t* bytes) {
    if (!h || !bytes) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 0 || T < 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "negative shape");
    const size_t frames = (size_t)B * T;
    const size_t hid = (frames * h->channels + 63) & ~(size_t)63, res = (frames * h->n_mels + 63) & ~(size_t)63;
    *bytes = (2 * hid + res) * sizeof(float);
    return IRIS_HIFIGAN_OK;
}

int32_t iris_postnet_forward(iris_postnet_handle* h, const void* mel_dev, int32_t B, int32_t T,
                             void* out_dev, void* workspace_dev, uint64_t workspace_bytes, void* stream_) {
    IRIS_ABI_BEGIN
    if (!h) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL handle");
    if (B < 0 || T < 0) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "negative shape");
    if (B == 0 || T == 0) return IRIS_HIFIGAN_OK;
    if (!mel_dev || !out_dev || !workspace_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL device pointer");
    if (B > 65535) return fail(IRIS_HIFIGAN_UNSUPPORTED, "batch %d exceeds 65535 (grid.y)", B);
    uint64_t need = 0;
    TRY(iris_postnet_workspace_bytes(h, B, T, &need));
    if (workspace_bytes < need)
        return fail(IRIS_HIFIGAN_WORKSPACE_TOO_SMALL, "workspace has %llu bytes, need %llu",
                    (unsigned long long)workspace_bytes, (unsigned long long)need);
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail(IRIS_HIFIGAN_HIP_ERROR, "cannot select device %d: %s", h->device, hipGetErrorString(guard.err));
    hipStream_t stream = (hipStream_t)stream_;
    const size_t frames = (size_t)B * T;
    const size_t hid = (frames * h->channels + 63) & ~(size_t)63;
    float* ws = (float*)workspace_dev;
    float* hbuf[2] = {ws, ws + hid};
    float* res = ws + 2 * hid;
    const float* x = (const float*)mel_dev;
    for (int i = 0; i < h->num_layers; ++i) {
        const ConvLayer& l = h->layers[i];
        const bool last = i == h->num_layers - 1;
        ConvLaunch a; init_launch(a);
        a.p[0].x = x; a.p[0].wp = (const f32x4*)(h->blob + l.w_off); a.p[0].bias = h->blob + l.b_off;
        a.p[0].res = nullptr; a.p[0].y = last ? res : hbuf[i & 1];
        a.p[0].ks = l.k; a.p[0].dil = 1; a.p[0].pad_left = (l.k - 1) / 2;
        a.B = B; a.L_in = T; a.L_out = T; a.C_in = l.C_in; a.C_out = l.C_out; a.n_idx = T;
        a.in_act = IN_ACT_NONE; a.x_channels_first = i == 0 ? 1 : 0;   // the mel arrives [B, n_mels, T]
        a.out_act = last ? 0 : 1;                                       // tanh (postnet.py:59)
        HIP_TRY(launch_conv(a, 1, stream));
        x = a.p[0].y;
    }
    dim3 grid((unsigned)((T + 255) / 256), (unsigned)B), block(256);
    HIP_TRY(launch_kernel(postnet_residual_kernel, grid, block, 0, stream, (const float*)mel_dev, (const float*)res,
                          (float*)out_dev, h->n_mels, T));              // x + res (postnet.py:67)
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

}  // extern "C"
