// iris_hifigan_bf16.hip -- the bf16-storage generator forward (dtype IRIS_HIFIGAN_BF16) behind
// iris_hifigan_forward.  Same launch plan as the fp32 path (iris_hifigan.hip; reference
// HiFiGANModel.forward, src/iris/hifigan_pretrained.py:123-143):
//   conv_pre (reads the fp32 channels-first mel) -> per stage { upsample (u phases), 2*num_dilations
//   grouped MRF launches } -> conv_post + tanh (fp32 waveform out).
// The MRF mean is formed by the consumer of a stage while it stages its input (SURVEY.md 8d accounting L).
#include "generator_internal.h"
#include "host_parallel.h"
#include "conv_mfma_bf16.h"
#include "convt_mfma_bf16.h"
#include "mrf_pair_bf16.h"
#include "conv_mfma_f32s.h"
#include "conv_post.h"

namespace iris {

using namespace b16;

namespace {

struct Ws16 {            // bf16 elements
    size_t pre, up, y[IRIS_HIFIGAN_MAX_KERNELS], xt[IRIS_HIFIGAN_MAX_KERNELS], total;
};

Ws16 ws16_layout(const iris_hifigan_handle* h, int B, int T) {
    Ws16 w;
    const size_t frames = (size_t)B * T;
    size_t per_frame_max = 0, L = 1;
    for (const auto& st : h->stages) {
        L *= st.rate;
        const size_t e = L * st.C;
        if (e > per_frame_max) per_frame_max = e;
    }
    size_t off = 0;
    auto take = [&](size_t halfs) { size_t o = off; off += (halfs + 127) & ~(size_t)127; return o; };
    w.pre = take(frames * h->pre.C_out);
    w.up = take(frames * per_frame_max);
    for (int j = 0; j < h->cfg.num_kernels; ++j) {
        w.y[j] = take(frames * per_frame_max);
        w.xt[j] = take(frames * per_frame_max);
    }
    w.total = off;
    return w;
}

void init_launch(Launch& a) { memset(&a, 0, sizeof(a)); a.out_stride = 1; }

}  // namespace

int bf16_build_blob(iris_hifigan_handle* h, const float* weights_host) {
    // every channel count of the bf16 path must be a multiple of 8 (16-byte bf16 pieces)
    bool ok = h->cfg.num_kernels <= kMaxGroup && (h->pre.C_out % 8) == 0;
    for (const auto& st : h->stages) ok = ok && (st.C % 8) == 0;
    if (!ok) { h->blob16 = nullptr; h->blob16_halfs = 0; return IRIS_HIFIGAN_OK; }   // bf16 forward reports UNSUPPORTED
    size_t off = 0;
    for_each_layer(h, [&](ConvLayer& l) {
        if (l.kind == 2)      l.w16_halfs = 0;                      // conv_post keeps fp32 weights
        else if (l.kind == 1) l.w16_halfs = packed_convt_phase_halfs(l.C_in, l.C_out, l.k, l.u) * l.u;
        else                  l.w16_halfs = packed_conv1d_halfs(l.C_in, l.C_out, l.k);
        l.w16_off = off;
        off += (l.w16_halfs + 127) & ~(size_t)127;
    });
    h->blob16_halfs = off;
    if (h->host_only) { h->blob16 = reinterpret_cast<uint16_t*>((uintptr_t)0x20000000); return IRIS_HIFIGAN_OK; }   // offsets only
    std::vector<uint16_t> host(off, 0);
    const float* src = weights_host;
    std::vector<std::function<void()>> jobs;          // one per layer, on a few host threads (host_parallel.h)
    for_each_layer(h, [&](ConvLayer& l) {
        uint16_t* dst = host.data() + l.w16_off;
        const ConvLayer* lp = &l;
        if (l.kind == 1)      jobs.push_back([=] { pack_convt_bf16(src, lp->C_in, lp->C_out, lp->k, lp->u, dst); });
        else if (l.kind == 0) jobs.push_back([=] { pack_conv1d_bf16(src, lp->C_in, lp->C_out, lp->k, dst); });
        src += l.ref_w_floats + l.C_out;
    });
    run_host_jobs(jobs);
    hipError_t e = hipMalloc(&h->blob16, off * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(h->blob16, host.data(), off * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->blob16) (void)hipFree(h->blob16);
        h->blob16 = nullptr;
        return fail(e == hipErrorOutOfMemory ? IRIS_HIFIGAN_OUT_OF_MEMORY : IRIS_HIFIGAN_HIP_ERROR,
                    "bf16 weight upload failed: %s", hipGetErrorString(e));
    }
    return IRIS_HIFIGAN_OK;
}

// ---- split-product mode: hi/mid planes of every ResBlock conv -------------------------------------------
int f32s_build_blob(iris_hifigan_handle* h, const float* weights_host) {
    h->blob_s3 = nullptr;
    for (const auto& st : h->stages)
        if (st.C < 32 || (st.C & 31)) return IRIS_HIFIGAN_OK;          // mode unavailable for this config
    size_t off = 0;
    for (auto& st : h->stages) {
        st.up.ws3_off = off;
        off += (2 * s3::packed_convt_plane_halfs(st.up.C_in, st.up.C_out, st.up.k, st.up.u) + 127) & ~(size_t)127;
        for (size_t j = 0; j < st.c1.size(); ++j)
            for (int half = 0; half < 2; ++half)
                for (auto& l : (half == 0 ? st.c1[j] : st.c2[j])) {
                    l.ws3_off = off;
                    off += (2 * s3::packed_plane_halfs(l.C_in, l.C_out, l.k) + 127) & ~(size_t)127;
                }
    }
    if (h->host_only) { h->blob_s3 = reinterpret_cast<uint16_t*>((uintptr_t)0x30000000); return IRIS_HIFIGAN_OK; }   // offsets only
    std::vector<uint16_t> host(off, 0);
    const float* src = weights_host;
    std::vector<std::function<void()>> jobs;
    for_each_layer(h, [&](ConvLayer& l) {
        uint16_t* dst = host.data() + l.ws3_off;
        const ConvLayer* lp = &l;
        if (l.kind == 0 && &l != &h->pre) jobs.push_back([=] { s3::pack_conv1d_split(src, lp->C_in, lp->C_out, lp->k, dst); });
        if (l.kind == 1) jobs.push_back([=] { s3::pack_convt_split(src, lp->C_in, lp->C_out, lp->k, lp->u, dst); });
        src += l.ref_w_floats + l.C_out;
    });
    run_host_jobs(jobs);
    hipError_t e = hipMalloc(&h->blob_s3, off * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(h->blob_s3, host.data(), off * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->blob_s3) (void)hipFree(h->blob_s3);
        h->blob_s3 = nullptr;
        return fail(e == hipErrorOutOfMemory ? IRIS_HIFIGAN_OUT_OF_MEMORY : IRIS_HIFIGAN_HIP_ERROR,
                    "split-product weight upload failed: %s", hipGetErrorString(e));
    }
    return IRIS_HIFIGAN_OK;
}

bool f32s_step_applicable(const iris_hifigan_handle* h, int C, int L, int nk) {
    if (!h->blob_s3 || nk > s3::kMaxGroup) return false;
    s3::Launch a; memset(&a, 0, sizeof(a));
    a.C = C; a.L = L;
    return s3::applicable(a, nk);
}

int f32s_launch_step(iris_hifigan_handle* h, const F32sStep& st, int nk, int B, int L, int C, float* sum_y, hipStream_t stream) {
    s3::Launch a; memset(&a, 0, sizeof(a));
    for (int j = 0; j < nk; ++j) {
        const ConvLayer& l = *st.layer[j];
        s3::Problem& p = a.p[j];
        p.x = st.x[j]; p.res = st.res[j]; p.y = st.y[j];
        p.wp = h->blob_s3 + l.ws3_off; p.bias = h->blob + l.b_off;
        p.ks = l.k; p.dil = l.dil; p.pad_left = l.dil * (l.k - 1) / 2;
    }
    a.B = B; a.L = L; a.C = C; a.slope = h->cfg.lrelu_slope; a.sum_y = sum_y;
    HIP_TRY(s3::launch(a, nk, stream));
    return IRIS_HIFIGAN_OK;
}

namespace {
// LeakyReLU + ConvTranspose1d of one stage as u phase launches in one grid (split products)
void fill_ups(s3::Launch& a, const ConvLayer& l, const float* x, const void* wp, const float* bias, float* y,
              int B, int L_in, float slope) {
    memset(&a, 0, sizeof(a));
    const int taps = b16::convt_taps(l.k, l.u);
    a.p[0].x = x; a.p[0].wp = wp; a.p[0].bias = bias; a.p[0].res = nullptr; a.p[0].y = y;
    a.p[0].ks = taps; a.p[0].dil = 1; a.p[0].pad_left = taps - 1;
    a.B = B; a.L = L_in * l.u; a.C = l.C_out; a.slope = slope;
    a.C_in = l.C_in; a.L_in = L_in; a.n_idx = L_in + taps - 1;
    a.out_stride = l.u; a.out_off = -(l.k - l.u) / 2; a.z_is_phase = 1;
    a.phase_bytes = (unsigned)(b16::packed_convt_phase_halfs(l.C_in, l.C_out, l.k, l.u) * 2);
    a.plane_bytes[0] = (unsigned)(s3::packed_convt_plane_halfs(l.C_in, l.C_out, l.k, l.u) * 2);
}
}  // namespace

bool f32s_ups_applicable(const iris_hifigan_handle* h, const ConvLayer& l, int L_in) {
    if (!h->blob_s3 || l.kind != 1 || l.u > 65535) return false;
    s3::Launch a;
    fill_ups(a, l, nullptr, nullptr, nullptr, nullptr, 1, L_in, 0.f);
    return s3::applicable(a, l.u) && (double)s3::packed_convt_plane_halfs(l.C_in, l.C_out, l.k, l.u) * 4.0 < 2147483648.0;
}

int f32s_launch_ups(iris_hifigan_handle* h, const ConvLayer& l, const float* x, float* y, int B, int L_in, hipStream_t stream) {
    s3::Launch a;
    fill_ups(a, l, x, h->blob_s3 + l.ws3_off, h->blob + l.b_off, y, B, L_in, h->cfg.lrelu_slope);
    HIP_TRY(s3::launch(a, l.u, stream));
    return IRIS_HIFIGAN_OK;
}

uint64_t bf16_workspace_bytes(const iris_hifigan_handle* h, int B, int T) {
    return ws16_layout(h, B, T).total * sizeof(uint16_t);
}

int bf16_workspace_map(const iris_hifigan_handle* h, int B, int T, iris_hifigan_workspace_map* out) {
    const Ws16 w = ws16_layout(h, B, T);
    out->element_bytes = 2;
    out->pre_offset = w.pre * 2; out->up_offset = w.up * 2; out->total_bytes = w.total * 2;
    for (int j = 0; j < h->cfg.num_kernels; ++j) { out->y_offset[j] = w.y[j] * 2; out->xt_offset[j] = w.xt[j] * 2; }
    return IRIS_HIFIGAN_OK;
}

int bf16_forward(iris_hifigan_handle* h, const void* mel_dev, int B, int T, void* wav_dev,
                 void* workspace_dev, uint64_t workspace_bytes, hipStream_t stream, const ForwardStop& stop,
                 int32_t* until_flags) {
    if (!h->blob16)
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "bf16 path needs channel counts that are multiples of 8 and at most %d MRF kernels", kMaxGroup);
    {   // the bf16 kernels address one batch item's tensor with 32-bit byte offsets and have no wider fallback
        double per_item = (double)T * h->pre.C_out, L = T;
        for (const auto& st : h->stages) { L *= st.rate; if (L * st.C > per_item) per_item = L * st.C; }
        if (per_item * 2.0 >= 2147483648.0)
            return fail(IRIS_HIFIGAN_UNSUPPORTED, "bf16 path: %d frames make a single item's activations 2^31 bytes or more; "
                        "split the utterance (iris.streaming) or use fp32", T);
    }
    const Ws16 w = ws16_layout(h, B, T);
    if (workspace_bytes < w.total * sizeof(uint16_t))
        return fail(IRIS_HIFIGAN_WORKSPACE_TOO_SMALL, "workspace has %llu bytes, need %llu",
                    (unsigned long long)workspace_bytes, (unsigned long long)(w.total * sizeof(uint16_t)));
    uint16_t* ws = (uint16_t*)workspace_dev;
    const uint16_t* wb = h->blob16;
    const float* blob = h->blob;
    const float slope = h->cfg.lrelu_slope;
    const int nk = h->cfg.num_kernels;
    Prof prof{h, stream, h->profiling ? h->n_rec : 0};
    const double fB = (double)B;

    // ---- conv_pre (hifigan_pretrained.py:124): fp32 mel in, bf16 out ----
    {
        Launch a; init_launch(a);
        const ConvLayer& l = h->pre;
        a.p[0].x = mel_dev; a.p[0].wp = wb + l.w16_off; a.p[0].bias = blob + l.b_off;
        a.p[0].res = nullptr; a.p[0].y = ws + w.pre;
        a.p[0].ks = l.k; a.p[0].dil = 1; a.p[0].pad_left = (l.k - 1) / 2;
        a.B = B; a.L_in = T; a.L_out = T; a.C_in = l.C_in; a.C_out = l.C_out; a.n_idx = T;
        a.in_act = IN_ACT_NONE; a.x_f32_cf = 1; a.slope = slope;
        TRY(prof.begin(0, -1, 0, 2.0 * fB * T * l.C_in * l.C_out * l.k,
                       fB * T * (4.0 * l.C_in + 2.0 * l.C_out) + 2.0 * (double)l.ref_w_floats + 4.0 * l.C_out));
        HIP_TRY(launch_conv_bf16(a, 1, stream));
        TRY(prof.end());
    }

    int L = T;
    // where the running x of branch j lives: the fused pair kernel cannot work in place (a block's input window
    // overlaps its neighbours' output rows), so its output alternates between the y and the xt buffer of the branch
    const uint16_t* cur[IRIS_HIFIGAN_MAX_KERNELS] = {nullptr};
    // the previous stage's last pair ran on the summing kernel (mrf_pair_bf16.h): its ONE output is the next layer's operand
    const uint16_t* mean16 = nullptr;     // bf16(LeakyReLU(MRF mean)), for the next ConvTranspose1d
    const float* mean32 = nullptr;        // the fp32 MRF mean of the last stage, for conv_post
    for (size_t i = 0; i < h->stages.size(); ++i) {
        const Stage& st = h->stages[i];
        const int L_out = L * st.rate;
        // ---- LeakyReLU + ConvTranspose1d (hifigan_pretrained.py:127-128) ----
        {
            Launch a; init_launch(a);
            const ConvLayer& l = st.up;
            const int taps = convt_taps(l.k, l.u);
            a.p[0].wp = wb + l.w16_off; a.p[0].bias = blob + l.b_off;
            a.p[0].res = nullptr; a.p[0].y = ws + w.up;
            a.p[0].ks = taps; a.p[0].dil = 1; a.p[0].pad_left = taps - 1;
            const int n_in = i == 0 ? 1 : nk;       // (accounting L counts the reference's three branch tensors whatever was fused)
            if (i == 0) { a.p[0].x = ws + w.pre; a.in_act = IN_ACT_LRELU; }
            else if (mean16) { a.p[0].x = mean16; a.in_act = IN_ACT_NONE; }      // already activated and rounded
            else {
                a.in_act = IN_ACT_MRF_LRELU; a.n_mrf = nk;
                for (int j = 0; j < nk; ++j) a.xmrf[j] = cur[j];
                a.p[0].x = a.xmrf[0];
            }
            a.B = B; a.L_in = L; a.L_out = L_out; a.C_in = l.C_in; a.C_out = l.C_out;
            a.n_idx = L + taps - 1; a.out_stride = l.u; a.out_off = -(l.k - l.u) / 2;
            a.z_is_phase = 1;
            a.phase_wp_bytes = (unsigned)(packed_convt_phase_halfs(l.C_in, l.C_out, l.k, l.u) * 2);
            a.slope = slope;
            TRY(prof.begin(1, (int)i, 0, 2.0 * fB * L * l.C_in * l.C_out * l.k,
                           2.0 * (fB * L * l.C_in * n_in + fB * L_out * l.C_out + (double)l.ref_w_floats) + 4.0 * l.C_out));
            HIP_TRY(launch_convt_bf16(a, l.k, l.u, stream));      // one GEMM launch (convt_mfma_bf16.h) where it applies
            TRY(prof.end());
        }
        // ---- MRF: num_kernels ResBlocks advance together (hifigan_pretrained.py:64-71,131-136) ----
        const int nd = h->cfg.num_dilations[0];
        const double n_el = fB * L_out * st.C;
        for (int j = 0; j < nk; ++j) cur[j] = ws + w.up;
        mean16 = nullptr;
        auto report_stop = [&]() -> int {
            if (until_flags) *until_flags = (cur[0] == ws + w.xt[0]) ? IRIS_HIFIGAN_UNTIL_X_IN_XT : 0;
            return prof.finish();
        };
        for (int m = 0; m < nd; ++m) {
            // C <= 64: conv1 and conv2 of the pair in ONE launch, xt stays in LDS (mrf_pair_bf16.h): two tensor passes
            // over HBM instead of five.  Algorithmic FLOP / bytes (accounting L) are those of both steps; the record
            // carries the index of the pair's second step.  (forward_until asking for the state after conv1 gets the
            // two separate launches for that pair.)
            {
                PairLaunch pa; memset(&pa, 0, sizeof(pa));
                double flops = 0, wbytes = 0;
                for (int j = 0; j < nk && j < kMaxGroup; ++j) {
                    const ConvLayer& l1 = st.c1[j][m];
                    const ConvLayer& l2 = st.c2[j][m];
                    PairProblem& p = pa.p[j];
                    p.x = cur[j];
                    p.y = (cur[j] == ws + w.y[j]) ? ws + w.xt[j] : ws + w.y[j];
                    p.w1 = wb + l1.w16_off; p.b1 = blob + l1.b_off;
                    p.w2 = wb + l2.w16_off; p.b2 = blob + l2.b_off;
                    p.ks = l1.k; p.dil = l1.dil;
                    flops += 2.0 * n_el * (l1.C_in * l1.k + l2.C_in * l2.k);
                    wbytes += 2.0 * (double)(l1.ref_w_floats + l2.ref_w_floats) + 4.0 * (l1.C_out + l2.C_out);
                }
                pa.B = B; pa.L = L_out; pa.C = st.C; pa.slope = slope;
                const bool want_xt = stop.stage == (int)i && stop.step == 2 * m;
                bool same_k = true;
                for (int j = 0; j < nk; ++j) same_k = same_k && st.c1[j][m].k == st.c2[j][m].k && st.c2[j][m].dil == 1;
                // The stage's LAST pair: one block runs the three branches of its rows and stores only the MRF mean, as the
                // operand of the layer that follows (bf16, activated) or -- last stage -- as the fp32 mean conv_post takes.
                // Not when the caller asked for a state of this stage (forward_until returns branch tensors).  The fp32 mean
                // of the last stage is twice a bf16 tensor: it goes into `up` + y[0], adjacent in the workspace and both
                // free while the pair reads xt[j] (an odd number of pairs per ResBlock; otherwise conv_post reads three).
                if (m == nd - 1 && stop.stage != (int)i && same_k && nk == 3) {
                    const bool last_stage = i + 1 == h->stages.size();
                    bool room = true;
                    if (last_stage) {
                        room = w.y[0] >= w.up && (w.y[0] - w.up) * 2 >= (size_t)n_el * 2 && (w.y[0] - w.up) <= (size_t)n_el + 128;
                        for (int j = 0; j < nk; ++j) room = room && cur[j] == ws + w.xt[j];
                        post::ConvPostLaunch probe; memset(&probe, 0, sizeof(probe));       // conv_post must take one fp32 input of this shape
                        probe.B = B; probe.L = L_out; probe.C = st.C; probe.n_in = 1; probe.k = h->post.k;
                        room = room && h->post.C_in == st.C && post::conv_post_rows_ok(probe, false);
                    }
                    if (room && pair_sum_applicable(pa, nk, last_stage)) {
                        void* dst = last_stage ? (void*)(ws + w.up) : (void*)pa.p[0].y;
                        TRY(prof.begin(2, (int)i, 2 * m + 1, flops, 2.0 * n_el * nk * 5 + wbytes));
                        HIP_TRY(launch_pair_bf16_sum(pa, dst, last_stage, stream));
                        TRY(prof.end());
                        if (last_stage) mean32 = (const float*)dst; else mean16 = (const uint16_t*)dst;
                        for (int j = 0; j < nk; ++j) cur[j] = nullptr;
                        continue;
                    }
                }
                if (!want_xt && same_k && pair_applicable(pa, nk)) {
                    TRY(prof.begin(2, (int)i, 2 * m + 1, flops, 2.0 * n_el * nk * 5 + wbytes));
                    HIP_TRY(launch_pair_bf16(pa, nk, stream));
                    TRY(prof.end());
                    for (int j = 0; j < nk; ++j) cur[j] = pa.p[j].y;
                    if (stop.stage == (int)i && stop.step == 2 * m + 1) return report_stop();
                    continue;
                }
            }
            for (int half = 0; half < 2; ++half) {
                Launch a; init_launch(a);
                double flops = 0, wbytes = 0;
                for (int j = 0; j < nk; ++j) {
                    const ConvLayer& l = half == 0 ? st.c1[j][m] : st.c2[j][m];
                    Problem& p = a.p[j];
                    // x entering this pair is cur[j]; conv1 writes the branch's other buffer, conv2 (own rows only:
                    // in place is safe) writes back to cur[j], or to y[j] when cur[j] is the shared upsample output
                    uint16_t* tmp = (cur[j] == ws + w.xt[j]) ? ws + w.y[j] : ws + w.xt[j];
                    uint16_t* dst = (cur[j] == ws + w.up) ? ws + w.y[j] : const_cast<uint16_t*>(cur[j]);
                    if (half == 0) { p.x = cur[j]; p.res = nullptr; p.y = tmp; }
                    else           { p.x = tmp; p.res = cur[j]; p.y = dst; }
                    p.wp = wb + l.w16_off; p.bias = blob + l.b_off;
                    p.ks = l.k; p.dil = l.dil; p.pad_left = l.dil * (l.k - 1) / 2;
                    flops += 2.0 * n_el * l.C_in * l.k;
                    wbytes += 2.0 * (double)l.ref_w_floats + 4.0 * l.C_out;
                }
                a.B = B; a.L_in = L_out; a.L_out = L_out; a.C_in = st.C; a.C_out = st.C;
                a.n_idx = L_out; a.in_act = IN_ACT_LRELU; a.slope = slope;
                TRY(prof.begin(2, (int)i, 2 * m + half, flops, 2.0 * n_el * nk * (half == 0 ? 2 : 3) + wbytes));
                HIP_TRY(launch_conv_bf16(a, nk, stream));
                TRY(prof.end());
                if (half == 1) for (int j = 0; j < nk; ++j) cur[j] = a.p[j].y;
                if (stop.stage == (int)i && stop.step == 2 * m + half) {
                    // after conv1 the flag says where xt is NOT: x is still in cur[], xt in the other buffer
                    return report_stop();
                }
            }
        }
        L = L_out;
    }

    // ---- LeakyReLU + conv_post + tanh (hifigan_pretrained.py:139-141): bf16 in, fp32 waveform out ----
    {
        const ConvLayer& l = h->post;
        TRY(prof.begin(3, -1, 0, 2.0 * fB * L * l.C_in * l.k,
                       2.0 * fB * L * l.C_in * nk + 4.0 * (fB * L + (double)l.ref_w_floats + 1)));
        const int C = l.C_in;
        post::ConvPostLaunch ar; memset(&ar, 0, sizeof(ar));
        for (int j = 0; j < nk && j < 4; ++j) ar.x[j] = cur[j];
        ar.n_in = nk; ar.inv_n = 1.0f / (float)nk;
        ar.w = blob + l.w_off; ar.bias = blob + l.b_off; ar.y = (float*)wav_dev;
        ar.B = B; ar.L = L; ar.C = C; ar.k = l.k; ar.slope = slope;
        if (mean32) {
            // the summing pair left the fp32 mean: conv_post as in the fp32 path (one fp32 input, LeakyReLU in fp32)
            ar.x[0] = mean32; ar.n_in = 1;
            if (!post::conv_post_rows_ok(ar, false)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "conv_post: shape not supported behind the summing pair");
            HIP_TRY(post::launch_conv_post_t<false>(ar, stream));
        } else if (post::conv_post_rows_ok(ar, true)) {
            // 16-byte staging, batch folded into the grid (conv_post.h); same arithmetic as the kernel below, which takes
            // the shapes this one cannot (other channel counts; a single item of 2^31 bytes or more)
            HIP_TRY(post::launch_conv_post_t<true>(ar, stream));
        } else {
            PostLaunch a; memset(&a, 0, sizeof(a));
            for (int j = 0; j < nk; ++j) a.x[j] = cur[j];
            a.n_in = nk; a.inv_n = 1.0f / (float)nk;
            a.w = blob + l.w_off; a.bias = blob + l.b_off; a.y = (float*)wav_dev;
            a.B = B; a.L = L; a.C = C; a.k = l.k; a.slope = slope;
            HIP_TRY(launch_conv_post_bf16(a, stream));
        }
        TRY(prof.end());
    }
    TRY(prof.finish());
    return IRIS_HIFIGAN_OK;
}

}  // namespace iris

// ------------------------------------------------------------------------------------------------
// single-layer entry points (parity tests of the bf16 kernel)
// ------------------------------------------------------------------------------------------------
namespace {
struct DevBytes {
    void* p = nullptr;
    ~DevBytes() { if (p) (void)hipFree(p); }
    hipError_t upload(const void* src, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    }
};
}  // namespace

extern "C" {

int32_t iris_hifigan_op_conv1d_bf16(const void* x_dev, const float* w_host, const float* bias_host,
                                    const void* res_dev, void* y_dev, int32_t B, int32_t L, int32_t C_in,
                                    int32_t C_out, int32_t k, int32_t dilation, int32_t in_act, float slope,
                                    void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    using namespace iris::b16;
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || C_out < 1 || k < 1 || !(k & 1) || dilation < 1 || B > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv1d shape");
    if ((C_in & 7) || (C_out & 3))
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "bf16 conv needs C_in %% 8 == 0 and C_out %% 4 == 0");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<uint16_t> packed(packed_conv1d_halfs(C_in, C_out, k));
    pack_conv1d_bf16(w_host, C_in, C_out, k, packed.data());
    DevBytes wb, bb;
    HIP_TRY(wb.upload(packed.data(), packed.size() * sizeof(uint16_t)));
    HIP_TRY(bb.upload(bias_host, sizeof(float) * C_out));
    Launch a; memset(&a, 0, sizeof(a)); a.out_stride = 1;
    a.p[0].x = x_dev; a.p[0].wp = wb.p; a.p[0].bias = (const float*)bb.p; a.p[0].res = (const uint16_t*)res_dev;
    a.p[0].y = (uint16_t*)y_dev; a.p[0].ks = k; a.p[0].dil = dilation; a.p[0].pad_left = dilation * (k - 1) / 2;
    a.B = B; a.L_in = L; a.L_out = L; a.C_in = C_in; a.C_out = C_out; a.n_idx = L;
    a.in_act = in_act ? IN_ACT_LRELU : IN_ACT_NONE; a.slope = slope;
    HIP_TRY(launch_conv_bf16(a, 1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_mrf_pair_bf16(const void* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                      const float* const* w2_host, const float* const* b2_host, void* const* y_dev,
                                      int32_t n_branches, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                      float slope, void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    using namespace iris::b16;
    if (!x_dev || !w1_host || !b1_host || !w2_host || !b2_host || !y_dev || !k || !dil)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (n_branches < 1 || n_branches > kMaxGroup || B < 1 || L < 1 || C < 1 || B > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad mrf_pair shape");
    hipStream_t stream = (hipStream_t)stream_;
    PairLaunch a; memset(&a, 0, sizeof(a));
    DevBytes w1b[kMaxGroup], w2b[kMaxGroup], b1b[kMaxGroup], b2b[kMaxGroup];
    for (int j = 0; j < n_branches; ++j) {
        if (!x_dev[j] || !w1_host[j] || !b1_host[j] || !w2_host[j] || !b2_host[j] || !y_dev[j])
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL branch argument");
        if (k[j] < 1 || !(k[j] & 1) || dil[j] < 1) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad kernel size / dilation");
        a.p[j].ks = k[j]; a.p[j].dil = dil[j];
    }
    a.B = B; a.L = L; a.C = C; a.slope = slope;
    if (!pair_applicable(a, n_branches))
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "the fused pair kernel takes C = 32, 64 or 128 and windows up to 64 KB");
    for (int j = 0; j < n_branches; ++j) {
        std::vector<uint16_t> packed(packed_conv1d_halfs(C, C, k[j]));
        pack_conv1d_bf16(w1_host[j], C, C, k[j], packed.data());
        HIP_TRY(w1b[j].upload(packed.data(), packed.size() * sizeof(uint16_t)));
        pack_conv1d_bf16(w2_host[j], C, C, k[j], packed.data());
        HIP_TRY(w2b[j].upload(packed.data(), packed.size() * sizeof(uint16_t)));
        HIP_TRY(b1b[j].upload(b1_host[j], sizeof(float) * C));
        HIP_TRY(b2b[j].upload(b2_host[j], sizeof(float) * C));
        a.p[j].x = (const uint16_t*)x_dev[j]; a.p[j].y = (uint16_t*)y_dev[j];
        a.p[j].w1 = w1b[j].p; a.p[j].w2 = w2b[j].p;
        a.p[j].b1 = (const float*)b1b[j].p; a.p[j].b2 = (const float*)b2b[j].p;
    }
    HIP_TRY(launch_pair_bf16(a, n_branches, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_mrf_pair_mean_bf16(const void* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                           const float* const* w2_host, const float* const* b2_host, void* mean_dev,
                                           int32_t mean_f32, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                           float slope, void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    using namespace iris::b16;
    if (!x_dev || !w1_host || !b1_host || !w2_host || !b2_host || !mean_dev || !k || !dil)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C < 1 || B > 65535) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad mrf_pair shape");
    const int nz = 3;
    hipStream_t stream = (hipStream_t)stream_;
    PairLaunch a; memset(&a, 0, sizeof(a));
    DevBytes w1b[kMaxGroup], w2b[kMaxGroup], b1b[kMaxGroup], b2b[kMaxGroup];
    for (int j = 0; j < nz; ++j) {
        if (!x_dev[j] || !w1_host[j] || !b1_host[j] || !w2_host[j] || !b2_host[j])
            return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL branch argument");
        if (k[j] < 1 || !(k[j] & 1) || dil[j] < 1) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad kernel size / dilation");
        if (x_dev[j] == mean_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "the mean must not overwrite an input");
        a.p[j].ks = k[j]; a.p[j].dil = dil[j];
    }
    a.B = B; a.L = L; a.C = C; a.slope = slope;
    if (!pair_sum_applicable(a, nz, mean_f32 != 0))
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "the summing pair kernel takes C = 32 or 64, three branches and windows up to 64 KB");
    for (int j = 0; j < nz; ++j) {
        std::vector<uint16_t> packed(packed_conv1d_halfs(C, C, k[j]));
        pack_conv1d_bf16(w1_host[j], C, C, k[j], packed.data());
        HIP_TRY(w1b[j].upload(packed.data(), packed.size() * sizeof(uint16_t)));
        pack_conv1d_bf16(w2_host[j], C, C, k[j], packed.data());
        HIP_TRY(w2b[j].upload(packed.data(), packed.size() * sizeof(uint16_t)));
        HIP_TRY(b1b[j].upload(b1_host[j], sizeof(float) * C));
        HIP_TRY(b2b[j].upload(b2_host[j], sizeof(float) * C));
        a.p[j].x = (const uint16_t*)x_dev[j]; a.p[j].y = nullptr;
        a.p[j].w1 = w1b[j].p; a.p[j].w2 = w2b[j].p;
        a.p[j].b1 = (const float*)b1b[j].p; a.p[j].b2 = (const float*)b2b[j].p;
    }
    HIP_TRY(launch_pair_bf16_sum(a, mean_dev, mean_f32 != 0, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_conv1d_f32s(const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev,
                                    float* y_dev, int32_t B, int32_t L, int32_t C, int32_t k, int32_t dilation, float slope,
                                    void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C < 1 || k < 1 || !(k & 1) || dilation < 1 || B > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv1d shape");
    s3::Launch a; memset(&a, 0, sizeof(a));
    a.B = B; a.L = L; a.C = C; a.slope = slope;
    if (!s3::applicable(a, 1)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "split-product conv needs C %% 32 == 0");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<uint16_t> packed(2 * s3::packed_plane_halfs(C, C, k));
    s3::pack_conv1d_split(w_host, C, C, k, packed.data());
    DevBytes wb, bb;
    HIP_TRY(wb.upload(packed.data(), packed.size() * sizeof(uint16_t)));
    HIP_TRY(bb.upload(bias_host, sizeof(float) * C));
    a.p[0].x = x_dev; a.p[0].wp = wb.p; a.p[0].bias = (const float*)bb.p; a.p[0].res = res_dev; a.p[0].y = y_dev;
    a.p[0].ks = k; a.p[0].dil = dilation; a.p[0].pad_left = dilation * (k - 1) / 2;
    HIP_TRY(s3::launch(a, 1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_conv_transpose1d_f32s(const float* x_dev, const float* w_host, const float* bias_host, float* y_dev,
                                              int32_t B, int32_t L, int32_t C_in, int32_t C_out, int32_t k, int32_t u,
                                              float slope, void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || C_out < 1 || u < 1 || k < u || ((k - u) & 1) || B > 65535 || u > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv_transpose1d shape");
    ConvLayer l; l.kind = 1; l.C_in = C_in; l.C_out = C_out; l.k = k; l.u = u;
    s3::Launch a;
    fill_ups(a, l, nullptr, nullptr, nullptr, nullptr, B, L, slope);
    if (!s3::applicable(a, u)) return fail(IRIS_HIFIGAN_UNSUPPORTED, "split-product upsample: channel counts not supported");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<uint16_t> packed(2 * s3::packed_convt_plane_halfs(C_in, C_out, k, u));
    s3::pack_convt_split(w_host, C_in, C_out, k, u, packed.data());
    DevBytes wb, bb;
    HIP_TRY(wb.upload(packed.data(), packed.size() * sizeof(uint16_t)));
    HIP_TRY(bb.upload(bias_host, sizeof(float) * C_out));
    fill_ups(a, l, x_dev, wb.p, (const float*)bb.p, y_dev, B, L, slope);
    HIP_TRY(s3::launch(a, u, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

int32_t iris_hifigan_op_conv_transpose1d_bf16(const void* x_dev, const float* w_host, const float* bias_host,
                                              void* y_dev, int32_t B, int32_t L, int32_t C_in, int32_t C_out,
                                              int32_t k, int32_t u, int32_t in_act, float slope, void* stream_) {
    IRIS_ABI_BEGIN
    using namespace iris;
    using namespace iris::b16;
    if (!x_dev || !w_host || !bias_host || !y_dev) return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "NULL argument");
    if (B < 1 || L < 1 || C_in < 1 || C_out < 1 || u < 1 || k < u || ((k - u) & 1) || B > 65535 || u > 65535)
        return fail(IRIS_HIFIGAN_INVALID_ARGUMENT, "bad conv_transpose1d shape");
    if ((C_in & 7) || (C_out & 3))
        return fail(IRIS_HIFIGAN_UNSUPPORTED, "bf16 conv needs C_in %% 8 == 0 and C_out %% 4 == 0");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t phase_halfs = packed_convt_phase_halfs(C_in, C_out, k, u);
    std::vector<uint16_t> packed(phase_halfs * u);
    pack_convt_bf16(w_host, C_in, C_out, k, u, packed.data());
    DevBytes wb, bb;
    HIP_TRY(wb.upload(packed.data(), packed.size() * sizeof(uint16_t)));
    HIP_TRY(bb.upload(bias_host, sizeof(float) * C_out));
    const int taps = convt_taps(k, u);
    Launch a; memset(&a, 0, sizeof(a));
    a.p[0].x = x_dev; a.p[0].wp = wb.p; a.p[0].bias = (const float*)bb.p; a.p[0].res = nullptr;
    a.p[0].y = (uint16_t*)y_dev; a.p[0].ks = taps; a.p[0].dil = 1; a.p[0].pad_left = taps - 1;
    a.B = B; a.L_in = L; a.L_out = L * u; a.C_in = C_in; a.C_out = C_out; a.n_idx = L + taps - 1;
    a.out_stride = u; a.out_off = -(k - u) / 2; a.z_is_phase = 1;
    a.phase_wp_bytes = (unsigned)(phase_halfs * 2);
    a.in_act = in_act ? IN_ACT_LRELU : IN_ACT_NONE; a.slope = slope;
    HIP_TRY(launch_convt_bf16(a, k, u, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return IRIS_HIFIGAN_OK;
    IRIS_ABI_END
}

}  // extern "C"
