/*
 * hifigan_oracle.c -- plain-C CPU restatement of the reference HiFiGAN generator forward.
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/ (and built by __graft_entry__.build());
 * never by the product in iris-tts_amd/.
 *
 * Follows, read as text:
 *   HiFiGANModel.forward   /root/reference/src/iris/hifigan_pretrained.py:123-143
 *   ResBlock.forward       /root/reference/src/iris/hifigan_pretrained.py:64-71
 *   Conv1d padding         int((k*d - d)/2)                     hifigan_pretrained.py:61-62
 *   ConvTranspose1d        stride u, padding (k-u)//2           hifigan_pretrained.py:100-108
 * Layout: channels-first [B][C][L] like the reference; weights in the reference layouts, folded
 * (weight-norm applied), concatenated in the order documented in include/iris_hifigan.h.
 * Every dot product is accumulated in double and rounded once to float: this oracle is at least as
 * accurate as the fp32 reference, so |hip - oracle| bounds |hip - reference| up to ~1e-6.
 * Pinned against the reference's own outputs by tests/test_oracle_golden.py (tests/golden/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX 8

typedef struct orc_config {
    int32_t in_channels, upsample_initial_channel, num_upsamples;
    int32_t upsample_rates[ORC_MAX], upsample_kernel_sizes[ORC_MAX];
    int32_t num_kernels, resblock_kernel_sizes[ORC_MAX], num_dilations[ORC_MAX];
    int32_t resblock_dilations[ORC_MAX][ORC_MAX];
    int32_t pre_kernel_size, post_kernel_size;
    float lrelu_slope;
} orc_config;

static void lrelu(float* x, size_t n, float slope) {
    for (size_t i = 0; i < n; ++i)
        if (!(x[i] > 0.f)) x[i] = x[i] * slope;
}

/* y[b,co,t] = bias[co] + sum_{ci,kap} w[co,ci,kap] * x[b,ci,t + (kap-(k-1)/2)*d]   (zero padded) */
void orc_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co,
                int L, int k, int d) {
    const int pad = (k * d - d) / 2;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int t = 0; t < L; ++t) {
                double acc = bias[co];
                for (int ci = 0; ci < Ci; ++ci) {
                    const float* xr = x + ((size_t)b * Ci + ci) * L;
                    const float* wr = w + ((size_t)co * Ci + ci) * k;
                    for (int kap = 0; kap < k; ++kap) {
                        const int s = t + kap * d - pad;
                        if (s >= 0 && s < L) acc += (double)wr[kap] * (double)xr[s];
                    }
                }
                y[((size_t)b * Co + co) * L + t] = (float)acc;
            }
}

/* y[b,co,i*u - p + kap] += x[b,ci,i] * w[ci,co,kap];  L_out = (L-1)*u - 2p + k */
void orc_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int Ci,
                          int Co, int L, int k, int u, int p) {
    const int Lo = (L - 1) * u - 2 * p + k;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int o = 0; o < Lo; ++o) {
                double acc = bias[co];
                for (int kap = 0; kap < k; ++kap) {
                    const int num = o + p - kap;
                    if (num < 0 || num % u) continue;
                    const int i = num / u;
                    if (i >= L) continue;
                    for (int ci = 0; ci < Ci; ++ci)
                        acc += (double)x[((size_t)b * Ci + ci) * L + i] * (double)w[((size_t)ci * Co + co) * k + kap];
                }
                y[((size_t)b * Co + co) * Lo + o] = (float)acc;
            }
}

/* mel [B][in][T] -> wav [B][hop*T]; weights = folded blob (include/iris_hifigan.h order). 0 on success. */
int orc_generator_forward(const orc_config* c, const float* weights, const float* mel, int B, int T,
                          float* wav) {
    const float* wp = weights;
    int ch = c->upsample_initial_channel, L = T;
    float* x = (float*)malloc(sizeof(float) * (size_t)B * ch * L);
    if (!x) return 1;
    orc_conv1d(mel, wp, wp + (size_t)ch * c->in_channels * c->pre_kernel_size, x, B, c->in_channels, ch, L,
               c->pre_kernel_size, 1);
    wp += (size_t)ch * c->in_channels * c->pre_kernel_size + ch;
    for (int i = 0; i < c->num_upsamples; ++i) {
        const int u = c->upsample_rates[i], k = c->upsample_kernel_sizes[i];
        const int co = ch / 2, Lo = L * u;
        lrelu(x, (size_t)B * ch * L, c->lrelu_slope);
        float* up = (float*)malloc(sizeof(float) * (size_t)B * co * Lo);
        if (!up) { free(x); return 1; }
        orc_conv_transpose1d(x, wp, wp + (size_t)ch * co * k, up, B, ch, co, L, k, u, (k - u) / 2);
        wp += (size_t)ch * co * k + co;
        free(x);
        ch = co; L = Lo;
        const size_t n = (size_t)B * ch * L;
        float* xs = (float*)calloc(n, sizeof(float));
        float* r = (float*)malloc(sizeof(float) * n);
        float* t1 = (float*)malloc(sizeof(float) * n);
        float* t2 = (float*)malloc(sizeof(float) * n);
        if (!xs || !r || !t1 || !t2) return 1;
        for (int j = 0; j < c->num_kernels; ++j) {
            const int rk = c->resblock_kernel_sizes[j], nd = c->num_dilations[j];
            const size_t wsz = (size_t)ch * ch * rk + ch;   /* one conv: weight + bias */
            const float* w1 = wp;                            /* convs1[0..nd) then convs2[0..nd) */
            const float* w2 = wp + wsz * nd;
            memcpy(r, up, sizeof(float) * n);
            for (int m = 0; m < nd; ++m) {
                memcpy(t1, r, sizeof(float) * n);
                lrelu(t1, n, c->lrelu_slope);
                orc_conv1d(t1, w1 + wsz * m, w1 + wsz * m + (size_t)ch * ch * rk, t2, B, ch, ch, L, rk,
                           c->resblock_dilations[j][m]);
                lrelu(t2, n, c->lrelu_slope);
                orc_conv1d(t2, w2 + wsz * m, w2 + wsz * m + (size_t)ch * ch * rk, t1, B, ch, ch, L, rk, 1);
                for (size_t e = 0; e < n; ++e) r[e] = t1[e] + r[e];
            }
            if (j == 0) memcpy(xs, r, sizeof(float) * n);
            else for (size_t e = 0; e < n; ++e) xs[e] += r[e];
            wp += wsz * nd * 2;
        }
        for (size_t e = 0; e < n; ++e) xs[e] = xs[e] / (float)c->num_kernels;
        free(r); free(t1); free(t2); free(up);
        x = xs;
    }
    lrelu(x, (size_t)B * ch * L, c->lrelu_slope);
    orc_conv1d(x, wp, wp + (size_t)ch * c->post_kernel_size, wav, B, ch, 1, L, c->post_kernel_size, 1);
    for (size_t e = 0; e < (size_t)B * L; ++e) wav[e] = tanhf(wav[e]);
    free(x);
    return 0;
}
