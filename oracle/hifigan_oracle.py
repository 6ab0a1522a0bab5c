"""CPU restatement of the reference HiFiGAN generator forward.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this file; the product (``iris-tts_amd/``) never does and has no CPU path.

What is restated, and from where (read as text from /root/reference):
  * ``generator_forward_*``     HiFiGANModel.forward          src/iris/hifigan_pretrained.py:123-143
                                (== HiFiGANGenerator.call      src/iris/vocoder.py:103-130)
  * ``resblock_*``              ResBlock.forward               hifigan_pretrained.py:64-71 (vocoder.py:33-40)
  * ``conv1d_np``               nn.Conv1d as configured at     hifigan_pretrained.py:50-57,92-94,119-121
                                padding = int((k*d - d)/2)      hifigan_pretrained.py:61-62
  * ``conv_transpose1d_np``     nn.ConvTranspose1d(k, stride=u, padding=(k-u)//2)   hifigan_pretrained.py:100-108
  * ``fold_weight_norm_torch``  nn.utils.weight_norm           hifigan_pretrained.py:49,55,92,100,119
  * ``wrapper_shapes``          squeeze rules of __call__/infer_hifigan/infer
                                hifigan_pretrained.py:222-240,314-315 ; vocoder.py:191-207

Two independent implementations are kept on purpose:
  * ``*_np``    explicit index formulas in numpy, float64 accumulation (slow, for small cases);
  * ``*_torch`` torch.nn.functional on folded fp32 weights, channels-first, the arithmetic the
                reference itself executes (ATen CPU kernels) -- used at full sizes and as the CPU
                baseline timed by bench.py.

Pinning: the reference ships no test vectors for this path, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports the reference's own ``iris.hifigan_pretrained``) and
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks both implementations
against them.  The Keras/JAX twin cannot be executed anywhere in this pipeline: for it the oracle
is the same function under the documented parameter map -- "parity unpinned" for that twin.
``generator_forward_bf16`` restates the bf16-storage variant (BASELINE.json configs[2]); the reference
has no bf16 path, so that function is "parity unpinned" too: it defines the variant's rounding points and
tests report its distance to the pinned fp32 functions.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

LRELU_SLOPE = 0.1


# ------------------------------------------------------------------------------------------------
# numpy implementation (explicit formulas, fp64 accumulation)
# ------------------------------------------------------------------------------------------------
def lrelu_np(x: np.ndarray, slope: float = LRELU_SLOPE) -> np.ndarray:
    return np.where(x > 0, x, x * np.asarray(slope, dtype=x.dtype))


def conv1d_np(x: np.ndarray, w: np.ndarray, b: np.ndarray, dilation: int = 1) -> np.ndarray:
    """y[n,co,t] = b[co] + sum_{ci,kap} w[co,ci,kap] * x[n,ci,t + (kap - (k-1)/2)*d], zero padded.
    x [N,Ci,L], w [Co,Ci,k] (odd k).  fp64 accumulation, fp32 result."""
    n, ci, L = x.shape
    co, ci2, k = w.shape
    assert ci == ci2 and k % 2 == 1
    pad = int((k * dilation - dilation) / 2)
    xp = np.zeros((n, ci, L + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + L] = x
    y = np.zeros((n, co, L), dtype=np.float64)
    w64 = w.astype(np.float64)
    for kap in range(k):
        seg = xp[:, :, kap * dilation: kap * dilation + L]          # x[t + kap*d - pad]
        y += np.einsum("oc,ncl->nol", w64[:, :, kap], seg)
    y += b.astype(np.float64)[None, :, None]
    return y.astype(np.float32)


def conv_transpose1d_np(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, padding: int) -> np.ndarray:
    """y[n,co,i*u - p + kap] += x[n,ci,i] * w[ci,co,kap];  L_out = (L-1)*u - 2p + k.
    x [N,Ci,L], w [Ci,Co,k]."""
    n, ci, L = x.shape
    ci2, co, k = w.shape
    assert ci == ci2
    L_out = (L - 1) * stride - 2 * padding + k
    full = np.zeros((n, co, (L - 1) * stride + k), dtype=np.float64)
    w64 = w.astype(np.float64)
    x64 = x.astype(np.float64)
    for kap in range(k):
        contrib = np.einsum("co,ncl->nol", w64[:, :, kap], x64)      # [n, co, L]
        full[:, :, kap: kap + (L - 1) * stride + 1: stride] += contrib
    y = full[:, :, padding: padding + L_out] + b.astype(np.float64)[None, :, None]
    return y.astype(np.float32)


def fold_weight_norm_np(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """w = v * g / ||v||, norm over all dims but 0, in fp64 then rounded (an independent check of the
    product's fp32 fold)."""
    v64 = v.astype(np.float64)
    norm = np.sqrt((v64 ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    return (v64 * (g.astype(np.float64).reshape(norm.shape) / norm)).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# layer bookkeeping shared by both implementations
# ------------------------------------------------------------------------------------------------
class OracleConfig:
    """Hyper-parameters, defaults = hifigan_pretrained.py:77-85."""

    def __init__(self, in_channels=80, upsample_rates=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                 upsample_initial_channel=512, resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5))):
        self.in_channels = in_channels
        self.upsample_rates = tuple(upsample_rates)
        self.upsample_kernel_sizes = tuple(upsample_kernel_sizes)
        self.upsample_initial_channel = upsample_initial_channel
        self.resblock_kernel_sizes = tuple(resblock_kernel_sizes)
        self.resblock_dilation_sizes = tuple(tuple(d) for d in resblock_dilation_sizes)
        self.num_kernels = len(self.resblock_kernel_sizes)
        self.num_upsamples = len(self.upsample_rates)

    @property
    def hop_length(self) -> int:
        return int(np.prod(self.upsample_rates))


def fold_state_dict(sd: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """{'<layer>.weight', '<layer>.bias'} from a weight-normed (or already plain) state dict."""
    out: Dict[str, np.ndarray] = {}
    for key, val in sd.items():
        val = np.asarray(val.detach().cpu().numpy() if hasattr(val, "detach") else val)
        if key.endswith(".weight_v"):
            name = key[: -len(".weight_v")]
            g = sd[name + ".weight_g"]
            g = np.asarray(g.detach().cpu().numpy() if hasattr(g, "detach") else g)
            out[name + ".weight"] = fold_weight_norm_np(g, val)
        elif key.endswith(".weight") or key.endswith(".bias"):
            out[key] = val.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------
# generator forward, numpy
# ------------------------------------------------------------------------------------------------
def resblock_np(x: np.ndarray, folded: Mapping[str, np.ndarray], prefix: str, dilations: Sequence[int],
                slope: float = LRELU_SLOPE) -> np.ndarray:
    for m, d in enumerate(dilations):
        xt = lrelu_np(x, slope)
        xt = conv1d_np(xt, folded[f"{prefix}.convs1.{m}.weight"], folded[f"{prefix}.convs1.{m}.bias"], d)
        xt = lrelu_np(xt, slope)
        xt = conv1d_np(xt, folded[f"{prefix}.convs2.{m}.weight"], folded[f"{prefix}.convs2.{m}.bias"], 1)
        x = xt + x
    return x


def generator_forward_np(folded: Mapping[str, np.ndarray], mel: np.ndarray, cfg: Optional[OracleConfig] = None,
                         taps: Optional[dict] = None, slope: float = LRELU_SLOPE) -> np.ndarray:
    """mel [B, in_channels, T] -> [B, 1, hop*T].  ``taps`` (a dict) receives intermediates."""
    cfg = cfg or OracleConfig()
    x = conv1d_np(np.asarray(mel, dtype=np.float32), folded["conv_pre.weight"], folded["conv_pre.bias"], 1)
    if taps is not None:
        taps["conv_pre"] = x
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = lrelu_np(x, slope)
        x = conv_transpose1d_np(x, folded[f"ups.{i}.weight"], folded[f"ups.{i}.bias"], u, (k - u) // 2)
        if taps is not None:
            taps[f"ups.{i}"] = x
        xs = None
        for j in range(cfg.num_kernels):
            r = resblock_np(x, folded, f"resblocks.{i * cfg.num_kernels + j}", cfg.resblock_dilation_sizes[j], slope)
            xs = r if xs is None else xs + r
        x = (xs / np.float32(cfg.num_kernels)).astype(np.float32)
        if taps is not None:
            taps[f"mrf.{i}"] = x
    x = lrelu_np(x, slope)
    x = conv1d_np(x, folded["conv_post.weight"], folded["conv_post.bias"], 1)
    return np.tanh(x).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# generator forward, torch.nn.functional (what the reference's modules execute)
# ------------------------------------------------------------------------------------------------
def fold_weight_norm_torch(g, v):
    import torch

    return torch._weight_norm(torch.as_tensor(v), torch.as_tensor(g), 0)


def to_torch_folded(sd: Mapping[str, object]) -> Dict[str, "object"]:
    """Folds with torch's own ``_weight_norm`` (fp32), like the reference's parametrised modules."""
    import torch

    out = {}
    for key, val in sd.items():
        t = torch.as_tensor(np.asarray(val)) if not hasattr(val, "detach") else val.detach().cpu()
        if key.endswith(".weight_v"):
            name = key[: -len(".weight_v")]
            g = sd[name + ".weight_g"]
            g = torch.as_tensor(np.asarray(g)) if not hasattr(g, "detach") else g.detach().cpu()
            out[name + ".weight"] = torch._weight_norm(t.float(), g.float(), 0).contiguous()
        elif key.endswith(".weight") or key.endswith(".bias"):
            out[key] = t.float().contiguous()
    return out


def generator_forward_torch(folded_t: Mapping[str, object], mel, cfg: Optional[OracleConfig] = None,
                            taps: Optional[dict] = None, slope: float = LRELU_SLOPE):
    """torch fp32, channels-first; mel [B, in_channels, T] tensor -> [B, 1, hop*T] tensor."""
    import torch
    import torch.nn.functional as F

    cfg = cfg or OracleConfig()
    with torch.no_grad():
        x = torch.as_tensor(mel).float()
        k = folded_t["conv_pre.weight"].shape[-1]
        x = F.conv1d(x, folded_t["conv_pre.weight"], folded_t["conv_pre.bias"], padding=(k - 1) // 2)
        if taps is not None:
            taps["conv_pre"] = x
        for i, (u, ku) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            x = F.leaky_relu(x, slope)
            x = F.conv_transpose1d(x, folded_t[f"ups.{i}.weight"], folded_t[f"ups.{i}.bias"], stride=u,
                                   padding=(ku - u) // 2)
            if taps is not None:
                taps[f"ups.{i}"] = x
            xs = None
            for j in range(cfg.num_kernels):
                p = f"resblocks.{i * cfg.num_kernels + j}"
                r = x
                for m, d in enumerate(cfg.resblock_dilation_sizes[j]):
                    w1, b1 = folded_t[f"{p}.convs1.{m}.weight"], folded_t[f"{p}.convs1.{m}.bias"]
                    w2, b2 = folded_t[f"{p}.convs2.{m}.weight"], folded_t[f"{p}.convs2.{m}.bias"]
                    kk = w1.shape[-1]
                    xt = F.leaky_relu(r, slope)
                    xt = F.conv1d(xt, w1, b1, dilation=d, padding=int((kk * d - d) / 2))
                    xt = F.leaky_relu(xt, slope)
                    xt = F.conv1d(xt, w2, b2, padding=int((kk - 1) / 2))
                    r = xt + r
                xs = r if xs is None else xs + r
            x = xs / cfg.num_kernels
            if taps is not None:
                taps[f"mrf.{i}"] = x
        x = F.leaky_relu(x, slope)
        kp = folded_t["conv_post.weight"].shape[-1]
        x = F.conv1d(x, folded_t["conv_post.weight"], folded_t["conv_post.bias"], padding=(kp - 1) // 2)
        return torch.tanh(x)


def generator_forward_bf16(folded_t: Mapping[str, object], mel, cfg: Optional[OracleConfig] = None,
                           taps: Optional[dict] = None, slope: float = LRELU_SLOPE):
    """The generator with bf16 STORAGE (BASELINE.json configs[2]) restated on the CPU: same dataflow as
    ``generator_forward_torch`` (reference src/iris/hifigan_pretrained.py:123-143), with a round-to-nearest-even
    to bf16 wherever the MI355X bf16 path stores a value or feeds the matrix cores:
      * weights (not biases) are rounded once;
      * the input of every conv except conv_post is rounded after its activation (mel, LeakyReLU(x),
        LeakyReLU(mean of the MRF branches));
      * every conv output is rounded after bias (and residual) have been added in fp32;
      * the MRF mean is ((y0 + y1) + y2) * fp32(1/3) in fp32; conv_post and tanh are fp32 on the unrounded
        LeakyReLU of that mean.
    Products are exact in fp32 (bf16 x bf16) and accumulated in fp32, as v_mfma_f32_32x32x16_bf16 does, up
    to summation order.  The reference has no bf16 path: this function is the definition of the variant, and
    its distance to the fp32 generator is what tests report ("bf16 tolerance unpinned by the reference").
    mel [B, in_channels, T] -> [B, 1, hop*T] fp32 tensor."""
    import torch
    import torch.nn.functional as F

    cfg = cfg or OracleConfig()

    def r16(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def wq(name):
        return r16(folded_t[name + ".weight"]), folded_t[name + ".bias"]

    with torch.no_grad():
        x = r16(torch.as_tensor(mel).float())
        w, b = wq("conv_pre")
        k = w.shape[-1]
        x = r16(F.conv1d(x, w, b, padding=(k - 1) // 2))
        if taps is not None:
            taps["conv_pre"] = x
        inv_n = torch.tensor(1.0 / cfg.num_kernels, dtype=torch.float32)
        for i, (u, ku) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            w, b = wq(f"ups.{i}")
            x = r16(F.conv_transpose1d(r16(F.leaky_relu(x, slope)), w, b, stride=u, padding=(ku - u) // 2))
            if taps is not None:
                taps[f"ups.{i}"] = x
            xs = None
            for j in range(cfg.num_kernels):
                p = f"resblocks.{i * cfg.num_kernels + j}"
                r = x
                for m, d in enumerate(cfg.resblock_dilation_sizes[j]):
                    w1, b1 = wq(f"{p}.convs1.{m}")
                    w2, b2 = wq(f"{p}.convs2.{m}")
                    kk = w1.shape[-1]
                    xt = r16(F.conv1d(r16(F.leaky_relu(r, slope)), w1, b1, dilation=d, padding=int((kk * d - d) / 2)))
                    r = r16(F.conv1d(r16(F.leaky_relu(xt, slope)), w2, b2, padding=int((kk - 1) / 2)) + r)
                xs = r if xs is None else xs + r
            x = xs * inv_n                      # fp32, unrounded: rounded where the next conv stages it
            if taps is not None:
                taps[f"mrf.{i}"] = x
        x = F.leaky_relu(x, slope)
        kp = folded_t["conv_post.weight"].shape[-1]
        x = F.conv1d(x, folded_t["conv_post.weight"], folded_t["conv_post.bias"], padding=(kp - 1) // 2)
        return torch.tanh(x)


# ------------------------------------------------------------------------------------------------
# host wrapper semantics
# ------------------------------------------------------------------------------------------------
def wrapper_shapes(entry: str, mel_shape: Tuple[int, ...], hop: int = 256) -> Tuple[int, ...]:
    """Output shape rules of the reference entry points.
    entry: 'generator_call' (HiFiGANGenerator.__call__, :222-240), 'infer_hifigan' (:310-317),
           'vocoder_infer' (HiFiGANVocoder.infer, vocoder.py:191-207)."""
    if len(mel_shape) == 2:
        return (mel_shape[1] * hop,)
    b, _, t = mel_shape
    if entry == "infer_hifigan" and b == 1:
        return (t * hop,)
    return (b, t * hop)
