"""Generator hyper-parameters, state-dict handling and seeded synthetic weights.

Self-contained (numpy only) on purpose: ``tests/golden/make_golden.py`` loads this file by path
next to the *reference's* ``iris`` package to feed both sides identical weights.

Reference facts restated here (all read from the reference as text):
  * hyper-parameters / constructor defaults: ``src/iris/hifigan_pretrained.py:77-85``,
    ``src/iris/vocoder.py:59-67``;
  * layer inventory and state-dict keys (234 tensors): ``hifigan_pretrained.py:92-121`` with
    ``nn.utils.weight_norm`` on every conv -> ``<layer>.weight_g``, ``<layer>.weight_v``, ``<layer>.bias``;
  * weight-norm: effective ``w = v * (g / ||v||)``, norm over every dim except dim 0
    (dim 0 = C_out for Conv1d, C_in for ConvTranspose1d) -- ``torch.nn.utils.weight_norm`` semantics;
  * checkpoint container variants accepted by the loader: ``hifigan_pretrained.py:168-182``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np

LRELU_SLOPE = 0.1  # hifigan_pretrained.py:66,68,127,139 ; vocoder.py:35,37,114,127


@dataclass(frozen=True)
class GeneratorConfig:
    """Constructor arguments of ``HiFiGANModel`` (hifigan_pretrained.py:77-85)."""

    in_channels: int = 80
    upsample_rates: Tuple[int, ...] = (8, 8, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    pre_kernel_size: int = 7   # hifigan_pretrained.py:93
    post_kernel_size: int = 7  # hifigan_pretrained.py:120
    lrelu_slope: float = LRELU_SLOPE

    def __post_init__(self):
        object.__setattr__(self, "upsample_rates", tuple(int(v) for v in self.upsample_rates))
        object.__setattr__(self, "upsample_kernel_sizes", tuple(int(v) for v in self.upsample_kernel_sizes))
        object.__setattr__(self, "resblock_kernel_sizes", tuple(int(v) for v in self.resblock_kernel_sizes))
        object.__setattr__(self, "resblock_dilation_sizes",
                           tuple(tuple(int(d) for d in ds) for ds in self.resblock_dilation_sizes))
        if len(self.upsample_rates) != len(self.upsample_kernel_sizes):
            raise ValueError("upsample_rates and upsample_kernel_sizes differ in length")
        # the reference zips kernel sizes with dilation lists (hifigan_pretrained.py:115):
        # surplus entries of the longer one are silently dropped
        n = min(len(self.resblock_kernel_sizes), len(self.resblock_dilation_sizes))
        object.__setattr__(self, "resblock_kernel_sizes", self.resblock_kernel_sizes[:n])
        object.__setattr__(self, "resblock_dilation_sizes", self.resblock_dilation_sizes[:n])

    @property
    def num_upsamples(self) -> int:
        return len(self.upsample_rates)

    @property
    def num_kernels(self) -> int:
        # hifigan_pretrained.py:88 uses len(resblock_kernel_sizes)
        return len(self.resblock_kernel_sizes)

    @property
    def hop_length(self) -> int:
        return int(np.prod(self.upsample_rates))

    def stage_channels(self, i: int) -> int:
        """Channels after upsample stage i (hifigan_pretrained.py:102-103,114)."""
        return self.upsample_initial_channel // (2 ** (i + 1))


@dataclass(frozen=True)
class LayerSpec:
    """One weight-normed convolution of the generator, in forward order."""

    name: str            # state-dict prefix, e.g. "resblocks.4.convs1.2"
    kind: str            # "conv" | "convt" | "post"
    c_in: int
    c_out: int
    k: int
    dilation: int = 1
    stride: int = 1

    @property
    def weight_shape(self) -> Tuple[int, int, int]:
        # Conv1d: [C_out, C_in, k]; ConvTranspose1d: [C_in, C_out, k]
        return (self.c_in, self.c_out, self.k) if self.kind == "convt" else (self.c_out, self.c_in, self.k)

    @property
    def fan_in(self) -> int:
        # torch's default init uses weight.size(1) * k as fan-in for both layouts
        return self.weight_shape[1] * self.k


def layer_specs(cfg: GeneratorConfig) -> List[LayerSpec]:
    """All convolutions in the order the C-ABI expects their folded weights
    (include/iris_hifigan.h, iris_hifigan_weight_count): conv_pre, then per stage the upsample
    followed by each ResBlock's convs1[*] then convs2[*], then conv_post."""
    specs: List[LayerSpec] = [LayerSpec("conv_pre", "conv", cfg.in_channels, cfg.upsample_initial_channel,
                                        cfg.pre_kernel_size)]
    ch = cfg.upsample_initial_channel
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        specs.append(LayerSpec(f"ups.{i}", "convt", ch, ch // 2, k, 1, u))
        ch //= 2
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            rb = i * cfg.num_kernels + j  # hifigan_pretrained.py:134
            for m, d in enumerate(dils):
                specs.append(LayerSpec(f"resblocks.{rb}.convs1.{m}", "conv", ch, ch, rk, d))
            for m, _ in enumerate(dils):
                specs.append(LayerSpec(f"resblocks.{rb}.convs2.{m}", "conv", ch, ch, rk, 1))
    specs.append(LayerSpec("conv_post", "post", ch, 1, cfg.post_kernel_size))
    return specs


def state_dict_keys(cfg: GeneratorConfig) -> List[str]:
    keys = []
    for s in layer_specs(cfg):
        keys += [f"{s.name}.bias", f"{s.name}.weight_g", f"{s.name}.weight_v"]
    return keys


# --------------------------------------------------------------------------------------------
# weight-norm folding
# --------------------------------------------------------------------------------------------
def fold_weight_norm(weight_g: np.ndarray, weight_v: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
    """w = v * (g / ||v||_2), the norm taken over all dims except dim 0, in fp32 like
    ``torch._weight_norm`` (used by nn.utils.weight_norm at hifigan_pretrained.py:49,55,92,100,119).
    ``out``: an fp32 array of v's shape to write into (``weight_blob`` folds straight into the blob)."""
    v = np.asarray(weight_v, dtype=np.float32)
    g = np.asarray(weight_g, dtype=np.float32).reshape(v.shape[0], *([1] * (v.ndim - 1)))
    norm = np.sqrt(np.sum(np.square(v, dtype=np.float32), axis=tuple(range(1, v.ndim)), keepdims=True,
                          dtype=np.float32))
    if out is None:
        out = np.empty(v.shape, dtype=np.float32)
    return np.multiply(v, g / norm, out=out)


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    if hasattr(t, "detach"):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def _layer_tensors(s: LayerSpec, state_dict: Mapping[str, object], w_out: Optional[np.ndarray] = None,
                   b_out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Folded weight (reference layout, fp32) and bias of one layer, validated; written into ``w_out`` / ``b_out`` when
    given.  Accepts weight-normed entries (``weight_g``/``weight_v``) or plain ``weight``."""
    if f"{s.name}.weight_v" in state_dict:
        v = _to_numpy(state_dict[f"{s.name}.weight_v"])
        if tuple(v.shape) != s.weight_shape:
            raise ValueError(f"{s.name}: weight shape {tuple(v.shape)} != expected {s.weight_shape}")
        w = fold_weight_norm(_to_numpy(state_dict[f"{s.name}.weight_g"]), v, out=w_out)
    elif f"{s.name}.weight" in state_dict:
        w = np.asarray(_to_numpy(state_dict[f"{s.name}.weight"]), dtype=np.float32)
        if tuple(w.shape) != s.weight_shape:
            raise ValueError(f"{s.name}: weight shape {tuple(w.shape)} != expected {s.weight_shape}")
        if w_out is not None:
            w_out[...] = w
            w = w_out
    else:
        raise KeyError(f"state dict has neither {s.name}.weight_v nor {s.name}.weight")
    b = np.asarray(_to_numpy(state_dict[f"{s.name}.bias"]), dtype=np.float32)
    if b.shape != (s.c_out,):
        raise ValueError(f"{s.name}: bias shape {b.shape} != ({s.c_out},)")
    if b_out is not None:
        b_out[...] = b
        b = b_out
    return w, b


def folded_layers(cfg: GeneratorConfig, state_dict: Mapping[str, object]) -> List[Tuple[LayerSpec, np.ndarray, np.ndarray]]:
    """[(spec, weight fp32 in reference layout, bias fp32)] for every layer."""
    out = []
    for s in layer_specs(cfg):
        w, b = _layer_tensors(s, state_dict)
        out.append((s, np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return out


def weight_blob(cfg: GeneratorConfig, state_dict: Mapping[str, object]) -> np.ndarray:
    """Flat fp32 blob for ``iris_hifigan_create``: weight then bias of every layer, reference layouts.  Every layer is
    folded straight into its place in the blob (one pass over the 13.9 M values: this is part of the cold start of the
    reference's load-then-vocode-once caller, scripts/synthesize.py:197-198)."""
    blob = np.empty(expected_weight_count(cfg), dtype=np.float32)
    off = 0
    for s in layer_specs(cfg):
        n = int(np.prod(s.weight_shape))
        _layer_tensors(s, state_dict, blob[off:off + n].reshape(s.weight_shape), blob[off + n:off + n + s.c_out])
        off += n + s.c_out
    assert off == blob.size
    return blob


def expected_weight_count(cfg: GeneratorConfig) -> int:
    return sum(int(np.prod(s.weight_shape)) + s.c_out for s in layer_specs(cfg))


# --------------------------------------------------------------------------------------------
# checkpoint containers (hifigan_pretrained.py:168-182)
# --------------------------------------------------------------------------------------------
def extract_state_dict(checkpoint) -> Mapping[str, object]:
    """Same selection order as the reference: ["generator"] | ["model"] | ["state_dict"] | the dict."""
    if not isinstance(checkpoint, dict):
        raise ValueError(f"Unexpected checkpoint format: {type(checkpoint)}")
    for key in ("generator", "model", "state_dict"):
        if key in checkpoint:
            return checkpoint[key]
    return checkpoint


# --------------------------------------------------------------------------------------------
# Keras twin layouts (vocoder.py:27-30,82,90,101): no weight-norm, channels-last kernels
# --------------------------------------------------------------------------------------------
def keras_to_reference_layout(spec: LayerSpec, kernel: np.ndarray) -> np.ndarray:
    """Conv1D kernel [k, C_in, C_out] -> [C_out, C_in, k]; Conv1DTranspose kernel
    [k, C_out, C_in] -> [C_in, C_out, k] (SURVEY.md section 8c parameter map)."""
    kernel = np.asarray(kernel, dtype=np.float32)
    return np.ascontiguousarray(kernel.transpose(2, 1, 0))


def reference_to_keras_layout(spec: LayerSpec, weight: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(weight, dtype=np.float32).transpose(2, 1, 0))


# --------------------------------------------------------------------------------------------
# seeded synthetic weights (no checkpoint ships with the reference: .gitignore:63-64)
# --------------------------------------------------------------------------------------------
def seeded_state_dict(cfg: GeneratorConfig = GeneratorConfig(), seed: int = 2024, gain: float = 1.0,
                      post_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic weight-normed state dict (numpy fp32), PCG64(seed).

    ``v`` and ``bias`` ~ U(+-1/sqrt(fan_in)) like torch's default conv init; ``g = ||v|| * U(0.75,1.25) * gain``
    so that folding is not the identity.  ``gain`` scales every layer but conv_post, ``post_gain``
    scales conv_post: the "amplified" set of the tests uses them to push pre-tanh values past +-3.
    Generation order = ``layer_specs`` order, per layer v, g-jitter, bias.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    for s in layer_specs(cfg):
        bound = 1.0 / np.sqrt(s.fan_in)
        v = rng.uniform(-bound, bound, size=s.weight_shape).astype(np.float32)
        jitter = rng.uniform(0.75, 1.25, size=(s.weight_shape[0],)).astype(np.float32)
        bias = rng.uniform(-bound, bound, size=(s.c_out,)).astype(np.float32)
        norm = np.sqrt(np.sum(v.astype(np.float64) ** 2, axis=(1, 2))).astype(np.float32)
        scale = post_gain if s.kind == "post" else gain
        g = (norm * jitter * np.float32(scale)).reshape(-1, 1, 1).astype(np.float32)
        sd[f"{s.name}.bias"] = bias
        sd[f"{s.name}.weight_g"] = g
        sd[f"{s.name}.weight_v"] = v
    return sd


def seeded_mel(seed: int, batch: int, frames: int, n_mels: int = 80, log_mel: bool = False) -> np.ndarray:
    """Synthetic mel: standard normal like the reference's own smoke input
    (test_hifigan_integration.py:49), or log-mel-like clip(N(-5.5,2), log(1e-5), 2) -- the range the
    reference's front-end produces (src/iris/data.py:65)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((batch, n_mels, frames), dtype=np.float32)
    if log_mel:
        x = np.clip(-5.5 + 2.0 * x, np.log(1e-5), 2.0).astype(np.float32)
    return x
