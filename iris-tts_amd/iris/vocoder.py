"""MI355X drop-in for the reference module ``iris.vocoder`` (the Keras/JAX twin of the generator).

Call surface kept from ``/root/reference/src/iris/vocoder.py``:
  * ``create_vocoder(weights_path=None) -> HiFiGANVocoder``                 (:216-226)
  * ``HiFiGANVocoder.infer(mel)`` / ``__call__`` : ``[80,T] -> [256T]``, ``[B,80,T] -> [B,256T]``
    (a batch-1 3-D input stays 2-D), float64 input accepted, float32 host array returned (:177-213)
  * ``HiFiGANVocoder.load_weights / save_weights``, attribute ``.model`` with ``.summary()``
    (:155,167-175; demo_vocoder.py:170)
  * a missing or ``None`` weights path is not an error: random weights, info log (:161-165)

The reference model is channels-last with plain (un-normalised) Keras parameters: ``Conv1D`` kernels
``[k, C_in, C_out]``, ``Conv1DTranspose`` kernels ``[k, C_out, C_in]`` and ``padding='same'``
(:27-30,82,90,101).  ``HiFiGANGenerator`` below keeps its parameters in exactly those layouts and
names; they are transposed to the library's layout when uploaded.  Keras and JAX are NOT used and
not needed: the forward runs in HIP on a gfx950 GPU.

Parity note: keras/jax are not installable in the build environment, so the numerical behaviour of
this twin is pinned through the PyTorch twin's goldens under the parameter map above
 -- the Keras path itself is "parity unpinned".  Keras ``.keras``/``.h5``
weight files need ``h5py``; without it ``load_weights`` accepts the ``.npz`` written by
``save_weights``.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ._engine import GeneratorEngine, require_gpu
from ._weights import (GeneratorConfig, LayerSpec, keras_to_reference_layout, layer_specs)

logger = logging.getLogger(__name__)

__all__ = ["ResBlock", "HiFiGANGenerator", "HiFiGANVocoder", "create_vocoder"]


def _keras_kernel_shape(spec: LayerSpec) -> Tuple[int, int, int]:
    if spec.kind == "convt":
        return (spec.k, spec.c_out, spec.c_in)  # Conv1DTranspose
    return (spec.k, spec.c_in, spec.c_out)      # Conv1D


def _glorot_uniform(rng: np.random.Generator, shape: Tuple[int, int, int], spec: LayerSpec) -> np.ndarray:
    # Keras' default kernel initialiser; fan_in/fan_out include the receptive field
    fan_in, fan_out = spec.c_in * spec.k, spec.c_out * spec.k
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


class ResBlock:
    """Names the parameters of one MRF branch like the reference layer (vocoder.py:13-49)."""

    def __init__(self, channels: int, kernel_size: int = 3, dilations: Tuple[int, ...] = (1, 3, 5)):
        self.channels, self.kernel_size, self.dilations = channels, kernel_size, tuple(dilations)

    def get_config(self) -> dict:
        return {"channels": self.channels, "kernel_size": self.kernel_size, "dilations": self.dilations}


class HiFiGANGenerator:
    """Generator with the constructor of the reference ``keras.Model`` (vocoder.py:59-67).

    ``model(x, training=False)`` takes channels-last ``[batch, time, mel]`` and returns
    ``[batch, time*hop, 1]`` like ``HiFiGANGenerator.call`` (:103-130); numpy or torch in, numpy out.
    """

    def __init__(
        self,
        in_channels: int = 80,
        upsample_rates: Tuple[int, ...] = (8, 8, 2, 2),
        upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4),
        upsample_initial_channel: int = 512,
        resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11),
        resblock_dilations: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5)),
        seed: Optional[int] = None,
        **kwargs,
    ):
        self.in_channels = in_channels
        self.upsample_rates = tuple(upsample_rates)
        self.upsample_kernel_sizes = tuple(upsample_kernel_sizes)
        self.upsample_initial_channel = upsample_initial_channel
        self.resblock_kernel_sizes = tuple(resblock_kernel_sizes)
        self.resblock_dilations = tuple(tuple(d) for d in resblock_dilations)
        self.name = kwargs.get("name", "hi_fi_gan_generator")
        self.config = GeneratorConfig(in_channels, self.upsample_rates, self.upsample_kernel_sizes,
                                      upsample_initial_channel, self.resblock_kernel_sizes,
                                      self.resblock_dilations)
        self.num_kernels = self.config.num_kernels
        self.num_upsamples = self.config.num_upsamples
        self._specs: List[LayerSpec] = layer_specs(self.config)
        self.resblocks = [ResBlock(self.config.stage_channels(i), k, d)
                          for i in range(self.num_upsamples)
                          for k, d in zip(self.resblock_kernel_sizes, self.resblock_dilations)]
        rng = np.random.default_rng(seed)
        # Keras defaults: glorot_uniform kernels, zero biases ("random weights (needs training)", :165)
        self.weights: Dict[str, np.ndarray] = {}
        for s in self._specs:
            self.weights[f"{s.name}.kernel"] = _glorot_uniform(rng, _keras_kernel_shape(s), s)
            self.weights[f"{s.name}.bias"] = np.zeros((s.c_out,), dtype=np.float32)
        self._engine: Optional[GeneratorEngine] = None

    # -- parameters --------------------------------------------------------------------------
    def get_config(self) -> dict:
        return {
            "in_channels": self.in_channels,
            "upsample_rates": self.upsample_rates,
            "upsample_kernel_sizes": self.upsample_kernel_sizes,
            "upsample_initial_channel": self.upsample_initial_channel,
            "resblock_kernel_sizes": self.resblock_kernel_sizes,
            "resblock_dilations": self.resblock_dilations,
        }

    def set_weights_dict(self, weights: Dict[str, np.ndarray]) -> None:
        for s in self._specs:
            for suffix, shape in (("kernel", _keras_kernel_shape(s)), ("bias", (s.c_out,))):
                key = f"{s.name}.{suffix}"
                if key not in weights:
                    raise KeyError(f"weights are missing {key}")
                arr = np.asarray(weights[key], dtype=np.float32)
                if arr.shape != shape:
                    raise ValueError(f"{key}: shape {arr.shape} != expected {shape}")
                self.weights[key] = np.ascontiguousarray(arr)
        self._drop_engine()

    def reference_state_dict(self) -> Dict[str, np.ndarray]:
        """Parameters in the PyTorch twin's layouts (plain ``weight``/``bias``: the Keras twin has
        no weight-norm)."""
        sd = {}
        for s in self._specs:
            sd[f"{s.name}.weight"] = keras_to_reference_layout(s, self.weights[f"{s.name}.kernel"])
            sd[f"{s.name}.bias"] = self.weights[f"{s.name}.bias"]
        return sd

    def count_params(self) -> int:
        return int(sum(v.size for v in self.weights.values()))

    def summary(self, print_fn=print) -> None:
        print_fn(f'Model: "{self.name}" (MI355X HIP build)')
        print_fn(f"{'Layer':<28}{'Kernel shape':<22}{'Params':>12}")
        for s in self._specs:
            kshape = _keras_kernel_shape(s)
            n = int(np.prod(kshape)) + s.c_out
            kind = {"conv": "Conv1D", "convt": "Conv1DTranspose", "post": "Conv1D+tanh"}[s.kind]
            print_fn(f"{s.name + ' (' + kind + ')':<28}{str(kshape):<22}{n:>12,}")
        print_fn(f"Total params: {self.count_params():,}")

    def save_weights(self, weights_path: str) -> None:
        path = Path(weights_path)
        if path.suffix in (".h5", ".keras"):
            raise NotImplementedError(
                "writing Keras .h5/.keras files needs h5py/keras, which this build does not use; "
                "save to a .npz path instead")
        np.savez(str(path), **self.weights)

    def load_weights(self, weights_path: str) -> None:
        path = Path(weights_path)
        if path.suffix in (".h5", ".keras"):
            raise NotImplementedError(
                f"{path.name}: reading Keras .h5/.keras weight files needs h5py, which is not available; "
                "convert the file to the .npz written by save_weights")
        with np.load(str(path), allow_pickle=False) as data:
            self.set_weights_dict({k: data[k] for k in data.files})

    # -- execution ---------------------------------------------------------------------------
    def _drop_engine(self) -> None:
        if self._engine is not None:
            self._engine.close()
        self._engine = None

    def invalidate(self) -> None:
        """Drops the packed device copy of the weights.  ``set_weights_dict`` / ``load_weights`` do this
        themselves; call it after editing the arrays in ``self.weights`` in place (numpy arrays carry no
        version counter, so such edits cannot be noticed): the next call repacks the current values."""
        self._drop_engine()

    def engine(self) -> GeneratorEngine:
        if self._engine is None:
            self._engine = GeneratorEngine(self.config, self.reference_state_dict(), require_gpu())
        return self._engine

    def __call__(self, x, training: bool = False) -> np.ndarray:
        """[batch, time, mel_channels] -> [batch, time*hop, 1] (vocoder.py:103-130)."""
        if training:
            raise NotImplementedError("the MI355X build is inference-only")
        eng = self.engine()
        if isinstance(x, torch.Tensor):
            mel_bt_c = x.to(device=eng.device, dtype=torch.float32)
        else:
            mel_bt_c = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(eng.device)
        if mel_bt_c.dim() != 3:
            raise ValueError(f"expected [batch, time, mel_channels], got {tuple(mel_bt_c.shape)}")
        # the library takes the mel channels-first; this transpose is a view + one small copy
        wav = eng.forward(mel_bt_c.transpose(1, 2).contiguous())
        return wav.cpu().numpy()[..., np.newaxis]

    call = __call__


class HiFiGANVocoder:
    """High-level interface (reference vocoder.py:145-213)."""

    def __init__(self, weights_path: Optional[str] = None):
        self.model = HiFiGANGenerator()
        if weights_path and Path(weights_path).exists():
            self.load_weights(weights_path)
        else:
            logger.info("Initialized HiFiGAN with random weights (needs training)")

    def load_weights(self, weights_path: str):
        self.model.load_weights(weights_path)
        logger.info(f"Loaded weights from {weights_path}")

    def save_weights(self, weights_path: str):
        self.model.save_weights(weights_path)
        logger.info(f"Saved weights to {weights_path}")

    def infer(self, mel: np.ndarray) -> np.ndarray:
        """[mel_channels, time] -> [samples];  [batch, mel_channels, time] -> [batch, samples]."""
        mel = np.asarray(mel)
        squeeze_batch = False
        if mel.ndim == 2:
            mel = mel.T[np.newaxis, ...]            # -> [1, time, mel]
            squeeze_batch = True
        elif mel.ndim == 3:
            mel = np.transpose(mel, (0, 2, 1))      # -> [batch, time, mel]
        audio = np.array(self.model(mel, training=False))
        audio = audio[..., 0]
        if squeeze_batch:
            audio = audio[0]
        return audio

    def __call__(self, mel: np.ndarray) -> np.ndarray:
        return self.infer(mel)


def create_vocoder(weights_path: Optional[str] = None) -> HiFiGANVocoder:
    """Factory with the reference's signature (vocoder.py:216-226)."""
    return HiFiGANVocoder(weights_path=weights_path)
