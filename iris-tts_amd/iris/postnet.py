"""MI355X drop-in for the reference module ``iris.postnet`` (Tacotron2-style PostNet, Keras).

Call surface kept from ``/root/reference/src/iris/postnet.py``: ``PostNet(n_mels, num_layers=4,
channels=256, kernel_size=5, dropout=0.5, name=None)`` (:16-46) and ``postnet(mels_bt_f,
training=False) -> [B, n_mels, T]`` = input + residual (:48-67); ``scripts/synthesize.py:152-166``
builds it with ``num_layers=3, channels=256, kernel_size=5`` and applies it right before the vocoder.

Parameters are kept in the Keras layouts and names (``Conv1D`` kernel ``[k, C_in, C_out]``, bias;
``BatchNormalization`` gamma, beta, moving_mean, moving_variance, epsilon 1e-3 = the Keras default).
At upload the inference BatchNorm is folded into the convolution, so a layer is one launch of the MFMA
conv kernel with a tanh epilogue (``iris-tts_amd/csrc/postnet.h``).  Inference only (dropout is the
identity); ``training=True`` raises.

Parity note: the reference PostNet exists only in Keras/JAX, which cannot run in this pipeline, and
the reference ships no vectors for it: the device path is checked against ``oracle/postnet_oracle.py``
(a numpy restatement of postnet.py:48-67) -- "parity unpinned".  ``.weights.h5`` files need h5py;
``load_weights``/``save_weights`` use ``.npz``.
"""
from __future__ import annotations

import ctypes
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from . import _native
from ._engine import require_gpu

BN_EPSILON = 1e-3  # keras.layers.BatchNormalization default


def fold_batchnorm(kernel: np.ndarray, bias: np.ndarray, gamma: np.ndarray, beta: np.ndarray,
                   mean: np.ndarray, var: np.ndarray, eps: float = BN_EPSILON):
    """Conv1D kernel [k, C_in, C_out] + inference BatchNorm -> (weight [C_out, C_in, k], bias [C_out]).
    BN(y) = gamma * (y - mean) / sqrt(var + eps) + beta, applied per output channel."""
    scale = (gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps))
    w = kernel.astype(np.float64).transpose(2, 1, 0) * scale[:, None, None]
    b = (bias.astype(np.float64) - mean.astype(np.float64)) * scale + beta.astype(np.float64)
    return np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)


class PostNet:
    def __init__(self, n_mels: int, num_layers: int = 4, channels: int = 256, kernel_size: int = 5,
                 dropout: float = 0.5, name: Optional[str] = None, seed: Optional[int] = None):
        assert num_layers >= 2, "PostNet needs at least 2 layers"
        self.n_mels, self.num_layers, self.channels = n_mels, num_layers, channels
        self.kernel_size, self.dropout_rate, self.name = kernel_size, dropout, name or "post_net"
        rng = np.random.default_rng(seed)
        self.weights: Dict[str, np.ndarray] = {}
        for i in range(num_layers):
            c_in = n_mels if i == 0 else channels
            c_out = n_mels if i == num_layers - 1 else channels
            limit = np.sqrt(6.0 / ((c_in + c_out) * kernel_size))            # glorot_uniform
            p = self._prefix(i)
            self.weights[f"{p}.kernel"] = rng.uniform(-limit, limit, (kernel_size, c_in, c_out)).astype(np.float32)
            self.weights[f"{p}.bias"] = np.zeros(c_out, np.float32)
            self.weights[f"{p}.gamma"] = np.ones(c_out, np.float32)
            self.weights[f"{p}.beta"] = np.zeros(c_out, np.float32)
            self.weights[f"{p}.moving_mean"] = np.zeros(c_out, np.float32)
            self.weights[f"{p}.moving_variance"] = np.ones(c_out, np.float32)
        self._handle = None
        self._workspace = None
        self._device = None

    def _prefix(self, i: int) -> str:
        return "conv_out" if i == self.num_layers - 1 else f"convs.{i}"

    def get_config(self) -> dict:
        return {"n_mels": self.n_mels, "num_layers": self.num_layers, "channels": self.channels,
                "kernel_size": self.kernel_size, "dropout": self.dropout_rate}

    # -- parameters --------------------------------------------------------------------------
    def set_weights_dict(self, weights: Dict[str, np.ndarray]) -> None:
        for key, cur in self.weights.items():
            if key not in weights:
                raise KeyError(f"weights are missing {key}")
            arr = np.asarray(weights[key], dtype=np.float32)
            if arr.shape != cur.shape:
                raise ValueError(f"{key}: shape {arr.shape} != expected {cur.shape}")
            self.weights[key] = np.ascontiguousarray(arr)
        self._drop()

    def save_weights(self, path: str) -> None:
        if Path(path).suffix in (".h5", ".keras"):
            raise NotImplementedError("Keras .h5/.keras files need h5py, which this build does not use; save to .npz")
        np.savez(str(path), **self.weights)

    def load_weights(self, path: str) -> None:
        if Path(path).suffix in (".h5", ".keras"):
            raise NotImplementedError(f"{Path(path).name}: reading Keras weight files needs h5py, which is not available")
        with np.load(str(path), allow_pickle=False) as data:
            self.set_weights_dict({k: data[k] for k in data.files})

    def folded_blob(self) -> np.ndarray:
        parts = []
        for i in range(self.num_layers):
            p = self._prefix(i)
            w, b = fold_batchnorm(*(self.weights[f"{p}.{n}"] for n in
                                    ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance")))
            parts += [w.ravel(), b.ravel()]
        return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)

    # -- execution ---------------------------------------------------------------------------
    def _drop(self) -> None:
        if self._handle is not None:
            _native.load().iris_postnet_destroy(self._handle)
        self._handle = None
        self._workspace = None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def _ensure(self):
        if self._handle is None:
            lib = _native.load()
            self._device = require_gpu()
            blob = self.folded_blob()
            h = ctypes.c_void_p()
            with torch.cuda.device(self._device):
                _native.check("iris_postnet_create", lib.iris_postnet_create(
                    self.n_mels, self.num_layers, self.channels, self.kernel_size,
                    blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_uint64(blob.size), ctypes.byref(h)))
            self._handle = h
        return _native.load()

    def forward_device(self, mel: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, T] fp32 device tensor -> refined mel, same shape (asynchronous on the current stream)."""
        lib = self._ensure()
        if mel.dim() != 3 or mel.shape[1] != self.n_mels:
            raise ValueError(f"expected mel [B, {self.n_mels}, T], got {tuple(mel.shape)}")
        mel = mel.to(device=self._device, dtype=torch.float32).contiguous()
        B, _, T = mel.shape
        out = torch.empty_like(mel)
        if B == 0 or T == 0:
            return out
        n = ctypes.c_uint64()
        _native.check("iris_postnet_workspace_bytes", lib.iris_postnet_workspace_bytes(self._handle, B, T, ctypes.byref(n)))
        if self._workspace is None or self._workspace.numel() < n.value:
            self._workspace = torch.empty(max(int(n.value), 256), dtype=torch.uint8, device=self._device)
        _native.check("iris_postnet_forward", lib.iris_postnet_forward(
            self._handle, ctypes.c_void_p(mel.data_ptr()), B, T, ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(self._workspace.data_ptr()), ctypes.c_uint64(self._workspace.numel()),
            ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)))
        return out

    def __call__(self, mels_bt_f, training: bool = False) -> np.ndarray:
        """[B, n_mels, T] -> [B, n_mels, T] (input + residual), reference postnet.py:48-67."""
        if training:
            raise NotImplementedError("the MI355X build is inference-only")
        if isinstance(mels_bt_f, torch.Tensor):
            return self.forward_device(mels_bt_f)
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(mels_bt_f, dtype=np.float32)))
        self._ensure()
        return self.forward_device(x.to(self._device)).cpu().numpy()

    call = __call__
