"""Device-side generator: owns the native handle and the activation workspace.

PyTorch-ROCm is used here for device memory and streams only (``torch.empty``, ``data_ptr()``,
``torch.cuda.current_stream()``); all arithmetic happens in the HIP library behind the C-ABI.
This object replaces ``self.model(mel_tensor)`` of the reference
(src/iris/hifigan_pretrained.py:231-232, src/iris/vocoder.py:200).
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import List, Mapping, Optional

import numpy as np
import torch

from . import _native
from ._weights import GeneratorConfig, expected_weight_count, weight_blob

# wall-clock of the most recent cold start, filled in as its steps run (bench.py's `load` record): the checkpoint loader
# adds torch_load_ms / model_construct_ms / load_state_dict_ms, the engine constructor fold_ms (weight-norm fold into the
# blob) and create_ms (iris_hifigan_create: host repack into fragment order + upload)
LAST_LOAD_TIMINGS: dict = {}

KIND_NAMES = {0: "conv_pre", 1: "upsample", 2: "mrf_resblock_conv", 3: "conv_post"}
DTYPES = {"f32": _native.DTYPE_F32, "bf16": _native.DTYPE_BF16, "f32s": _native.DTYPE_F32_SPLIT}


def _dtype_code(dtype: str) -> int:
    try:
        return DTYPES[dtype]
    except KeyError:
        raise ValueError(f"dtype must be one of {sorted(DTYPES)}, got {dtype!r}") from None


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "iris vocoder (MI355X build): no HIP device is visible. This build has no CPU path; "
            "run on a gfx950 GPU.")
    return torch.device("cuda", torch.cuda.current_device())


class GeneratorEngine:
    """One HiFiGAN generator resident on one GPU.

    One forward in flight per engine: the engine owns ONE activation workspace (every forward and every
    captured hipGraph of it writes there) and the native handle owns per-forward device state, so forwards of
    one engine must be stream-ordered -- issue them on one stream, or build one engine per stream.  The native
    calls select the engine's device themselves (``include/iris_hifigan.h``), so an engine may live on a GPU
    that is not the caller's current device."""

    # forward() can replay a captured hipGraph instead of issuing 24-30 launches when batch * frames is at most
    # graph_max_frames.  OFF by default (0): measured on MI355X / ROCm 7.2 (profiles/r04_notes.md) a replay is no faster than
    # the eager launches once those carry no profiling events -- 100 frames: 0.754 ms eager, 0.798 ms replayed, 0.765 ms replayed
    # with the copy-out; 500 frames: 2.546 / 2.524; 1000: 4.743 / 4.753 -- round 3's 0.836 -> 0.809 ms had compared a replay
    # with eager launches that carried 11 events.  IRIS_VOCODER_GRAPH_FRAMES or the constructor argument turn it on.
    GRAPH_MAX_FRAMES = 0

    def __init__(self, cfg: GeneratorConfig, state_dict: Mapping[str, object],
                 device: Optional[torch.device] = None, dtype: Optional[str] = None,
                 graph_max_frames: Optional[int] = None):
        # default arithmetic of forward(): "f32" unless the caller or IRIS_VOCODER_DTYPE says "bf16" (the drop-in
        # wrappers construct engines without a dtype, so the environment variable switches them too)
        self.default_dtype = dtype or os.environ.get("IRIS_VOCODER_DTYPE", "f32")
        _dtype_code(self.default_dtype)
        self.cfg = cfg
        self.lib = _native.load()
        self.device = device if device is not None else require_gpu()
        if self.device.type != "cuda":
            raise RuntimeError(f"GeneratorEngine needs a HIP device, got {self.device}")
        t0 = time.perf_counter()
        blob = weight_blob(cfg, state_dict)
        assert blob.size == expected_weight_count(cfg)
        t1 = time.perf_counter()
        self._handle = ctypes.c_void_p()
        ccfg = _native.make_config(cfg)
        with torch.cuda.device(self.device):
            _native.check("iris_hifigan_create", self.lib.iris_hifigan_create(
                ctypes.byref(ccfg), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                ctypes.c_uint64(blob.size), ctypes.byref(self._handle)))
        LAST_LOAD_TIMINGS.update({"fold_ms": 1e3 * (t1 - t0), "create_ms": 1e3 * (time.perf_counter() - t1)})
        hop = ctypes.c_int32()
        _native.check("iris_hifigan_hop_length", self.lib.iris_hifigan_hop_length(self._handle, ctypes.byref(hop)))
        self.hop_length = int(hop.value)
        self._workspace: Optional[torch.Tensor] = None
        self._graphs: dict = {}
        self._profiling = 0
        # 0 = always eager; the environment variable switches the drop-in wrappers (which construct engines themselves)
        if graph_max_frames is None:
            graph_max_frames = int(os.environ.get("IRIS_VOCODER_GRAPH_FRAMES", self.GRAPH_MAX_FRAMES))
        self.graph_max_frames = max(0, int(graph_max_frames))

    # -- lifetime ----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self.lib.iris_hifigan_destroy(self._handle)
            self._handle = ctypes.c_void_p()
        self._graphs = {}
        self._workspace = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward -----------------------------------------------------------------------------
    def workspace_bytes(self, batch: int, frames: int, dtype: Optional[str] = None) -> int:
        dtype = dtype or self.default_dtype
        n = ctypes.c_uint64()
        _native.check("iris_hifigan_workspace_bytes", self.lib.iris_hifigan_workspace_bytes(
            self._handle, batch, frames, _dtype_code(dtype), ctypes.byref(n)))
        return int(n.value)

    def _get_workspace(self, nbytes: int) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None  # release before growing
            self._workspace = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self._workspace

    def release_workspace(self) -> None:
        """Frees the activation workspace (it grows to the largest shape seen: 229 KB per mel frame in fp32, bounded at 65,536
        frames = 15 GB because larger batches run as passes over sub-batches) and the
        captured graphs that point into it; the next forward allocates what it needs."""
        self._graphs = {}
        self._workspace = None

    def forward(self, mel: torch.Tensor, out: Optional[torch.Tensor] = None, dtype: Optional[str] = None) -> torch.Tensor:
        """mel: fp32 device tensor [B, in_channels, T] -> waveform fp32 [B, hop*T] (asynchronous on
        the current stream).  ``dtype`` selects the storage/arithmetic of the layers in between: "f32" (the
        parity path, <= 1e-4 against the reference) or "bf16" (bf16 activations and weights, fp32
        accumulation; BASELINE.json configs[2]); None = the engine's default_dtype."""
        dtype = dtype or self.default_dtype
        code = _dtype_code(dtype)
        if mel.dim() != 3 or mel.shape[1] != self.cfg.in_channels:
            raise ValueError(f"expected mel [B, {self.cfg.in_channels}, T], got {tuple(mel.shape)}")
        if mel.device != self.device:
            raise ValueError(f"mel is on {mel.device}, engine on {self.device}")
        mel = mel.to(torch.float32).contiguous()
        batch, _, frames = mel.shape
        if out is None:
            out = torch.empty((batch, frames * self.hop_length), dtype=torch.float32, device=self.device)
        elif out.shape != (batch, frames * self.hop_length) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 tensor [B, hop*T]")
        if batch == 0 or frames == 0:
            return out
        if (batch * frames <= self.graph_max_frames and not self._profiling
                and not torch.cuda.is_current_stream_capturing()):
            # short inputs: one graph launch instead of 24-30 kernel launches.  The graph owns static buffers; the result is
            # copied into `out` (a fresh tensor unless the caller gave one), so nothing the caller holds is overwritten by
            # the next call.  Profiling (per-launch events) and captures by the caller take the eager path below.
            out.copy_(self._replay(mel, dtype))
            return out
        nbytes = self.workspace_bytes(batch, frames, dtype)
        ws = self._get_workspace(nbytes)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _native.check("iris_hifigan_forward", self.lib.iris_hifigan_forward(
            self._handle, ctypes.c_void_p(mel.data_ptr()), batch, frames, ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(ws.data_ptr()), ctypes.c_uint64(ws.numel()), code,
            ctypes.c_void_p(stream)))
        return out

    __call__ = forward

    # -- hipGraph replay ---------------------------------------------------------------------------
    def forward_graph(self, mel: torch.Tensor, dtype: Optional[str] = None) -> torch.Tensor:
        """Same result as ``forward`` but the launches of one forward are captured once per
        (batch, frames) into a hipGraph and replayed: the host issues one graph launch instead of 24-30
        kernel launches, and the inter-kernel gaps shrink to the graph's own.  ``iris_hifigan_forward`` is
        capture-safe by construction (no allocation, no synchronisation, caller's stream; the packing of a dtype other
        than fp32 is built BEFORE the capture).  The returned tensor is the graph's static output buffer: it is
        overwritten by the next replay of the same shape (``forward`` on a short input copies it out instead).
        Per-launch profiling records are not produced in this mode."""
        dtype = dtype or self.default_dtype
        if mel.dim() != 3 or mel.shape[1] != self.cfg.in_channels:
            raise ValueError(f"expected mel [B, {self.cfg.in_channels}, T], got {tuple(mel.shape)}")
        batch, _, frames = mel.shape
        if batch == 0 or frames == 0:
            return self.forward(mel, dtype=dtype)
        return self._replay(mel, dtype)

    def _replay(self, mel: torch.Tensor, dtype: str) -> torch.Tensor:
        batch, _, frames = mel.shape
        key = (batch, frames, dtype)
        entry = self._graphs.get(key)
        if entry is not None and (self._workspace is None or self._workspace.data_ptr() != entry[3]):
            # the workspace was re-allocated (a larger shape came by): the captured pointers are stale
            del self._graphs[key]
            entry = None
        if entry is None:
            entry = self._capture(mel, key)
        graph, static_in, static_out, _ = entry
        static_in.copy_(mel.to(device=self.device, dtype=torch.float32))
        graph.replay()
        return static_out

    def _capture(self, mel: torch.Tensor, key) -> tuple:
        batch, frames, dtype = key
        static_in = torch.empty((batch, self.cfg.in_channels, frames), dtype=torch.float32, device=self.device)
        static_out = torch.empty((batch, frames * self.hop_length), dtype=torch.float32, device=self.device)
        self.prepare(dtype)                                                  # nothing may be built inside the capture
        self._get_workspace(self.workspace_bytes(batch, frames, dtype))     # allocate before capture
        ws_ptr = self._workspace.data_ptr()
        static_in.copy_(mel)
        was_profiling = self._profiling
        limit, self.graph_max_frames = self.graph_max_frames, 0             # the forwards below are the eager ones
        if was_profiling:                    # no event records inside a capture; the records collected so far are kept
            _native.check("iris_hifigan_pause_profiling", self.lib.iris_hifigan_pause_profiling(self._handle, 1))
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                                # warm-up outside capture
                self.forward(static_in, out=static_out, dtype=dtype)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.forward(static_in, out=static_out, dtype=dtype)
        finally:
            self.graph_max_frames = limit
            if was_profiling:
                _native.check("iris_hifigan_pause_profiling", self.lib.iris_hifigan_pause_profiling(self._handle, 0))
        entry = (graph, static_in, static_out, ws_ptr)
        if len(self._graphs) >= 32:                                      # (bounded: a caller with many distinct short shapes)
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = entry
        return entry

    def prepare(self, dtype: Optional[str] = None) -> None:
        """Builds the weight packing ``dtype`` needs now (otherwise the first forward of the dtype does, synchronously).
        fp32 needs nothing beyond what the constructor uploaded."""
        code = _dtype_code(dtype or self.default_dtype)
        with torch.cuda.device(self.device):
            _native.check("iris_hifigan_prepare", self.lib.iris_hifigan_prepare(self._handle, code))

    def release_host_weights(self) -> None:
        """Drops the native handle's host copy of the reference-layout weights (55.7 MB for V1), which it keeps to build
        the bf16 / split-product packings on first use.  After this, a dtype that was not prepared cannot be used."""
        _native.check("iris_hifigan_release_host_weights", self.lib.iris_hifigan_release_host_weights(self._handle))

    # -- profiling (bench.py roofline leg) -----------------------------------------------------
    def set_profiling(self, enabled) -> None:
        """False/0 off; True/1 one record (two events) per launch; 2 the MRF launches of a stage share one record
        (11 events per forward instead of 31: an event costs about 3 us of stream time)."""
        _native.check("iris_hifigan_set_profiling", self.lib.iris_hifigan_set_profiling(self._handle, int(enabled)))
        self._profiling = int(enabled)

    # -- intermediates (parity tests) ------------------------------------------------------------
    def forward_until(self, mel: torch.Tensor, stage: int, step: int, dtype: Optional[str] = None) -> dict:
        """Runs the forward up to and including MRF step ``step`` of upsample stage ``stage`` and returns the
        intermediates that are then in the workspace, as channels-first numpy arrays like the reference's
        tensors: ``pre`` [B, C0, T], ``up`` [B, C, L] (the stage's ConvTranspose1d output), ``y`` / ``xt``
        (lists per ResBlock branch, [B, C, L]) and ``mean_in_y0`` (True when the stage's last step already
        stored the MRF mean, hifigan_pretrained.py:131-137, in ``y[0]``).  bf16 storage is widened to fp32."""
        dtype = dtype or self.default_dtype
        code = _dtype_code(dtype)
        mel = mel.to(device=self.device, dtype=torch.float32).contiguous()
        batch, _, frames = mel.shape
        ws = self._get_workspace(self.workspace_bytes(batch, frames, dtype))
        wmap = _native.WorkspaceMap()
        _native.check("iris_hifigan_workspace_layout", self.lib.iris_hifigan_workspace_layout(
            self._handle, batch, frames, code, ctypes.byref(wmap)))
        folded = ctypes.c_int32(0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _native.check("iris_hifigan_forward_until", self.lib.iris_hifigan_forward_until(
            self._handle, ctypes.c_void_p(mel.data_ptr()), batch, frames, ctypes.c_void_p(ws.data_ptr()),
            ctypes.c_uint64(ws.numel()), code, stage, step, ctypes.byref(folded), ctypes.c_void_p(stream)))
        torch.cuda.synchronize(self.device)
        tdt = torch.float32 if wmap.element_bytes == 4 else torch.bfloat16
        length, ch = frames, self.cfg.upsample_initial_channel
        for i in range(stage + 1):
            length *= self.cfg.upsample_rates[i]
            ch //= 2

        def view(off: int, L: int, C: int) -> np.ndarray:
            n = batch * L * C * wmap.element_bytes
            t = ws[off:off + n].view(tdt).reshape(batch, L, C)
            return t.float().cpu().numpy().transpose(0, 2, 1).copy()

        nk = self.cfg.num_kernels
        y_off, xt_off = wmap.y_offset, wmap.xt_offset
        if folded.value & 2:                      # IRIS_HIFIGAN_UNTIL_X_IN_XT: the buffers' roles are swapped here
            y_off, xt_off = xt_off, y_off
        return {"pre": view(wmap.pre_offset, frames, self.cfg.upsample_initial_channel),
                "up": view(wmap.up_offset, length, ch),
                "y": [view(y_off[j], length, ch) for j in range(nk)],
                "xt": [view(xt_off[j], length, ch) for j in range(nk)],
                "mean_in_y0": bool(folded.value & 1)}

    def read_profile(self) -> List[dict]:
        """Per-launch records of every forward since set_profiling(True); the stream must have
        been synchronised."""
        n = ctypes.c_int32()
        _native.check("iris_hifigan_read_profile", self.lib.iris_hifigan_read_profile(
            self._handle, None, 0, ctypes.byref(n)))
        cap = max(int(n.value), 1)
        recs = (_native.LaunchRecord * cap)()
        _native.check("iris_hifigan_read_profile", self.lib.iris_hifigan_read_profile(
            self._handle, recs, cap, ctypes.byref(n)))
        out = []
        for i in range(min(n.value, cap)):
            r = recs[i]
            out.append({"kind": KIND_NAMES.get(r.kind, str(r.kind)), "stage": r.stage, "step": r.step,
                        "launches": max(int(r.launches), 1), "flops": r.flops, "bytes": r.bytes, "ms": r.ms})
        return out


def algorithmic_work(cfg: GeneratorConfig) -> dict:
    """FLOP and bytes (accounting L) per mel frame for cfg.  For the V1
    config: 614,105,088 FLOP and 1,305,936 activation elements per frame."""
    from ._weights import layer_specs

    specs = layer_specs(cfg)
    mac = 0
    elems = 0
    length = 1  # time positions per frame at the current layer input
    nk = cfg.num_kernels
    for s in specs:
        if s.kind == "conv" and s.name == "conv_pre":
            mac += s.c_in * s.c_out * s.k
            elems += s.c_in + s.c_out
        elif s.kind == "convt":
            mac += length * s.c_in * s.c_out * s.k
            stage = int(s.name.split(".")[1])
            n_in = 1 if stage == 0 else nk
            elems += length * s.c_in * n_in + length * s.stride * s.c_out
            length *= s.stride
        elif s.kind == "conv":
            mac += length * s.c_in * s.c_out * s.k
            n = length * s.c_out
            elems += 2 * n if ".convs1." in s.name else 3 * n
        else:  # post
            mac += length * s.c_in * s.k
            elems += length * s.c_in * nk + length
    weights = expected_weight_count(cfg)
    return {"flop_per_frame": 2 * mac, "mac_per_frame": mac, "elements_per_frame": elems,
            "weight_values": weights, "hop_length": cfg.hop_length}
