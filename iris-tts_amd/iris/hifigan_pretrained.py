"""MI355X drop-in for the reference module ``iris.hifigan_pretrained``.

Same call surface as the reference (``/root/reference/src/iris/hifigan_pretrained.py``):

    ==============================  ==========================================  =====================
    name                            behaviour kept                              reference lines
    ==============================  ==========================================  =====================
    ``HiFiGANModel(**hyper)``       ctor kwargs, state-dict keys (234 for V1),  77-121, 123-143
                                    ``forward([B,80,T]) -> [B,1,hop*T]``
    ``HiFiGANGenerator(ckpt)``      checkpoint container variants, errors,      146-242
                                    ``.model/.device/.checkpoint_path``,
                                    numpy in -> numpy out, squeeze rules
    ``get_pretrained_hifigan``      module-global singleton keyed on the path   245-283
    ``infer_hifigan``               ``(mel, sample_rate, hop_length, ckpt)``    286-317
    ==============================  ==========================================  =====================

What differs: the forward runs in hand-written HIP on a gfx950 GPU behind the C-ABI of
``include/iris_hifigan.h``; there is no CPU execution path (a missing GPU or extension raises).
Checkpoints are read with ``torch.load(weights_only=True)`` (the reference unpickles arbitrary
objects, hifigan_pretrained.py:165) and key mismatches are logged instead of silently dropped
(the reference passes ``strict=False``, :190).
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from ._engine import LAST_LOAD_TIMINGS, GeneratorEngine, require_gpu
from ._weights import GeneratorConfig, layer_specs, extract_state_dict

logger = logging.getLogger(__name__)

__all__ = ["ResBlock", "HiFiGANModel", "HiFiGANGenerator", "get_pretrained_hifigan", "infer_hifigan"]


class _WeightNormedConv(nn.Module):
    """Parameter holder with the reference's weight-norm parametrisation
    (``weight_g``, ``weight_v``, ``bias``); the arithmetic lives in the HIP library."""

    def __init__(self, weight_shape: Sequence[int], c_out: int, init: bool = True):
        super().__init__()
        self.bias = nn.Parameter(torch.empty(c_out, dtype=torch.float32), requires_grad=False)
        self.weight_g = nn.Parameter(torch.empty(weight_shape[0], 1, 1, dtype=torch.float32), requires_grad=False)
        self.weight_v = nn.Parameter(torch.empty(*weight_shape, dtype=torch.float32), requires_grad=False)
        self.initialised = False
        if init:
            self.reset_parameters()

    def reset_parameters(self) -> None:
        """torch's default conv init (U(+-1/sqrt(fan_in)) for weight and bias); weight_norm sets g = ||v|| so that the
        effective weight equals v.  A checkpoint loader that is about to overwrite the parameters defers this (13.9 M
        random numbers are a tenth of a second of its cold start) and runs it only for layers the checkpoint lacks."""
        shape = self.weight_v.shape
        bound = 1.0 / float(np.sqrt(shape[1] * shape[2]))
        with torch.no_grad():
            self.weight_v.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)
            self.weight_g.copy_(self.weight_v.flatten(1).norm(dim=1).reshape(-1, 1, 1))
        self.initialised = True

    @property
    def weight(self) -> torch.Tensor:
        """The folded weight ``v * g / ||v||``, like ``module.weight`` of a ``torch.nn.utils.weight_norm`` module
        (what the reference's layers expose, hifigan_pretrained.py:49-57)."""
        return torch._weight_norm(self.weight_v, self.weight_g, 0)


class ResBlock(nn.Module):
    """Parameters of one MRF branch: ``convs1[m]`` (dilated) and ``convs2[m]`` (dilation 1)
    -- reference hifigan_pretrained.py:38-71.  It cannot be called on its own: the fused MRF
    kernels advance all branches of a stage together."""

    def __init__(self, channels: int, kernel_size: int = 3, dilations: Sequence[int] = (1, 3, 5), _init_weights: bool = True):
        super().__init__()
        self.channels, self.kernel_size, self.dilations = channels, kernel_size, tuple(dilations)
        shape = (channels, channels, kernel_size)
        self.convs1 = nn.ModuleList(_WeightNormedConv(shape, channels, _init_weights) for _ in self.dilations)
        self.convs2 = nn.ModuleList(_WeightNormedConv(shape, channels, _init_weights) for _ in self.dilations)

    def forward(self, x):  # pragma: no cover - deliberate
        raise RuntimeError("ResBlock is a parameter container in the MI355X build; call HiFiGANModel")


class HiFiGANModel(nn.Module):
    """HiFiGAN generator with the reference's constructor, parameter names and forward contract;
    ``forward`` runs on the GPU through the HIP library."""

    def __init__(
        self,
        in_channels: int = 80,
        upsample_rates: Sequence[int] = (8, 8, 2, 2),
        upsample_kernel_sizes: Sequence[int] = (16, 16, 4, 4),
        upsample_initial_channel: int = 512,
        resblock_kernel_sizes: Sequence[int] = (3, 7, 11),
        resblock_dilation_sizes: Sequence[Sequence[int]] = ((1, 3, 5), (1, 3, 5), (1, 3, 5)),
        _init_weights: bool = True,
    ):
        """``_init_weights=False`` (not in the reference; used by the checkpoint loader below): parameters are allocated but
        not drawn; ``finish_init`` then draws only the layers a checkpoint did not provide."""
        super().__init__()
        self.config = GeneratorConfig(
            in_channels=in_channels,
            upsample_rates=tuple(upsample_rates),
            upsample_kernel_sizes=tuple(upsample_kernel_sizes),
            upsample_initial_channel=upsample_initial_channel,
            resblock_kernel_sizes=tuple(resblock_kernel_sizes),
            resblock_dilation_sizes=tuple(tuple(d) for d in resblock_dilation_sizes),
        )
        cfg = self.config
        self.num_kernels = cfg.num_kernels
        self.num_upsamples = cfg.num_upsamples
        specs = {s.name: s for s in layer_specs(cfg)}
        self.conv_pre = _WeightNormedConv(specs["conv_pre"].weight_shape, specs["conv_pre"].c_out, _init_weights)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for i in range(cfg.num_upsamples):
            s = specs[f"ups.{i}"]
            self.ups.append(_WeightNormedConv(s.weight_shape, s.c_out, _init_weights))
            ch = cfg.stage_channels(i)
            for k, d in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
                self.resblocks.append(ResBlock(ch, k, d, _init_weights))
        self.conv_post = _WeightNormedConv(specs["conv_post"].weight_shape, 1, _init_weights)
        self._engine: Optional[GeneratorEngine] = None
        self.eval()

    # -- parameter changes invalidate the packed device copy ---------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._drop_engine()
        result = super().load_state_dict(state_dict, strict=strict, assign=assign)
        if result.missing_keys or result.unexpected_keys:
            logger.warning("HiFiGAN state dict mismatch: %d missing keys (e.g. %s), %d unexpected keys (e.g. %s)",
                           len(result.missing_keys), result.missing_keys[:3],
                           len(result.unexpected_keys), result.unexpected_keys[:3])
        return result

    def finish_init(self, missing_keys: Sequence[str] = ()) -> None:
        """After a deferred construction: draws the default init for every layer that still has un-drawn parameters and
        was not (completely) provided by the checkpoint -- what ``load_state_dict(strict=False)`` leaves untouched in the
        reference keeps its random init there too (hifigan_pretrained.py:186-190)."""
        missing_layers = {k.rsplit(".", 1)[0] for k in missing_keys}
        for name, mod in self.named_modules():
            if isinstance(mod, _WeightNormedConv) and not mod.initialised:
                if name in missing_layers:
                    mod.reset_parameters()
                mod.initialised = True
        self._drop_engine()

    def to(self, *args, **kwargs):
        """Selects the GPU the engine will live on.  Parameters stay on the host: they are folded
        and repacked into MFMA fragment order when the engine is (re)built."""
        device = None
        if args and isinstance(args[0], (str, torch.device)):
            device = torch.device(args[0])
        device = torch.device(kwargs["device"]) if "device" in kwargs else device
        if device is not None:
            if device.type != "cuda":
                raise RuntimeError("the MI355X build of HiFiGANModel only runs on a HIP device (got %s)" % device)
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            self._target_device = device
        self._drop_engine()
        return self

    def _drop_engine(self):
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.close()
        self.__dict__["_engine"] = None
        self.__dict__["_packed_from"] = None

    def invalidate(self) -> None:
        """Drops the packed device copy of the weights; the next forward folds and repacks the current
        parameters.  Tracked in-place edits (``with torch.no_grad(): p.copy_(...)``, ``p.mul_(...)``,
        ``load_state_dict``) are noticed through the tensors' version counters and repack by themselves; edits
        that bypass the counters -- ``p.data.copy_(...)`` (``.data`` has a counter of its own by design), writing
        through ``p.numpy()`` -- must be followed by ``invalidate()``, or the GPU keeps running the old weights."""
        self._drop_engine()

    @staticmethod
    def _version_of(t):
        """A tensor's in-place version counter, or None for tensors that have none (created under
        ``torch.inference_mode()``: reading ``_version`` raises there).  Untracked tensors repack only on
        ``invalidate()`` / ``load_state_dict``."""
        try:
            return t._version
        except RuntimeError:
            return None

    def _parameter_versions(self):
        tensors = self.__dict__.get("_packed_from")
        if tensors is None:
            return None
        return tuple(self._version_of(t) for t in tensors)

    def engine(self) -> GeneratorEngine:
        if any(isinstance(m, _WeightNormedConv) and not m.initialised for m in self.modules()):
            # a deferred construction that nobody finished: every layer nothing was loaded into gets its default draw
            self.finish_init([f"{n}.weight_v" for n, m in self.named_modules()
                              if isinstance(m, _WeightNormedConv) and not m.initialised])
        if self._engine is not None and self._parameter_versions() != self.__dict__.get("_packed_versions"):
            logger.info("HiFiGAN parameters changed since they were packed for the GPU: repacking")
            self._drop_engine()
        if self._engine is None:
            device = getattr(self, "_target_device", None) or require_gpu()
            tensors = list(self.state_dict(keep_vars=True).values())
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            self._engine = GeneratorEngine(self.config, sd, device)
            self.__dict__["_packed_from"] = tensors
            self.__dict__["_packed_versions"] = tuple(self._version_of(t) for t in tensors)
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B, in_channels, T] -> [B, 1, hop*T]; same contract as hifigan_pretrained.py:123-143."""
        eng = self.engine()
        wav = eng.forward(x.to(device=eng.device, dtype=torch.float32))
        return wav.unsqueeze(1)


def _load_checkpoint(path: Path):
    """weights_only first; a pickled module (reference accepts those, :168-171) is refused."""
    try:
        return torch.load(str(path), map_location="cpu", weights_only=True)
    except Exception as exc:  # pickle with arbitrary objects, or not a torch file
        raise RuntimeError(
            f"Could not load HiFiGAN checkpoint. The checkpoint format may not be compatible. Error: {exc}"
        ) from exc


class HiFiGANGenerator:
    """Wrapper for a pre-trained generator checkpoint (reference hifigan_pretrained.py:146-242)."""

    def __init__(self, checkpoint_path: Union[str, Path]):
        self.checkpoint_path = Path(checkpoint_path)
        if not self.checkpoint_path.exists():
            raise FileNotFoundError(f"Checkpoint not found: {self.checkpoint_path}")
        logger.info(f"Loading HiFiGAN from: {self.checkpoint_path}")
        t0 = time.perf_counter()
        checkpoint = _load_checkpoint(self.checkpoint_path)
        state_dict = extract_state_dict(checkpoint)  # ValueError for non-dict, as :199-200
        t1 = time.perf_counter()
        logger.info("Creating HiFiGAN model with standard architecture...")
        # always the default config, like :186.  The random init is deferred: the checkpoint is about to overwrite it, and
        # layers it does not provide are drawn afterwards (finish_init) -- same outcome as the reference's strict=False load
        self.model = HiFiGANModel(_init_weights=False)
        t2 = time.perf_counter()
        try:
            result = self.model.load_state_dict(state_dict, strict=False)
            self.model.finish_init(result.missing_keys)
            if result.missing_keys:
                # a layer the checkpoint provides only in part was drawn whole: put the provided part back (in the reference
                # that part is loaded over the construction-time draw, :186-190)
                layers = {k.rsplit(".", 1)[0] for k in result.missing_keys}
                own = self.model.state_dict()
                part = {k: v for k, v in state_dict.items() if k in own and k.rsplit(".", 1)[0] in layers}
                if part:
                    nn.Module.load_state_dict(self.model, part, strict=False)
        except Exception as exc:
            logger.error(f"Failed to load state dict: {exc}")
            raise RuntimeError(
                f"Could not load HiFiGAN checkpoint. The checkpoint format may not be compatible. Error: {exc}"
            ) from exc
        self.model.eval()
        self.device = require_gpu()
        self.model.to(self.device)
        t3 = time.perf_counter()
        LAST_LOAD_TIMINGS.clear()
        LAST_LOAD_TIMINGS.update({"torch_load_ms": 1e3 * (t1 - t0), "model_construct_ms": 1e3 * (t2 - t1),
                                  "load_state_dict_ms": 1e3 * (t3 - t2)})
        logger.info(f"HiFiGAN loaded successfully on device: {self.device}")

    def __call__(self, mel: np.ndarray) -> np.ndarray:
        """[batch, n_mels, time] -> [batch, samples];  [n_mels, time] -> [samples]  (:208-242)."""
        mel = np.asarray(mel)
        squeeze_batch = False
        if mel.ndim == 2:
            mel = mel[np.newaxis, ...]
            squeeze_batch = True
        mel_tensor = torch.from_numpy(np.ascontiguousarray(mel)).float().to(self.device)
        with torch.no_grad():
            audio_tensor = self.model(mel_tensor)  # [batch, 1, samples]
        audio = audio_tensor.cpu().numpy()
        audio = audio.squeeze(1)
        if squeeze_batch:
            audio = audio[0]
        return audio


_vocoder_instance: Optional[HiFiGANGenerator] = None
_vocoder_checkpoint_path: Optional[Path] = None


def default_checkpoint_path() -> Path:
    """Same relative location as the reference (:270-273): <repo root>/models/hifigan/..."""
    return (Path(__file__).resolve().parent.parent.parent / "models" / "hifigan"
            / "models--speechbrain--tts-hifigan-ljspeech" / "snapshots"
            / "17fbdc3aae35b81e1554111fa54eab5f2b70cedb" / "generator.ckpt")


def get_pretrained_hifigan(checkpoint_path: Optional[Union[str, Path]] = None,
                           force_reload: bool = False) -> HiFiGANGenerator:
    """Singleton accessor (reference :250-283): reloads when the path changes or on force_reload."""
    global _vocoder_instance, _vocoder_checkpoint_path
    path = Path(checkpoint_path) if checkpoint_path is not None else default_checkpoint_path()
    if force_reload or _vocoder_instance is None or _vocoder_checkpoint_path != path:
        logger.info("Initializing HiFiGAN vocoder...")
        _vocoder_instance = HiFiGANGenerator(path)
        _vocoder_checkpoint_path = path
    return _vocoder_instance


def infer_hifigan(mel: np.ndarray, sample_rate: Optional[int] = None, hop_length: Optional[int] = None,
                  checkpoint_path: Optional[Union[str, Path]] = None) -> np.ndarray:
    """Entry point for ``--vocoder_entry iris.hifigan_pretrained:infer_hifigan`` (reference :286-317).
    ``sample_rate`` and ``hop_length`` are accepted and ignored, as in the reference."""
    vocoder = get_pretrained_hifigan(checkpoint_path)
    audio = vocoder(mel)
    if audio.ndim == 2 and audio.shape[0] == 1:
        audio = audio[0]
    return audio
