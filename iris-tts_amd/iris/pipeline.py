"""Mel -> (PostNet) -> HiFiGAN -> waveform chunks, chained on one GPU (BASELINE.json configs[4]).

In the reference the three steps are separate host round trips (``scripts/synthesize.py:148-166`` runs the PostNet
and returns a numpy mel, ``:171-216`` converts it again and calls the vocoder).  Here the refined mel never leaves
HBM: ``PostNet.forward_device`` writes a device tensor that the vocoder engine reads chunk by chunk
(``iris.streaming``: 256-frame chunks + 13-frame halo, seams identical to the one-shot forward), so the first
audio is available after one chunk instead of after the whole utterance.
"""
from __future__ import annotations

from typing import Callable, Iterator, Optional

import numpy as np
import torch

from .streaming import StreamingVocoder


class MelToWavePipeline:
    """``postnet``: an ``iris.postnet.PostNet`` (or any callable mapping a device mel ``[B, n_mels, T]`` to a
    refined one), or None to vocode the mel as it is.  ``vocode``: ``GeneratorEngine.forward`` (or any callable
    ``[B, n_mels, W] -> [B, hop*W]``).  ``config``: the generator's ``GeneratorConfig`` -- hop length and the
    minimal halo are derived from it (V1 values when omitted; ``vocode.__self__.cfg`` is picked up when ``vocode``
    is a bound ``GeneratorEngine.forward``)."""

    def __init__(self, postnet: Optional[Callable], vocode: Callable, device: Optional[torch.device] = None,
                 hop_length: Optional[int] = None, chunk_frames: int = 256, halo_frames: Optional[int] = None,
                 group_chunks: int = 1, config=None):
        self.postnet = postnet
        self.device = device
        if config is None:
            config = getattr(getattr(vocode, "__self__", None), "cfg", None)
        self.streamer = StreamingVocoder(vocode, hop_length=hop_length, chunk_frames=chunk_frames,
                                         halo_frames=halo_frames, group_chunks=group_chunks, config=config)

    def refine(self, mel) -> torch.Tensor:
        """Host or device mel ``[B, n_mels, T]`` -> refined device mel (one PostNet pass over the whole utterance:
        0.3 % of the vocoder's work)."""
        if not isinstance(mel, torch.Tensor):
            mel = torch.from_numpy(np.ascontiguousarray(np.asarray(mel, dtype=np.float32)))
        if mel.dim() != 3:
            raise ValueError(f"expected mel [B, n_mels, T], got shape {tuple(mel.shape)}")
        if self.device is not None:
            mel = mel.to(self.device)
        if self.postnet is None:
            return mel
        fwd = getattr(self.postnet, "forward_device", self.postnet)
        return fwd(mel)

    def stream(self, mel) -> Iterator[torch.Tensor]:
        """Yields ``[B, hop*chunk]`` device tensors in order; their concatenation equals ``infer(mel)``."""
        yield from self.streamer.stream(self.refine(mel))

    def infer(self, mel) -> torch.Tensor:
        return self.streamer.infer(self.refine(mel))

    __call__ = infer
