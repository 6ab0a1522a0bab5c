"""Chunked ("streaming") vocoding of long mels (BASELINE.json configs[4]).

The reference has no streaming (``TTSPipeline`` is a stub, src/iris/model.py:17-27; every caller hands
the vocoder a whole utterance).  The generator is, however, a finite-receptive-field convolution
stack: an output sample depends on the mel only within +-12.64 frames (3235 samples: probed on the
reference).  A chunk of ``chunk_frames`` frames vocoded together with
``halo_frames >= 13`` frames of context on each side therefore yields exactly the samples the one-shot
forward produces for that chunk: the zero padding the layers apply at the window edges can only reach
samples inside the halo, which are dropped.

``plan_chunks`` is pure host logic; ``StreamingVocoder`` drives any ``forward(mel_window) -> waveform``
callable -- the GPU engine in production, the CPU oracle in the tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, List, Optional

RECEPTIVE_FIELD_FRAMES = 13  # ceil(12.64); V1 config (== receptive_field_frames(GeneratorConfig()))


def receptive_field_frames(cfg) -> int:
    """Mel frames of context on either side that can reach the samples of one frame, for any generator
    configuration (``GeneratorConfig`` or an object with the same attributes).

    Exact interval propagation from the output back to the mel through the layers of
    ``HiFiGANModel.forward`` (src/iris/hifigan_pretrained.py:123-143): a Conv1d(k, dilation d, 'same')
    widens an index interval by d*(k-1)/2 on both sides; ConvTranspose1d(k, stride u, padding p=(k-u)/2)
    sends input i to outputs i*u - p + [0, k), so outputs [a, b] need inputs
    [ceil((a + p - k + 1)/u), floor((b + p)/u)]; a ResBlock chains, per dilation d, Conv1d(k, d) and
    Conv1d(k, 1) (:64-71), and the MRF takes the widest branch.  13 for the V1 configuration."""
    hop = 1
    for u in cfg.upsample_rates:
        hop *= int(u)
    c = 4096                                    # any frame far from the edges
    a, b = c * hop, c * hop + hop - 1
    post = (int(getattr(cfg, "post_kernel_size", 7)) - 1) // 2
    a, b = a - post, b + post
    branches = list(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes))
    for i in reversed(range(len(cfg.upsample_rates))):
        ext = max(sum((int(k) - 1) // 2 * (int(d) + 1) for d in dils) for k, dils in branches)
        a, b = a - ext, b + ext
        u, k = int(cfg.upsample_rates[i]), int(cfg.upsample_kernel_sizes[i])
        pad = (k - u) // 2
        a = -((-(a + pad - (k - 1))) // u)     # ceil
        b = (b + pad) // u
    pre = (int(getattr(cfg, "pre_kernel_size", 7)) - 1) // 2
    a, b = a - pre, b + pre
    return max(c - a, b - c, 0)


@dataclass(frozen=True)
class Chunk:
    """Frames [start, stop) are emitted; frames [win_start, win_stop) are vocoded."""

    start: int
    stop: int
    win_start: int
    win_stop: int

    def emit_slice(self, hop: int) -> slice:
        """Sample range of the window's waveform that belongs to [start, stop)."""
        return slice((self.start - self.win_start) * hop, (self.stop - self.win_start) * hop)


def plan_chunks(n_frames: int, chunk_frames: int = 256, halo_frames: int = RECEPTIVE_FIELD_FRAMES) -> List[Chunk]:
    if n_frames < 0 or chunk_frames < 1 or halo_frames < 0:
        raise ValueError("n_frames >= 0, chunk_frames >= 1 and halo_frames >= 0 are required")
    chunks = []
    for start in range(0, n_frames, chunk_frames):
        stop = min(start + chunk_frames, n_frames)
        chunks.append(Chunk(start, stop, max(0, start - halo_frames), min(n_frames, stop + halo_frames)))
    return chunks


class StreamingVocoder:
    """Vocodes ``mel [B, n_mels, T]`` chunk by chunk.

    ``forward`` maps a window ``[B, n_mels, W]`` to ``[B, hop*W]`` (e.g. ``GeneratorEngine.forward``);
    inputs/outputs may be torch tensors or numpy arrays -- they are only sliced along the last axis.
    """

    def __init__(self, forward: Callable, hop_length: Optional[int] = None, chunk_frames: int = 256,
                 halo_frames: Optional[int] = None, group_chunks: int = 1, config=None):
        """``config``: the generator's ``GeneratorConfig``; when given, the hop length and the minimal halo are
        computed from it (``receptive_field_frames``), otherwise the V1 values (256, 13) are assumed -- pass the
        config for any other architecture, or the seams silently differ from the one-shot output."""
        need = receptive_field_frames(config) if config is not None else RECEPTIVE_FIELD_FRAMES
        if halo_frames is None:
            halo_frames = need
        if halo_frames < need:
            raise ValueError(
                f"halo_frames={halo_frames} is smaller than the generator's receptive field "
                f"({need} frames): chunk seams would differ from the one-shot output")
        if group_chunks < 1:
            raise ValueError("group_chunks >= 1 is required")
        if hop_length is None:
            hop_length = int(config.hop_length) if config is not None else 256
        elif config is not None and hop_length != int(config.hop_length):
            raise ValueError(f"hop_length={hop_length} does not match the configuration's {config.hop_length}")
        self.forward = forward
        self.hop_length = hop_length
        self.chunk_frames = chunk_frames
        self.halo_frames = halo_frames
        self.group_chunks = group_chunks

    def stream(self, mel) -> Iterator:
        """Yields the waveform of each chunk, ``[B, hop*(stop-start)]``, in order.

        With ``group_chunks = G > 1`` the first chunk is still vocoded alone (time to first audio is unchanged);
        after it, up to G consecutive chunks are vocoded as ONE window -- they are adjacent in time, so their windows
        merge into ``[first.win_start, last.win_stop)`` with a single halo on either side instead of one per chunk --
        and their waveforms are yielded one by one.  A sample depends on the mel within the halo only, so the samples
        are those of the per-chunk windows (and of the one-shot forward); a short window no longer leaves most of the
        GPU idle (fp32, one MI355X: a 282-frame window takes 1.7 ms, 781 frames 3.9 ms), and G chunks cost
        ``G*chunk + 2*halo`` frames instead of ``G*(chunk + 2*halo)``.  (Round 3 stacked equal-width windows along the
        batch axis: same idea, but every chunk kept its own halo and the ragged last chunk ran alone.)"""
        if mel.ndim != 3:
            raise ValueError(f"expected mel [B, n_mels, T], got shape {tuple(mel.shape)}")
        chunks = plan_chunks(mel.shape[2], self.chunk_frames, self.halo_frames)
        i = 0
        while i < len(chunks):
            n = 1 if (self.group_chunks == 1 or i == 0) else min(self.group_chunks, len(chunks) - i)
            group = chunks[i:i + n]
            win_start, win_stop = group[0].win_start, group[-1].win_stop
            wav = self.forward(mel[:, :, win_start:win_stop])
            for c in group:
                yield wav[:, (c.start - win_start) * self.hop_length:(c.stop - win_start) * self.hop_length]
            i += n

    def infer(self, mel):
        """Concatenation of ``stream(mel)``; equals the one-shot forward of the whole mel."""
        parts = list(self.stream(mel))
        if not parts:
            return self.forward(mel)
        if hasattr(parts[0], "detach"):
            import torch
            return torch.cat(parts, dim=1)
        import numpy as np
        return np.concatenate(parts, axis=1)


class StreamingSession:
    """Incremental chunked vocoding: the mel of ONE utterance arrives piece by piece (``push``), audio leaves chunk by chunk
    as soon as its right-hand context exists, and ``flush`` ends the utterance.  The concatenation of everything returned
    equals the one-shot forward of the concatenated mel: chunk ``[s, s + chunk)`` is vocoded from the window
    ``[max(0, s - halo), min(T, s + chunk + halo))`` exactly as ``plan_chunks`` would cut it once the length T is known, and a
    sample depends on the mel within the halo only (``receptive_field_frames``).

    The reference has no streaming (``TTSPipeline`` is a stub, src/iris/model.py:17-27); BASELINE.json configs[4] asks for
    256-frame chunks.  ``StreamingVocoder`` covers a mel that is already complete; this class is for a producer (an
    acoustic model emitting frames) that is still running.  Latency: a chunk is emitted ``halo`` frames (13 for V1 = 0.15 s
    of audio) after its last frame has arrived.  At most ``chunk + 2 * halo`` frames are buffered."""

    def __init__(self, forward: Callable, hop_length: Optional[int] = None, chunk_frames: int = 256,
                 halo_frames: Optional[int] = None, config=None):
        sv = StreamingVocoder(forward, hop_length=hop_length, chunk_frames=chunk_frames, halo_frames=halo_frames, config=config)
        self.forward, self.hop_length = sv.forward, sv.hop_length
        self.chunk_frames, self.halo_frames = sv.chunk_frames, sv.halo_frames
        self._pieces: list = []       # buffered mel pieces [B, n_mels, t]; together they cover frames [self._base, self._total)
        self._base = 0                # absolute index of the first buffered frame
        self._total = 0               # frames received so far
        self._next = 0                # first frame not yet emitted
        self._closed = False

    @property
    def frames_received(self) -> int:
        return self._total

    @property
    def frames_emitted(self) -> int:
        return self._next

    def _cat(self):
        if len(self._pieces) > 1:
            first = self._pieces[0]
            if hasattr(first, "detach"):
                import torch
                self._pieces = [torch.cat(self._pieces, dim=2)]
            else:
                import numpy as np
                self._pieces = [np.concatenate(self._pieces, axis=2)]
        return self._pieces[0]

    def _emit_ready(self, final: bool) -> List:
        out = []
        while self._next < self._total:
            stop = min(self._next + self.chunk_frames, self._total)
            whole = stop - self._next == self.chunk_frames
            if not final and (not whole or self._total < stop + self.halo_frames):
                break                                             # the chunk, or its right-hand context, is still incomplete
            win_start = max(0, self._next - self.halo_frames)
            win_stop = min(self._total, stop + self.halo_frames)
            buf = self._cat()
            wav = self.forward(buf[:, :, win_start - self._base:win_stop - self._base])
            out.append(wav[:, (self._next - win_start) * self.hop_length:(stop - win_start) * self.hop_length])
            self._next = stop
            keep_from = max(0, self._next - self.halo_frames)    # the next chunk's left-hand context
            if keep_from > self._base:
                self._pieces = [buf[:, :, keep_from - self._base:]]
                self._base = keep_from
        return out

    def push(self, mel_piece) -> List:
        """Appends ``mel_piece [B, n_mels, t]`` (t >= 0) and returns the waveform chunks ``[B, hop * chunk]`` that became
        computable, in order (possibly none)."""
        if self._closed:
            raise RuntimeError("the session was flushed: start a new one for the next utterance")
        if mel_piece.ndim != 3:
            raise ValueError(f"expected mel [B, n_mels, t], got shape {tuple(mel_piece.shape)}")
        if self._pieces and tuple(mel_piece.shape[:2]) != tuple(self._pieces[0].shape[:2]):
            raise ValueError("every piece of an utterance must have the same batch size and mel bins")
        if mel_piece.shape[2] > 0:
            self._pieces.append(mel_piece)
            self._total += int(mel_piece.shape[2])
        return self._emit_ready(final=False)

    def flush(self) -> List:
        """Ends the utterance: returns the chunks that were waiting for context (the last one may be shorter than
        ``chunk_frames``); the right edge of the last window is the true end of the mel, as in the one-shot forward."""
        out = self._emit_ready(final=True)
        self._closed = True
        self._pieces = []
        return out
