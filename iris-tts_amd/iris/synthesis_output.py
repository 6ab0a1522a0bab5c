"""Output stage of the synthesis path: mel -> vocoder entry -> WAV.

What the reference does after the mel is ready (``scripts/synthesize.py:171-216``): ``np.array(mel)``,
call the vocoder, cast to float32, ``squeeze`` to 1-D, log the duration, create the output directory
and ``soundfile.write(path, audio, sample_rate)``, falling back to ``np.save(path.with_suffix(".npy"))``
when soundfile is missing or fails.  The reference also documents -- but never wired -- a pluggable
vocoder selected as ``--vocoder_entry module:function`` with the signature
``function(mel, sample_rate, hop_length) -> np.ndarray[samples]`` (``HIFIGAN_SETUP.md:61-75``,
``README.md:155-158``; ``synthesize.py`` imports ``importlib`` at :9 and never uses it).

This module wires exactly that convention and the output stage; ``python -m iris.synthesis_output``
is the CLI for a mel that already exists as a ``.npy`` file (the text front-end, encoder and VAE that
produce it are outside this repo's scope).
"""
from __future__ import annotations

import argparse
import importlib
import logging
import wave
from pathlib import Path
from typing import Callable, Optional, Union

import numpy as np

logger = logging.getLogger(__name__)

DEFAULT_VOCODER_ENTRY = "iris.hifigan_pretrained:infer_hifigan"
VocoderEntry = Callable[..., np.ndarray]


def resolve_vocoder_entry(spec: str) -> VocoderEntry:
    """``"package.module:function"`` -> the callable (``HIFIGAN_SETUP.md:61-75``)."""
    if not isinstance(spec, str) or spec.count(":") != 1:
        raise ValueError(f"--vocoder_entry must look like 'module:function', got {spec!r}")
    module_name, func_name = spec.split(":")
    if not module_name or not func_name:
        raise ValueError(f"--vocoder_entry must look like 'module:function', got {spec!r}")
    module = importlib.import_module(module_name)
    try:
        fn = getattr(module, func_name)
    except AttributeError as exc:
        raise AttributeError(f"module {module_name!r} has no attribute {func_name!r}") from exc
    if not callable(fn):
        raise TypeError(f"{spec} is not callable")
    return fn


def to_mono_float32(audio) -> np.ndarray:
    """float32 cast + squeeze to 1-D, as ``synthesize.py:199-203``."""
    audio = np.asarray(audio, dtype=np.float32)
    if audio.ndim > 1:
        audio = audio.squeeze()
    if audio.ndim != 1:
        raise ValueError(f"expected a single waveform after squeeze, got shape {audio.shape}")
    return audio


def write_wav(path: Union[str, Path], audio: np.ndarray, sample_rate: int = 22050) -> Path:
    """Writes a mono WAV.  ``soundfile`` is used when importable (the reference's writer,
    ``synthesize.py:211-213``); otherwise the standard-library ``wave`` module writes 16-bit PCM, which is
    also what soundfile's default WAV subtype produces.  On failure the samples are saved as ``.npy``
    next to the requested path, like the reference (:214-216).  Returns the path actually written."""
    out_path = Path(path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    audio = to_mono_float32(audio)
    try:
        try:
            import soundfile as sf  # optional dependency
            sf.write(str(out_path), audio, sample_rate)
        except ImportError:
            pcm = np.round(np.clip(audio, -1.0, 1.0) * 32767.0).astype("<i2")
            with wave.open(str(out_path), "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(int(sample_rate))
                w.writeframes(pcm.tobytes())
        logger.info(f"Wrote {out_path}")
        return out_path
    except Exception as exc:  # same fallback as the reference
        npy = out_path.with_suffix(".npy")
        np.save(str(npy), audio)
        logger.warning(f"WAV write failed; wrote numpy array instead: {npy} ({exc})")
        return npy


def vocode_to_wav(mel: np.ndarray, output_wav: Union[str, Path], vocoder_entry: str = DEFAULT_VOCODER_ENTRY,
                  sample_rate: int = 22050, hop_length: int = 256) -> np.ndarray:
    """mel ``[1, n_mels, T]`` or ``[n_mels, T]`` -> waveform written to ``output_wav``; returns the samples."""
    fn = resolve_vocoder_entry(vocoder_entry)
    mel = np.array(mel)
    logger.info(f"Using vocoder entry {vocoder_entry} ...")
    audio = to_mono_float32(fn(mel, sample_rate, hop_length))
    logger.info(f"Generated audio: {audio.shape}, duration={len(audio) / sample_rate:.2f}s")
    write_wav(output_wav, audio, sample_rate)
    return audio


def main(argv: Optional[list] = None) -> int:
    parser = argparse.ArgumentParser(description="Vocode a mel-spectrogram (.npy) to a WAV file")
    parser.add_argument("--mel", required=True, help=".npy file with a mel [n_mels, T] or [1, n_mels, T]")
    parser.add_argument("--output_wav", type=str, default="outputs/sample.wav")        # synthesize.py:67
    parser.add_argument("--vocoder", type=str, default="hifigan", choices=["hifigan"])  # README.md:155-158
    parser.add_argument("--vocoder_entry", type=str, default=DEFAULT_VOCODER_ENTRY)
    parser.add_argument("--sample_rate", type=int, default=22050)                       # synthesize.py:77
    parser.add_argument("--hop_length", type=int, default=256)                          # synthesize.py:78
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    mel = np.load(args.mel, allow_pickle=False)
    vocode_to_wav(mel, args.output_wav, args.vocoder_entry, args.sample_rate, args.hop_length)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
