"""Output stage + ``--vocoder_entry module:function`` convention.
CPU tests use a stand-in entry function; the GPU test runs the real drop-in entry."""
import sys
import types
import wave

import numpy as np
import pytest
import torch

from iris import synthesis_output as so


@pytest.fixture()
def fake_entry(monkeypatch):
    calls = []
    mod = types.ModuleType("fake_vocoder_mod")

    def vocode(mel, sample_rate, hop_length):          # the documented signature (HIFIGAN_SETUP.md:66-75)
        calls.append((mel.shape, sample_rate, hop_length))
        t = mel.shape[-1]
        return np.linspace(-1.2, 1.2, t * hop_length, dtype=np.float64)[None, :]   # [1, samples], float64, clips

    mod.vocode = vocode
    mod.not_callable = 3
    monkeypatch.setitem(sys.modules, "fake_vocoder_mod", mod)
    return calls


def test_entry_resolution_errors(fake_entry):
    assert so.resolve_vocoder_entry("fake_vocoder_mod:vocode") is sys.modules["fake_vocoder_mod"].vocode
    for bad in ("fake_vocoder_mod", "a:b:c", ":f", "m:", 7):
        with pytest.raises(ValueError):
            so.resolve_vocoder_entry(bad)
    with pytest.raises(ModuleNotFoundError):
        so.resolve_vocoder_entry("no_such_module_xyz:f")
    with pytest.raises(AttributeError):
        so.resolve_vocoder_entry("fake_vocoder_mod:missing")
    with pytest.raises(TypeError):
        so.resolve_vocoder_entry("fake_vocoder_mod:not_callable")
    # the default entry is the drop-in module's function with the documented signature
    fn = so.resolve_vocoder_entry(so.DEFAULT_VOCODER_ENTRY)
    assert fn.__name__ == "infer_hifigan"


def test_vocode_to_wav_roundtrip(tmp_path, fake_entry):
    mel = np.zeros((1, 80, 5), dtype=np.float32)
    out = tmp_path / "nested" / "dir" / "sample.wav"        # parent directories are created (synthesize.py:208)
    audio = so.vocode_to_wav(mel, out, "fake_vocoder_mod:vocode", sample_rate=22050, hop_length=256)
    assert fake_entry == [((1, 80, 5), 22050, 256)]
    assert audio.dtype == np.float32 and audio.shape == (1280,)      # squeezed to 1-D float32 (synthesize.py:199-203)
    with wave.open(str(out), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 1280)
        pcm = np.frombuffer(w.readframes(1280), dtype="<i2")
    assert pcm[0] == -32767 and pcm[-1] == 32767                     # clipped to [-1, 1]
    assert np.abs(pcm / 32767.0 - np.clip(audio, -1, 1)).max() <= 0.5 / 32767 + 1e-7


def test_write_wav_falls_back_to_npy(tmp_path, monkeypatch):
    def boom(*a, **k):
        raise OSError("disk full")
    monkeypatch.setattr(so.wave, "open", boom)
    monkeypatch.setitem(sys.modules, "soundfile", None)              # force the stdlib path
    p = so.write_wav(tmp_path / "x.wav", np.zeros(10, np.float32))
    assert p.suffix == ".npy" and np.load(p).shape == (10,)
    with pytest.raises(ValueError):
        so.to_mono_float32(np.zeros((2, 3)))


def test_cli(tmp_path, fake_entry):
    mel_path = tmp_path / "mel.npy"
    np.save(mel_path, np.zeros((80, 3), dtype=np.float32))
    out = tmp_path / "o.wav"
    assert so.main(["--mel", str(mel_path), "--output_wav", str(out), "--vocoder", "hifigan",
                    "--vocoder_entry", "fake_vocoder_mod:vocode", "--hop_length", "256"]) == 0
    with wave.open(str(out), "rb") as w:
        assert w.getnframes() == 768


@pytest.mark.gpu
def test_default_entry_on_gpu(tmp_path, monkeypatch):
    """--vocoder_entry iris.hifigan_pretrained:infer_hifigan end to end (seeded checkpoint at the default path)."""
    from iris import hifigan_pretrained as hp
    from iris._weights import seeded_mel, seeded_state_dict
    ck = tmp_path / "generator.ckpt"
    torch.save({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed=2025, gain=1.18, post_gain=20.0).items()}, ck)
    monkeypatch.setattr(hp, "default_checkpoint_path", lambda: ck)
    monkeypatch.setattr(hp, "_vocoder_instance", None)
    mel = seeded_mel(5, 1, 40, log_mel=True)
    audio = so.vocode_to_wav(mel, tmp_path / "a.wav")
    assert audio.shape == (40 * 256,) and np.isfinite(audio).all() and np.abs(audio).max() <= 1.0
    with wave.open(str(tmp_path / "a.wav"), "rb") as w:
        assert w.getnframes() == 40 * 256 and w.getframerate() == 22050
        pcm = np.frombuffer(w.readframes(40 * 256), dtype="<i2")
    # the waveform against the CPU oracle, and the WAV's PCM samples against the oracle's waveform put through the
    # stage's own 16-bit rule (clip to [-1, 1], scale by 32767, round).  soundfile -- what the reference calls,
    # scripts/synthesize.py:212 -- is not importable here: its exact PCM rounding is "parity unpinned"; one LSB covers it.
    from oracle import hifigan_oracle as orc
    sd = seeded_state_dict(seed=2025, gain=1.18, post_gain=20.0)
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel).numpy()[0, 0]
    assert np.abs(audio - want).max() <= 1e-4
    assert np.abs(pcm.astype(np.int32) - np.rint(np.clip(want, -1, 1) * 32767.0).astype(np.int32)).max() <= 4   # 1e-4 * 32767 = 3.3 LSB
