"""Streaming chunker: planning logic (CPU) and seam exactness against the oracle (CPU) and the HIP
engine (GPU).  The reference has no streaming; the property under test is the one DESIGN.md states:
chunked output == one-shot output, because the generator's receptive field is +-12.64 frames."""
import numpy as np
import pytest
import torch

from iris.streaming import RECEPTIVE_FIELD_FRAMES, StreamingSession, StreamingVocoder, plan_chunks
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc


def test_plan_covers_every_frame_once():
    for T in (0, 1, 255, 256, 257, 1024, 1000):
        chunks = plan_chunks(T, 256, 13)
        assert sum(c.stop - c.start for c in chunks) == T
        pos = 0
        for c in chunks:
            assert c.start == pos and c.win_start <= c.start < c.stop <= c.win_stop <= T
            assert c.start - c.win_start == min(13, c.start) and c.win_stop - c.stop == min(13, T - c.stop)
            s = c.emit_slice(256)
            assert s.stop - s.start == 256 * (c.stop - c.start)
            pos = c.stop
    assert len(plan_chunks(1024, 256)) == 4          # BASELINE.json configs[4]: 4 chunks of 256 frames
    with pytest.raises(ValueError):
        plan_chunks(10, 0)
    with pytest.raises(ValueError):
        StreamingVocoder(lambda m: m, halo_frames=RECEPTIVE_FIELD_FRAMES - 1)


def test_chunked_oracle_equals_one_shot():
    """On the CPU oracle: 13 frames of halo make the seams exact (receptive field +-12.64 frames)."""
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    folded = orc.to_torch_folded(sd)
    mel = seeded_mel(1005, 1, 70, log_mel=True)
    fwd = lambda m: orc.generator_forward_torch(folded, np.ascontiguousarray(m)).numpy()[:, 0, :]
    full = fwd(mel)
    chunked = StreamingVocoder(fwd, chunk_frames=16).infer(mel)
    assert chunked.shape == full.shape
    assert np.abs(chunked - full).max() <= 2e-5      # fp32 rounding only (ATen picks kernels by length and thread count)
    # and a halo that is too small is visibly wrong at the seams (the check above is not vacuous)
    bad = np.concatenate([fwd(mel[:, :, max(0, s - 4):min(70, s + 16 + 4)])[:, (s - max(0, s - 4)) * 256:][:, :256 * min(16, 70 - s)]
                          for s in range(0, 70, 16)], axis=1)
    assert np.abs(bad - full).max() > 1e-4


@pytest.mark.gpu
def test_streaming_engine_equals_one_shot_gpu():
    from iris._engine import GeneratorEngine
    dev = torch.device("cuda", 0)
    cfg = GeneratorConfig()
    eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0), dev)
    mel = torch.from_numpy(seeded_mel(1005, 2, 1024, log_mel=True)).to(dev)       # configs[4]: T=1024, 4 chunks
    full = eng.forward(mel).clone()
    sv = StreamingVocoder(lambda m: eng.forward(m.contiguous()).clone(), hop_length=eng.hop_length, chunk_frames=256)
    parts = list(sv.stream(mel))
    assert len(parts) == 4 and all(p.shape == (2, 65536) for p in parts)
    chunked = torch.cat(parts, dim=1)
    # same fmaf chains per output element whatever the tile origin: seams are exact to rounding
    assert (chunked - full).abs().max().item() <= 1e-6
    # ... and not merely self-consistent: item 1 of the chunked output against the CPU oracle's one-shot waveform
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel[1:2].cpu().numpy()).numpy()[0, 0]
    assert np.abs(chunked[1].cpu().numpy() - want).max() <= 1e-4
    eng.close()


def test_grouped_chunks_equal_ungrouped():
    """group_chunks vocodes consecutive chunks as one merged window: same samples, same order, first chunk alone."""
    calls = []

    def fwd(m):                                    # +-13-frame dependence like the generator, hop 4
        calls.append(tuple(m.shape))
        k = np.ones(27) / 27.0
        y = np.stack([np.convolve(m[b].sum(0), k, mode="same") for b in range(m.shape[0])])
        return np.repeat(y, 4, axis=1)

    mel = np.random.default_rng(0).standard_normal((2, 5, 1500)).astype(np.float32)
    plain = list(StreamingVocoder(fwd, hop_length=4, chunk_frames=256).stream(mel))
    calls.clear()
    grouped = list(StreamingVocoder(fwd, hop_length=4, chunk_frames=256, group_chunks=3).stream(mel))
    assert len(plain) == len(grouped) == 6
    for a, b in zip(plain, grouped):
        assert a.shape == b.shape and np.array_equal(a, b)
    # chunk 0 alone (frames 0..269), then chunks 1-3 as ONE window (frames 243..1037), then chunks 4-5 (frames 1011..1500)
    assert calls == [(2, 5, 269), (2, 5, 794), (2, 5, 489)]
    with pytest.raises(ValueError):
        StreamingVocoder(fwd, group_chunks=0)


@pytest.mark.gpu
def test_grouped_streaming_on_gpu_is_exact():
    import torch
    from iris._engine import GeneratorEngine
    cfg = GeneratorConfig()
    dev = torch.device("cuda", 0)
    eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=3, gain=1.1, post_gain=10.0), dev)
    mel = torch.from_numpy(seeded_mel(9, 1, 1400, log_mel=True)).to(dev)
    one_shot = eng.forward(mel).clone()
    grouped = StreamingVocoder(eng.forward, group_chunks=4, config=cfg).infer(mel)
    assert torch.equal(grouped, one_shot)
    sd = seeded_state_dict(cfg, seed=3, gain=1.1, post_gain=10.0)
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel.cpu().numpy()).numpy()[0, 0]
    assert np.abs(grouped[0].cpu().numpy() - want).max() <= 1e-4
    eng.close()


def test_incremental_session_equals_complete_mel_streaming():
    """StreamingSession (mel arriving piece by piece) cuts the same windows as plan_chunks does once the length is known:
    same chunks, same samples, whatever the piece sizes; a chunk leaves `halo` frames after its last frame arrived."""
    calls = []

    def fwd(m):                                    # +-13-frame dependence like the generator, hop 4
        calls.append(tuple(m.shape))
        k = np.ones(27) / 27.0
        y = np.stack([np.convolve(m[b].sum(0), k, mode="same") for b in range(m.shape[0])])
        return np.repeat(y, 4, axis=1)

    rng = np.random.default_rng(5)
    for T in (0, 1, 13, 255, 256, 257, 269, 270, 700, 1024):
        mel = rng.standard_normal((2, 5, T)).astype(np.float32)
        want = list(StreamingVocoder(fwd, hop_length=4, chunk_frames=256).stream(mel))
        want_calls = calls[:]
        calls.clear()
        ses = StreamingSession(fwd, hop_length=4, chunk_frames=256)
        got, pos = [], 0
        while pos < T:
            n = int(rng.integers(0, 90))
            before = len(got)
            got += ses.push(mel[:, :, pos:pos + n])
            pos = min(T, pos + n)
            assert ses.frames_received == pos
            if len(got) > before:                  # emitted exactly when the right-hand context became complete
                assert ses.frames_emitted + RECEPTIVE_FIELD_FRAMES <= pos
            assert pos - ses.frames_emitted < 256 + RECEPTIVE_FIELD_FRAMES + 90
        got += ses.flush()
        assert calls == want_calls, T              # the very same windows
        calls.clear()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.shape == b.shape and np.array_equal(a, b)
        with pytest.raises(RuntimeError):
            ses.push(mel[:, :, :1])
    with pytest.raises(ValueError):
        StreamingSession(fwd, halo_frames=RECEPTIVE_FIELD_FRAMES - 1)


def test_incremental_session_on_the_oracle():
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    folded = orc.to_torch_folded(sd)
    mel = seeded_mel(1006, 1, 61, log_mel=True)
    fwd = lambda m: orc.generator_forward_torch(folded, np.ascontiguousarray(m)).numpy()[:, 0, :]
    full = fwd(mel)
    ses = StreamingSession(fwd, chunk_frames=16, config=cfg)
    parts = []
    for s in range(0, 61, 7):
        parts += ses.push(mel[:, :, s:s + 7])
    parts += ses.flush()
    got = np.concatenate(parts, axis=1)
    assert got.shape == full.shape and np.abs(got - full).max() <= 2e-5


@pytest.mark.gpu
def test_incremental_session_on_gpu_is_exact():
    from iris._engine import GeneratorEngine
    cfg = GeneratorConfig()
    dev = torch.device("cuda", 0)
    sd = seeded_state_dict(cfg, seed=3, gain=1.1, post_gain=10.0)
    eng = GeneratorEngine(cfg, sd, dev)
    mel = torch.from_numpy(seeded_mel(10, 2, 900, log_mel=True)).to(dev)
    one_shot = eng.forward(mel).clone()
    ses = StreamingSession(lambda m: eng.forward(m.contiguous()).clone(), config=cfg)
    parts = []
    for s in range(0, 900, 100):                   # an acoustic model emitting 100 frames at a time
        parts += ses.push(mel[:, :, s:s + 100])
    parts += ses.flush()
    assert [p.shape[1] for p in parts] == [65536, 65536, 65536, 132 * 256]
    got = torch.cat(parts, dim=1)
    assert torch.equal(got, one_shot)
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel[1:2].cpu().numpy()).numpy()[0, 0]
    assert np.abs(got[1].cpu().numpy() - want).max() <= 1e-4
    eng.close()
