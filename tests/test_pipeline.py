"""PostNet -> vocoder pipeline with chunked output (BASELINE configs[4])."""
import numpy as np
import pytest
import torch


def test_pipeline_control_flow_cpu():
    """Host logic only: stand-in stages with the generator's receptive-field structure."""
    from iris.pipeline import MelToWavePipeline
    hop = 4

    def postnet(m):                       # any elementwise + neighbour op
        return m + 0.5 * torch.roll(m, 1, dims=2)

    def vocode(m):                        # depends on +-13 frames like the generator does
        k = torch.ones(1, m.shape[1], 27) / 27.0
        y = torch.nn.functional.conv1d(m, k, padding=13)          # [B,1,W]
        return y.repeat_interleave(hop, dim=2)[:, 0, :]

    mel = torch.randn(2, 8, 700)
    pipe = MelToWavePipeline(postnet, vocode, hop_length=hop, chunk_frames=256)
    want = vocode(postnet(mel))
    chunks = list(pipe.stream(mel.numpy()))
    assert [c.shape[1] for c in chunks] == [256 * hop, 256 * hop, 188 * hop]
    assert torch.allclose(torch.cat(chunks, dim=1), want, atol=1e-6)
    assert torch.allclose(pipe(mel), want, atol=1e-6)
    assert torch.allclose(MelToWavePipeline(None, vocode, hop_length=hop).infer(mel), vocode(mel), atol=1e-6)
    with pytest.raises(ValueError):
        pipe.infer(torch.zeros(8, 10))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,exact", [("f32", True), ("bf16", True)])
def test_pipeline_on_gpu_equals_separate_stages(dtype, exact):
    """configs[4] shape: one utterance, 256-frame chunks.  The chained pipeline (mel stays in HBM) equals
    PostNet -> host -> one-shot vocoder bit for bit."""
    from iris._engine import GeneratorEngine
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    from iris.pipeline import MelToWavePipeline
    from iris.postnet import PostNet
    dev = torch.device("cuda", 0)
    cfg = GeneratorConfig()
    eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=11, gain=1.1, post_gain=10.0), dev, dtype=dtype)
    post = PostNet(n_mels=80, num_layers=3, channels=256, kernel_size=5, dropout=0.3, seed=5)
    mel = seeded_mel(21, 1, 600, log_mel=True)
    pipe = MelToWavePipeline(post, eng.forward, device=dev, chunk_frames=256)
    chunks = [c.clone() for c in pipe.stream(mel)]
    assert [c.shape for c in chunks] == [(1, 65536), (1, 65536), (1, 88 * 256)]
    refined_host = post(mel)                                        # the reference's flow: numpy out ...
    one_shot = eng.forward(torch.from_numpy(refined_host).to(dev))  # ... numpy in
    got = torch.cat(chunks, dim=1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, one_shot) if exact else torch.allclose(got, one_shot, atol=1e-5)
    if dtype == "f32":
        # not only self-consistent: against the CPU restatements (PostNet oracle -> generator oracle), one shot.
        # (PostNet is Keras-only in the reference: its oracle is "parity unpinned", tests/test_postnet.py.)
        from oracle import hifigan_oracle as orc
        from oracle import postnet_oracle as porc
        sd = seeded_state_dict(cfg, seed=11, gain=1.1, post_gain=10.0)
        refined = porc.postnet_forward_np(post.weights, mel, 3)
        want = orc.generator_forward_torch(orc.to_torch_folded(sd), refined).numpy()[:, 0, :]
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4
    eng.close()
