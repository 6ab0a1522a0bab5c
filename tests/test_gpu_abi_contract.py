"""The header's contract (include/iris_hifigan.h, "Conventions") driven on the GPU through the C-ABI: a workspace that is
too small is refused, forwards run on the stream they are given, two handles with their own workspaces run concurrently on
two streams, and a forward never allocates inside a stream capture (IRIS_HIFIGAN_NOT_PREPARED instead)."""
import ctypes

import numpy as np
import pytest
import torch

from iris import _native
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc

pytestmark = pytest.mark.gpu
TOL_WAV = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def weights():
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    return cfg, sd, orc.to_torch_folded(sd)


def _raw_forward(eng, mel, wav, ws, ws_bytes, dtype_code, stream):
    return eng.lib.iris_hifigan_forward(eng._handle, ctypes.c_void_p(mel.data_ptr()), mel.shape[0], mel.shape[2],
                                        ctypes.c_void_p(wav.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ctypes.c_uint64(ws_bytes),
                                        dtype_code, ctypes.c_void_p(stream))


def test_workspace_too_small_is_refused_and_nothing_is_written(dev, weights):
    from iris._engine import GeneratorEngine
    cfg, sd, _ = weights
    eng = GeneratorEngine(cfg, sd, dev)
    mel = torch.from_numpy(seeded_mel(1, 2, 40)).to(dev)
    for dtype, code in (("f32", _native.DTYPE_F32), ("bf16", _native.DTYPE_BF16), ("f32s", _native.DTYPE_F32_SPLIT)):
        need = eng.workspace_bytes(2, 40, dtype)
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        wav = torch.full((2, 40 * 256), 7.0, device=dev)
        rc = _raw_forward(eng, mel, wav, ws, need - 1, code, torch.cuda.current_stream(dev).cuda_stream)
        assert rc == _native.STATUS_WORKSPACE_TOO_SMALL, (dtype, rc)
        msg = eng.lib.iris_hifigan_last_error().decode()
        assert str(need) in msg and str(need - 1) in msg
        torch.cuda.synchronize()
        assert bool((wav == 7.0).all()) and int(ws.count_nonzero()) == 0       # refused before any launch
        assert _raw_forward(eng, mel, wav, ws, need, code, torch.cuda.current_stream(dev).cuda_stream) == 0
        torch.cuda.synchronize()
        assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
    eng.close()


def test_forward_runs_on_the_stream_it_is_given(dev, weights):
    """A non-default stream: the forward is ordered behind work queued on THAT stream (a long fill of the input) and in
    front of what is queued after it, with no synchronisation on the default stream in between."""
    from iris._engine import GeneratorEngine
    cfg, sd, folded = weights
    eng = GeneratorEngine(cfg, sd, dev, graph_max_frames=0)
    mel_np = seeded_mel(21, 1, 300, log_mel=True)
    want = orc.generator_forward_torch(folded, mel_np).numpy()[:, 0, :]
    side = torch.cuda.Stream(device=dev)
    big = torch.empty(64 * 1024 * 1024, device=dev)
    mel = torch.zeros((1, 80, 300), device=dev)
    src = torch.from_numpy(mel_np).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            big.normal_()                     # keeps `side` busy: the copy below has not happened when forward is queued
        mel.copy_(src)
        wav = eng.forward(mel)                # reads torch.cuda.current_stream() = side
        out = wav * 1.0                       # a consumer on the same stream
    side.synchronize()
    assert np.abs(out.cpu().numpy() - want).max() <= TOL_WAV
    eng.close()


def test_two_handles_on_two_streams_run_concurrently(dev, weights):
    """One handle = one forward in flight; two handles with their own workspaces may overlap on two streams.  Both results are
    checked against the oracle, and bit for bit against the same forwards issued one after the other."""
    from iris._engine import GeneratorEngine
    cfg, sd, folded = weights
    engs = [GeneratorEngine(cfg, sd, dev, graph_max_frames=0) for _ in range(2)]
    mels_np = [seeded_mel(31 + i, 1, 282, log_mel=True) for i in range(2)]
    mels = [torch.from_numpy(m).to(dev) for m in mels_np]
    serial = [engs[i].forward(mels[i]).clone() for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [None, None]
    for rep in range(3):                                  # interleaved issue: launch k of one forward next to launch k of the other
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i] = engs[i].forward(mels[i])
    for s in streams:
        s.synchronize()
    for i in range(2):
        assert torch.equal(outs[i], serial[i])
        want = orc.generator_forward_torch(folded, mels_np[i]).numpy()[:, 0, :]
        assert np.abs(outs[i].cpu().numpy() - want).max() <= TOL_WAV
    for e in engs:
        e.close()


def test_forward_in_capture_without_prepare_is_refused_not_allocating(dev, weights):
    """a C-ABI caller that captures its FIRST forward of a dtype must get an explicit error, not an invalidated
    capture (the lazy build allocates and synchronises) and not a dtype that is disabled for the handle's lifetime."""
    from iris._engine import GeneratorEngine
    cfg, sd, _ = weights
    eng = GeneratorEngine(cfg, sd, dev, graph_max_frames=0)
    mel = torch.from_numpy(seeded_mel(5, 1, 40)).to(dev)
    need = max(eng.workspace_bytes(1, 40, "f32"), eng.workspace_bytes(1, 40, "bf16"))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    wav = torch.empty((1, 40 * 256), device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        try:
            rc16 = _raw_forward(eng, mel, wav, ws, need, _native.DTYPE_BF16, side.cuda_stream)      # packing not built: refused
            msg = eng.lib.iris_hifigan_last_error().decode()
            rc32 = _raw_forward(eng, mel, wav, ws, need, _native.DTYPE_F32, side.cuda_stream)       # fp32 needs nothing: captured
        finally:
            graph.capture_end()
    assert rc16 == _native.STATUS_NOT_PREPARED and "iris_hifigan_prepare" in msg
    assert rc32 == 0
    graph.replay()
    torch.cuda.synchronize()
    eager = eng.forward(mel)
    assert torch.equal(wav, eager)                        # the capture survived the refused call
    b16 = eng.forward(mel, dtype="bf16")                  # ... and bf16 still works afterwards (built lazily, outside a capture)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(b16).all()) and float((b16 - eager).abs().max()) <= 6e-2
    eng.close()


def test_release_host_weights(dev, weights):
    from iris._engine import GeneratorEngine
    cfg, sd, _ = weights
    eng = GeneratorEngine(cfg, sd, dev)
    mel = torch.from_numpy(seeded_mel(6, 1, 30)).to(dev)
    eng.prepare("bf16")
    eng.release_host_weights()
    eng.release_host_weights()                             # idempotent
    a = eng.forward(mel, dtype="f32")
    b = eng.forward(mel, dtype="bf16")                     # prepared before the release
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) <= 6e-2
    with pytest.raises(_native.NativeCallError) as exc:
        eng.forward(mel, dtype="f32s")                     # never prepared: its packing cannot be built any more
    assert exc.value.status == _native.STATUS_NOT_PREPARED
    eng.close()


def test_short_inputs_replay_a_graph_and_return_fresh_tensors(dev, weights):
    """GeneratorEngine.forward can replay a captured hipGraph for short inputs (graph_max_frames); the caller still owns what it gets back (a later
    call of the same shape must not overwrite it) and the samples are those of the eager launches, bit for bit."""
    from iris._engine import GeneratorEngine
    cfg, sd, _ = weights
    eng = GeneratorEngine(cfg, sd, dev, graph_max_frames=384)          # (off by default: a replay measured no faster than eager launches)
    eager = GeneratorEngine(cfg, sd, dev)
    assert eager.graph_max_frames == 0 and eng.graph_max_frames == 384
    mels = [torch.from_numpy(seeded_mel(40 + i, 1, 100)).to(dev) for i in range(3)]
    outs = [eng.forward(m) for m in mels]                 # same shape three times: one capture, three replays
    assert len(eng._graphs) == 1
    torch.cuda.synchronize()
    for m, o in zip(mels, outs):
        assert torch.equal(o, eager.forward(m))
    given = torch.empty((1, 100 * 256), device=dev)
    assert eng.forward(mels[0], out=given) is given and torch.equal(given, outs[0])
    eng.set_profiling(1)                                  # profiling wants per-launch events: eager path
    eng.forward(mels[0])
    torch.cuda.synchronize()
    assert len(eng.read_profile()) >= 20
    eng.set_profiling(False)
    long = torch.from_numpy(seeded_mel(50, 1, eng.graph_max_frames + 1)).to(dev)
    eng.forward(long)
    assert len(eng._graphs) == 1                          # above the threshold: eager, nothing captured
    eng.close()
    eager.close()
