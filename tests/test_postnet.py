"""PostNet on device against the numpy restatement of the reference's
Keras model (oracle/postnet_oracle.py).  Parity unpinned by the reference (Keras-only, no vectors)."""
import numpy as np
import pytest
import torch

from iris.postnet import PostNet, fold_batchnorm
from oracle import postnet_oracle as porc
from oracle import hifigan_oracle as orc


def _randomise(pn: PostNet, seed: int) -> None:
    """Non-trivial BatchNorm statistics so that the fold is exercised."""
    rng = np.random.default_rng(seed)
    w = dict(pn.weights)
    for key, val in w.items():
        if key.endswith(".bias") or key.endswith(".beta") or key.endswith(".moving_mean"):
            w[key] = rng.normal(0, 0.3, val.shape).astype(np.float32)
        elif key.endswith(".gamma"):
            w[key] = rng.uniform(0.5, 1.5, val.shape).astype(np.float32)
        elif key.endswith(".moving_variance"):
            w[key] = rng.uniform(0.2, 2.0, val.shape).astype(np.float32)
    pn.set_weights_dict(w)


def test_constructor_and_fold_on_cpu(tmp_path):
    pn = PostNet(n_mels=80, num_layers=3, channels=256, kernel_size=5, dropout=0.3, seed=0)   # synthesize.py:152-158
    assert pn.get_config() == {"n_mels": 80, "num_layers": 3, "channels": 256, "kernel_size": 5, "dropout": 0.3}
    assert pn.weights["convs.0.kernel"].shape == (5, 80, 256) and pn.weights["conv_out.kernel"].shape == (5, 256, 80)
    assert PostNet(80).num_layers == 4 and len([k for k in PostNet(80).weights if k.endswith("kernel")]) == 4
    with pytest.raises(AssertionError):
        PostNet(80, num_layers=1)
    _randomise(pn, 1)
    # folded conv == conv followed by inference BatchNorm
    x = np.random.default_rng(2).standard_normal((2, 7, 80))
    p = "convs.0"
    y_ref = porc.conv1d_same_keras(x, pn.weights[f"{p}.kernel"], pn.weights[f"{p}.bias"])
    y_ref = (pn.weights[f"{p}.gamma"] * (y_ref - pn.weights[f"{p}.moving_mean"]) / np.sqrt(pn.weights[f"{p}.moving_variance"] + 1e-3)
             + pn.weights[f"{p}.beta"])
    w, b = fold_batchnorm(*(pn.weights[f"{p}.{n}"] for n in ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance")))
    y_fold = orc.conv1d_np(x.transpose(0, 2, 1).astype(np.float32), w, b, 1).transpose(0, 2, 1)
    assert np.abs(y_fold - y_ref).max() <= 2e-5
    assert pn.folded_blob().size == 5 * (80 * 256 + 256 * 256 + 256 * 80) + 256 + 256 + 80
    path = tmp_path / "pn.npz"
    pn.save_weights(str(path))
    pn2 = PostNet(80, num_layers=3, seed=5)
    pn2.load_weights(str(path))
    np.testing.assert_array_equal(pn2.weights["conv_out.gamma"], pn.weights["conv_out.gamma"])
    with pytest.raises(NotImplementedError):
        pn.load_weights("postnet_best.weights.h5")
    with pytest.raises(NotImplementedError):
        pn(np.zeros((1, 80, 4)), training=True)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            pn(np.zeros((1, 80, 4), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("n_mels,layers,ch,k,B,T", [(80, 3, 256, 5, 2, 333), (80, 4, 256, 5, 1, 1000), (20, 2, 24, 3, 3, 17), (80, 3, 256, 5, 1, 1)])
def test_postnet_matches_oracle_gpu(n_mels, layers, ch, k, B, T):
    pn = PostNet(n_mels, num_layers=layers, channels=ch, kernel_size=k, seed=3)
    _randomise(pn, 4)
    mel = np.random.default_rng(6).standard_normal((B, n_mels, T)).astype(np.float32)
    got = pn(mel)
    want = porc.postnet_forward_np(pn.weights, mel, layers)
    assert got.shape == mel.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    # device tensor in -> device tensor out, and it chains straight into the vocoder's input layout
    dev_out = pn(torch.from_numpy(mel).cuda())
    assert dev_out.is_cuda and np.array_equal(dev_out.cpu().numpy(), got)
