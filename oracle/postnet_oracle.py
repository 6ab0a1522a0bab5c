"""CPU restatement of the reference PostNet at inference.  TEST INFRASTRUCTURE ONLY.

Follows ``/root/reference/src/iris/postnet.py:48-67`` read as text: transpose to [B, T, C]; for the
first L-1 layers Conv1D(k, padding='same') -> BatchNormalization(inference) -> tanh (dropout is the
identity at inference); then Conv1D(n_mels) -> BatchNormalization; transpose back and add to the input.
Keras semantics used: Conv1D kernel [k, C_in, C_out] with cross-correlation and symmetric zero padding
for odd k; BatchNormalization(y) = gamma * (y - moving_mean) / sqrt(moving_variance + 1e-3) + beta.
The BatchNorm is applied UNFOLDED here, so the product's fold is checked, not assumed.

Parity unpinned: Keras/JAX cannot run in this pipeline and the reference ships no PostNet vectors.
"""
import numpy as np

BN_EPSILON = 1e-3


def conv1d_same_keras(x_btc: np.ndarray, kernel: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """x [B, T, C_in], kernel [k, C_in, C_out] -> [B, T, C_out]; fp64 accumulation."""
    k = kernel.shape[0]
    pad = (k - 1) // 2
    B, T, _ = x_btc.shape
    xp = np.zeros((B, T + 2 * pad, x_btc.shape[2]), dtype=np.float64)
    xp[:, pad:pad + T] = x_btc
    y = np.zeros((B, T, kernel.shape[2]), dtype=np.float64)
    for kap in range(k):
        y += xp[:, kap:kap + T] @ kernel[kap].astype(np.float64)
    return y + bias.astype(np.float64)


def postnet_forward_np(weights: dict, mels_bt_f: np.ndarray, num_layers: int) -> np.ndarray:
    x = np.transpose(np.asarray(mels_bt_f, dtype=np.float64), (0, 2, 1))
    h = x
    for i in range(num_layers):
        p = "conv_out" if i == num_layers - 1 else f"convs.{i}"
        h = conv1d_same_keras(h, weights[f"{p}.kernel"], weights[f"{p}.bias"])
        h = (weights[f"{p}.gamma"].astype(np.float64) * (h - weights[f"{p}.moving_mean"].astype(np.float64))
             / np.sqrt(weights[f"{p}.moving_variance"].astype(np.float64) + BN_EPSILON) + weights[f"{p}.beta"].astype(np.float64))
        if i < num_layers - 1:
            h = np.tanh(h)
    return (np.asarray(mels_bt_f, dtype=np.float64) + np.transpose(h, (0, 2, 1))).astype(np.float32)
