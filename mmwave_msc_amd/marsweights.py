"""Seeded random MARS weights in Keras' tensor shapes (train.py:33-106) -- numpy only, no torch: the CPU-side tools (the
oracle's child processes of bench_e2e.oracle_reference, fixture generators) import this without paying for `import torch`."""
import numpy as np

N_KEYPOINTS = 57   # 19 joints x (x, y, z)  (preprocessing.py:377)


def random_keras_weights(seed: int = 0, frames: int = 3) -> dict:
    """Seeded random weights with Keras shapes (Glorot-like scales, non-trivial BN stats)."""
    rng = np.random.default_rng(seed)
    three_d = frames > 1
    k = (3, 3, 3) if three_d else (3, 3)
    flat = (frames if three_d else 1) * 64 * 32
    hidden = 512 * (3 if three_d else 1)

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    rf = int(np.prod(k))
    w = {
        "conv1_w": glorot(k + (5, 16), rf * 5, rf * 16), "conv1_b": rng.normal(0, 0.05, 16).astype(np.float32),
        "conv2_w": glorot(k + (16, 32), rf * 16, rf * 32), "conv2_b": rng.normal(0, 0.05, 32).astype(np.float32),
        "bn1_gamma": rng.uniform(0.5, 1.5, 32).astype(np.float32), "bn1_beta": rng.normal(0, 0.1, 32).astype(np.float32),
        "bn1_mean": rng.normal(0.2, 0.1, 32).astype(np.float32), "bn1_var": rng.uniform(0.05, 0.5, 32).astype(np.float32),
        "dense1_w": glorot((flat, hidden), flat, hidden), "dense1_b": rng.normal(0, 0.05, hidden).astype(np.float32),
        "bn2_gamma": rng.uniform(0.5, 1.5, hidden).astype(np.float32), "bn2_beta": rng.normal(0, 0.1, hidden).astype(np.float32),
        "bn2_mean": rng.normal(0.2, 0.1, hidden).astype(np.float32), "bn2_var": rng.uniform(0.05, 0.5, hidden).astype(np.float32),
        "dense2_w": glorot((hidden, N_KEYPOINTS), hidden, N_KEYPOINTS), "dense2_b": rng.normal(0, 0.05, N_KEYPOINTS).astype(np.float32),
    }
    return w
