"""The MARS CNN forward pass a second time, on torch's CPU operators -- a second witness for oracle/mars_np.py and the CNN of
the CPU baselines (bench.py: e2e.cpu_baseline, single_scene.cpu_baseline).

TEST INFRASTRUCTURE ONLY.  Restates `define_CNN_3D` / `define_CNN` (/root/reference/src/train.py:71-106, 33-68) at inference
layer by layer, exactly as Keras evaluates the graph -- no folding, no re-ordering of weights: Conv (channels-last, "same",
ReLU) twice, BatchNormalization(eps 1e-3) on the channel axis, Flatten in (d,h,w,c) order, Dense + ReLU, BatchNormalization,
Dense.  Where mars_np.py slides the kernel with numpy slices, this one calls torch.nn.functional.conv3d / conv2d: two
independent evaluations of the same published layer semantics (keras / tensorflow are absent from the image: parity with
Keras itself stays unpinned).  dtype float64 for the witness test, float32 for the CPU baseline (Keras' own dtype)."""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


class MarsTorchCPU:
    def __init__(self, weights: dict, dtype=torch.float32, threads: int = None):
        self.dtype = dtype
        if threads:
            torch.set_num_threads(int(threads))
        w = {k: torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype) for k, v in weights.items()}
        self.three_d = w["conv1_w"].ndim == 5
        perm = (4, 3, 0, 1, 2) if self.three_d else (3, 2, 0, 1)      # Keras (k.., in, out) -> torch (out, in, k..)
        self.c1w, self.c2w = w["conv1_w"].permute(*perm).contiguous(), w["conv2_w"].permute(*perm).contiguous()
        self.w = w

    def _bn(self, x, name):
        w = self.w
        return w[name + "_gamma"] * (x - w[name + "_mean"]) / torch.sqrt(w[name + "_var"] + BN_EPS) + w[name + "_beta"]

    @torch.no_grad()
    def forward(self, x):
        """x: (B,3,8,8,5) or (B,8,8,5) array-like -> (B,57) numpy."""
        w = self.w
        h = torch.as_tensor(np.asarray(x), dtype=self.dtype)
        conv = F.conv3d if self.three_d else F.conv2d
        to_cf = (0, 4, 1, 2, 3) if self.three_d else (0, 3, 1, 2)       # channels-last -> channels-first for torch
        to_cl = (0, 2, 3, 4, 1) if self.three_d else (0, 2, 3, 1)
        h = h.permute(*to_cf)
        h = F.relu(conv(h, self.c1w, w["conv1_b"], padding=1))
        h = F.relu(conv(h, self.c2w, w["conv2_b"], padding=1))
        h = self._bn(h.permute(*to_cl), "bn1")                          # BatchNormalization on the last (channel) axis
        h = h.reshape(h.shape[0], -1)                                   # Flatten: row-major over (d,h,w,c)
        h = F.relu(h @ w["dense1_w"] + w["dense1_b"])
        h = self._bn(h, "bn2")
        return (h @ w["dense2_w"] + w["dense2_b"]).numpy()

    def predict(self, x, verbose=0):    # Keras-style entry (Tracking.py:732)
        return self.forward(x)
