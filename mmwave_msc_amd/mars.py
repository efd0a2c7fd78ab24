"""MARS 19-keypoint regressor on PyTorch-ROCm (inference).

Architecture = the reference's Keras models (src/train.py): `define_CNN_3D`
(71-106) for 3-frame inputs (B,3,8,8,5) -- the variant `estimate_posture` feeds
(Tracking.py:718-734, Utils.py:517-520) -- and `define_CNN` (33-68) for
(B,8,8,5) when FB_FRAMES_BATCH == 0.  Dropout layers are identity at inference;
BatchNormalization uses Keras' default epsilon 1e-3.

Weights use Keras conventions on disk (dict / .npz, keys below) and are laid out
for the GPU at load time: conv kernels (kd,kh,kw,in,out) -> (out,in,kd,kh,kw);
Dense-1 rows are re-ordered from Keras' channels-last flatten (d,h,w,c) to the
channels-first flatten (c,d,h,w) so no activation transpose is needed, and both
BatchNorms are folded into the following Dense layer (exact in real arithmetic,
~1e-6 in fp32).  The reference's MARS.h5 is not in its repo; `.npz` files with the
same tensors drop in (see INTEGRATION.md for the h5 -> npz one-liner).

Keys: conv1_w conv1_b conv2_w conv2_b bn1_gamma bn1_beta bn1_mean bn1_var
      dense1_w dense1_b bn2_gamma bn2_beta bn2_mean bn2_var dense2_w dense2_b
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-3
N_KEYPOINTS = 57
SPLIT_SCALE = 2048.0   # 2^11: fp16 keeps 11 significant bits (csrc/k_mars.hip: kSplitScale)
FP16_SAFE_MAX = 6.0e4  # weights at or above this magnitude do not go through the split-fp16 arithmetic (fp16 max = 65 504)


def interleave_split(m32: torch.Tensor) -> torch.Tensor:
    """(R, K) fp32 -> (R, 2K) fp16: every value split a = hi + 2^-11 lo' and stored in runs of 32, [hi 0..31 | lo' 0..31 |
    hi 32..63 | ...] -- the operand layout of k_mars_dense1 (csrc/k_dense.hip); K must be a multiple of 32."""
    hi = m32.half()
    lo = ((m32 - hi.float()) * SPLIT_SCALE).half()
    r, k = m32.shape
    return torch.stack([hi.reshape(r, k // 32, 32), lo.reshape(r, k // 32, 32)], 2).reshape(r, 2 * k)


def deinterleave_split(a2: torch.Tensor):
    """(R, >= 2K) interleaved fp16 -> (hi, lo') as (R, K) views' copies (tests, diagnostics)."""
    r = a2.shape[0]
    v = a2.reshape(r, -1, 2, 32)
    return v[:, :, 0, :].reshape(r, -1), v[:, :, 1, :].reshape(r, -1)


from .marsweights import random_keras_weights  # noqa: E402,F401  (torch-free module: CPU-side tools import it without torch)


class MarsCNN(nn.Module):
    def __init__(self, frames: int = 3, arith: str = "f16x3"):
        """`arith` = how the conv pair and Dense-1 (99.7 % of the CNN's multiply-adds) are evaluated on the GPU:
        "f32"   on the fp32 matrix cores: the fused conv kernel k_mars_conv (3-frame model; torch convolutions for the
                single-frame one) and one fp32 GEMM (hipBLASLt through torch);
        "f16x3" every fp32 operand split as a = hi + 2^-11 lo' (hi, lo' fp16) and a.w = hi.w_hi + 2^-11 (hi.w_lo' + lo'.w_hi)
                on the fp16 matrix cores with fp32 accumulation (k_mars_conv16 + k_mars_dense1, both this package's kernels):
                every partial product is exact, the dropped lo'.lo' term is 2^-22 relative, and the result is CLOSER to the
                fp64 oracle than fp32 arithmetic (Dense-1 outputs: 1.2e-6 vs 2.5e-6 max error, measured in round 2).
                Not a reduced-precision mode: fp16 storage never holds a value that is not re-completed by its lo' half --
                WITHIN fp16's range: |a| < 65 504.  Weights outside it make from_keras_weights fall back to "f32"; an input
                or activation outside it raises the conv kernel's range word (range_overflow(): that sample's outputs are
                meaningless then), and predict() -- synchronous anyway -- computes such a batch again in fp32."""
        super().__init__()
        if arith not in ("f32", "f16x3", "torch"):
            raise ValueError(arith)   # ("torch": torch's own convolutions and GEMMs on the GPU, only ever on request -- see forward())
        self.arith = arith
        self.arith_fallback = None   # why a model asked for as "f16x3" runs "f32" (from_keras_weights: weights beyond fp16's range)
        self.range_fallbacks = 0     # predict() calls recomputed in fp32 because an input / activation left fp16's range
        self.range_recomputed = 0    # range_overflow() reads that found the device-side fp32 fix-up had run (3-frame model)
        self._sflags = None          # per-sample range verdicts of the split conv kernel (device int32, kept zero between calls)
        self._fix_scratch = None     # mmw_mars_range_fixup's device scratch
        self.frames = int(frames)
        self.three_d = self.frames > 1
        conv = nn.Conv3d if self.three_d else nn.Conv2d
        self.conv1 = conv(5, 16, 3, padding=1)
        self.conv2 = conv(16, 32, 3, padding=1)
        flat = (self.frames if self.three_d else 1) * 64 * 32
        hidden = 512 * (3 if self.three_d else 1)
        self.dense1 = nn.Linear(flat, hidden)   # BN1 folded in, rows in channels-first flatten order
        self.dense2 = nn.Linear(hidden, N_KEYPOINTS)  # BN2 folded in
        # hand-written fused conv pair (csrc/k_mars.hip): Keras-layout kernels and a Dense-1 whose rows follow Keras' own
        # (d,h,w,c) flatten order.  The fp16-split kernel (arith "f16x3") serves both models; the fp32-MFMA kernel
        # (arith "f32") exists for the 3-frame model only -- the single-frame model then uses torch's convolutions.
        self.use_hip_conv = self.frames in (3, 1)
        self.use_hip_conv_f32 = self.frames == 3
        kt = 27 if self.three_d else 9
        self.register_buffer("k_w1", torch.zeros(kt * 5 * 16))
        self.register_buffer("k_b1", torch.zeros(16))
        self.register_buffer("k_w2", torch.zeros(kt * 16 * 32))
        self.register_buffer("k_b2", torch.zeros(32))
        self.dense1_dhwc = nn.Linear(flat, hidden)
        # the same matrix split for the fp16 matrix cores, transposed (K contiguous) with the halves interleaved in runs of 32,
        # [hi 0..31 | lo' 0..31 | hi 32..63 | ...]: the layout k_mars_dense1 stages (csrc/k_dense.hip)
        self.register_buffer("d1_w2_t", torch.zeros((hidden, 2 * flat), dtype=torch.float16))
        # one word the split conv kernel raises when it splits an input or activation outside fp16's range (range_overflow())
        self.register_buffer("range_flag", torch.zeros(1, dtype=torch.int32))
        for p in self.parameters():
            p.requires_grad_(False)

    @classmethod
    def from_keras_weights(cls, w: dict, arith: str = "f16x3") -> "MarsCNN":
        three_d = np.asarray(w["conv1_w"]).ndim == 5
        flat = np.asarray(w["dense1_w"]).shape[0]
        frames = flat // (64 * 32) if three_d else 1
        m = cls(frames if three_d else 1, arith)
        f64 = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
        perm = (4, 3, 0, 1, 2) if three_d else (3, 2, 0, 1)
        m.conv1.weight.copy_(torch.from_numpy(f64["conv1_w"].transpose(perm).copy()).float())
        m.conv1.bias.copy_(torch.from_numpy(f64["conv1_b"]).float())
        m.conv2.weight.copy_(torch.from_numpy(f64["conv2_w"].transpose(perm).copy()).float())
        m.conv2.bias.copy_(torch.from_numpy(f64["conv2_b"]).float())
        # BN1 (per channel c) folded into dense1, rows re-ordered (d,h,w,c) -> (c,d,h,w)
        a1 = f64["bn1_gamma"] / np.sqrt(f64["bn1_var"] + BN_EPS)
        c1 = f64["bn1_beta"] - a1 * f64["bn1_mean"]
        spatial = flat // 32
        w1 = f64["dense1_w"].reshape(spatial, 32, -1)            # [s, c, out]
        b1 = f64["dense1_b"] + np.einsum("c,sco->o", c1, w1)
        w1 = (w1 * a1[None, :, None]).transpose(1, 0, 2).reshape(flat, -1)  # [(c,s), out]
        m.k_w1.copy_(torch.from_numpy(f64["conv1_w"].reshape(-1)).float())
        m.k_b1.copy_(torch.from_numpy(f64["conv1_b"]).float())
        m.k_w2.copy_(torch.from_numpy(f64["conv2_w"].reshape(-1)).float())
        m.k_b2.copy_(torch.from_numpy(f64["conv2_b"]).float())
        wk = (f64["dense1_w"].reshape(spatial, 32, -1) * a1[None, :, None]).reshape(flat, -1)  # Keras row order kept
        m.dense1_dhwc.weight.copy_(torch.from_numpy(wk.T.copy()).float())
        m.dense1_dhwc.bias.copy_(torch.from_numpy(b1).float())
        w32 = torch.from_numpy(wk.copy()).float()                    # (K, N), what the fp32 GEMM multiplies with
        m.d1_w2_t.copy_(interleave_split(w32.t().contiguous()))
        # fp16's range is the split arithmetic's: a folded weight outside it would become inf in its hi half
        wmax = max(float(np.abs(f64[k]).max()) for k in ("conv1_w", "conv2_w", "conv1_b", "conv2_b"))
        if m.arith == "f16x3" and max(wmax, float(w32.abs().max())) >= FP16_SAFE_MAX:
            m.arith = m.fp32_arith()
            m.arith_fallback = (f"a weight of magnitude {max(wmax, float(w32.abs().max())):.3g} does not fit fp16: "
                                + ("fp32 matrix cores instead" if m.arith == "f32" else "torch's fp32 kernels instead (the single-frame model has no fp32 HIP kernel)"))
        m.dense1.weight.copy_(torch.from_numpy(w1.T.copy()).float())
        m.dense1.bias.copy_(torch.from_numpy(b1).float())
        a2 = f64["bn2_gamma"] / np.sqrt(f64["bn2_var"] + BN_EPS)
        c2 = f64["bn2_beta"] - a2 * f64["bn2_mean"]
        w2 = f64["dense2_w"]
        m.dense2.weight.copy_(torch.from_numpy((w2 * a2[:, None]).T.copy()).float())
        m.dense2.bias.copy_(torch.from_numpy(f64["dense2_b"] + c2 @ w2).float())
        return m

    @classmethod
    def from_npz(cls, path: str) -> "MarsCNN":
        z = np.load(path)
        return cls.from_keras_weights({k: z[k] for k in z.files})

    @classmethod
    def from_h5(cls, path: str) -> "MarsCNN":
        """`keras.models.load_model(P_MODEL_PATH)` (offline_main.py:33) for the weights: reads the Keras `.h5` the
        reference's train.py:252 writes, without keras or h5py (h5weights.py)."""
        from .h5weights import load_keras_h5
        return cls.from_keras_weights(load_keras_h5(path))

    @classmethod
    def load(cls, path: str) -> "MarsCNN":
        """By extension: `.h5` / `.hdf5` (Keras) or `.npz` (train.py of this package)."""
        return cls.from_h5(path) if path.lower().endswith((".h5", ".hdf5")) else cls.from_npz(path)

    def _hip_convs(self, x: torch.Tensor) -> torch.Tensor:
        """Conv3D+ReLU twice in one HIP kernel on the fp32 matrix cores (mmw_mars_conv3d) -> (B, 6144) in (d,h,w,c) order."""
        from . import _lib
        L = _lib.load()
        x = x.contiguous()
        out = torch.empty((x.shape[0], 6144), dtype=torch.float32, device=x.device)
        rc = L.mmw_mars_conv3d(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), self.k_w1.data_ptr(),
                               self.k_b1.data_ptr(), self.k_w2.data_ptr(), self.k_b2.data_ptr(), out.data_ptr(), x.shape[0])
        if rc != 0:
            raise _lib.MmwError(rc, (L.mmw_last_error(None) or b"").decode())
        return out

    # the activation's row stride: 2 * flat values + 256: whole rows then start 24.25 / 8.25 KB apart instead of a multiple
    # of 4 KB, which spreads a tile's rows over the L2 channels (k_mars_dense1 1.144 -> 1.093 ms at 18 k rows)
    ROW_PAD = 256

    def _hip_convs_split(self, x: torch.Tensor, sflags: torch.Tensor | None = None) -> torch.Tensor:
        """The conv pair on the fp16 matrix cores with split operands (mmw_mars_conv_split), the activation already split
        for Dense-1: (B, 2 * flat) fp16, halves interleaved in runs of 32 (interleave_split), flat = frames * 2048 in (d,h,w,c)
        order.  The result is a VIEW: its rows are 2 * flat + ROW_PAD apart and the storage holds whole 256-row tiles (the rows
        past B are never written: Dense-1's kernel reads them, into rows nobody reads)."""
        from . import _lib
        L = _lib.load()
        x = x.contiguous()
        flat = self.frames * 2048
        rows = (x.shape[0] + 255) // 256 * 256
        ld = 2 * flat + self.ROW_PAD
        out = torch.empty((rows, ld), dtype=torch.float16, device=x.device)
        # the list the fp32 fix-up behind Dense-2 works off (forward): [count, taken, 64 sample indices] -- the model's own, or the
        # caller's (posture.PosturePipeline keeps one per frame in flight and runs the fix-up on another stream)
        if sflags is not None:
            sflags = sflags.data_ptr()
        elif self.has_range_fixup():
            if self._sflags is None or self._sflags.device != x.device:
                self._sflags = torch.zeros((2 + 64,), dtype=torch.int32, device=x.device)
            sflags = self._sflags.data_ptr()
        rc = L.mmw_mars_conv_split(torch.cuda.current_stream(x.device).cuda_stream, self.frames, x.data_ptr(), self.k_w1.data_ptr(),
                                   self.k_b1.data_ptr(), self.k_w2.data_ptr(), self.k_b2.data_ptr(), out.data_ptr(), ld, x.shape[0],
                                   self.range_flag.data_ptr(), sflags)
        if rc != 0:
            raise _lib.MmwError(rc, (L.mmw_last_error(None) or b"").decode())
        return out[: x.shape[0], : 2 * flat]

    def _dense1_split(self, a2: torch.Tensor) -> torch.Tensor:
        """relu(bias + hi.W_hi + 2^-11 (hi.W_lo' + lo'.W_hi)) in one kernel (mmw_mars_dense1_split, csrc/k_dense.hip) on the view
        _hip_convs_split returns (any other (B, 2 flat) interleaved fp16 tensor is copied into a padded buffer first)."""
        from . import _lib
        L = _lib.load()
        B, k2 = a2.shape
        rows = (B + 255) // 256 * 256
        ld = a2.stride(0)
        if a2.stride(1) != 1 or ld < k2 or (ld & 7) or a2.untyped_storage().nbytes() < (a2.storage_offset() + (rows - 1) * ld + k2) * 2:
            buf = torch.zeros((rows, k2 + self.ROW_PAD), dtype=torch.float16, device=a2.device)
            buf[:B, :k2] = a2
            a2, ld = buf[:B, :k2], k2 + self.ROW_PAD
        n = self.d1_w2_t.shape[0]
        h = torch.empty((rows, n), dtype=torch.float32, device=a2.device)
        rc = L.mmw_mars_dense1_split(torch.cuda.current_stream(a2.device).cuda_stream, a2.data_ptr(), ld, self.d1_w2_t.data_ptr(), k2,
                                     self.dense1_dhwc.bias.data_ptr(), h.data_ptr(), rows, k2 // 2, n)
        if rc != 0:
            raise _lib.MmwError(rc, (L.mmw_last_error(None) or b"").decode())
        return h[:B]

    def has_range_fixup(self) -> bool:
        """The split arithmetic repairs itself (3-frame model): samples that left fp16's range are recomputed in fp32 on the device."""
        return self.use_hip_conv_f32

    def new_fixup_list(self, device) -> torch.Tensor:
        """A fix-up list of its own for a caller that keeps several frames in flight (forward(..., fixup=False, sflags=...) then
        range_fixup(..., sflags=...) for the same frame, on any stream ordered behind that forward)."""
        return torch.zeros((2 + 64,), dtype=torch.int32, device=device)

    def range_fixup(self, x: torch.Tensor, kp: torch.Tensor, sflags: torch.Tensor):
        """The repair forward(..., fixup=False, sflags=sflags) left out, on torch's current stream: rows of kp[n][57] whose samples
        left fp16's range recomputed in fp32 from x[n] (the same feature tensors)."""
        if self.has_range_fixup():
            self._range_fixup(x, kp, sflags)

    def _range_fixup(self, x: torch.Tensor, kp: torch.Tensor, sflags: torch.Tensor | None = None):
        """mmw_mars_range_fixup behind a split-arithmetic forward: the flagged samples' rows of kp recomputed in Keras' fp32 -- no host
        wait; a batch without such samples pays four empty launches."""
        from . import _lib
        L = _lib.load()
        if self._fix_scratch is None or self._fix_scratch.device != x.device:
            self._fix_scratch = torch.empty((512 + 64 * (960 + 6144 + 1536 + 57) * 4,), dtype=torch.uint8, device=x.device)   # MMW_RANGE_FIXUP_SCRATCH
        w1 = self.dense1_dhwc.weight
        rc = L.mmw_mars_range_fixup(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), (sflags if sflags is not None else self._sflags).data_ptr(), x.shape[0],
                                    self.k_w1.data_ptr(), self.k_b1.data_ptr(), self.k_w2.data_ptr(), self.k_b2.data_ptr(), w1.data_ptr(), w1.stride(0),
                                    self.dense1_dhwc.bias.data_ptr(), self.dense2.weight.data_ptr(), self.dense2.bias.data_ptr(),
                                    self._fix_scratch.data_ptr(), kp.data_ptr(), self.range_flag.data_ptr())
        if rc != 0:
            raise _lib.MmwError(rc, (L.mmw_last_error(None) or b"").decode())

    SMALL_BATCH = 64   # mmw_mars_head_small's limit

    def has_small_path(self) -> bool:
        return self.use_hip_conv_f32

    def forward_small(self, x: torch.Tensor) -> torch.Tensor:
        """A handful of samples (one scene's tracks: the drop-in's estimate_posture): the fp32 conv kernel (mmw_mars_conv3d) and
        the thin Dense-1 / Dense-2 kernels of mmw_mars_head_small -- Keras' own fp32 arithmetic, the weight matrix cut over the
        whole chip instead of one band of tiles (~100 us less for 2 rows), no fp16 range to watch.  3-frame model, B <= 64."""
        from . import _lib
        if not (x.is_cuda and x.dtype == torch.float32 and self.use_hip_conv_f32 and x.shape[0] <= self.SMALL_BATCH):
            raise ValueError("MarsCNN.forward_small: a CUDA fp32 batch of at most 64 samples of the 3-frame model")
        L = _lib.load()
        with torch.cuda.device(x.device):
            act = self._hip_convs(x)
            B = act.shape[0]
            w1 = self.dense1_dhwc.weight
            hidden = torch.empty((B, w1.shape[0]), dtype=torch.float32, device=x.device)
            kp = torch.empty((B, N_KEYPOINTS), dtype=torch.float32, device=x.device)
            rc = L.mmw_mars_head_small(torch.cuda.current_stream(x.device).cuda_stream, act.data_ptr(), act.stride(0), w1.data_ptr(), w1.stride(0),
                                       self.dense1_dhwc.bias.data_ptr(), self.dense2.weight.data_ptr(), self.dense2.bias.data_ptr(),
                                       hidden.data_ptr(), kp.data_ptr(), B, w1.shape[1], w1.shape[0])
            if rc != 0:
                raise _lib.MmwError(rc, (L.mmw_last_error(None) or b"").decode())
        return kp

    def range_overflow(self, clear: bool = True) -> bool:
        """True when a split-arithmetic forward since the last call left some sample's keypoints MEANINGLESS: an input or
        activation outside fp16's range that was not repaired.  The 3-frame model repairs such samples itself (forward():
        mmw_mars_range_fixup recomputes them in fp32 on the device), so for it this only reports a batch with more than 64 of
        them; the single-frame model has no fp32 kernel and reports every one.  `range_recomputed` says how often the repair
        ran.  Reads one device word: synchronises."""
        word = int(self.range_flag.item())
        if word and clear:
            self.range_flag.zero_()
        if word & 1:
            self.range_recomputed += 1
        return bool(word & 2) if self.has_range_fixup() else bool(word & 1)

    def fp32_arith(self) -> str:
        """What replaces the split arithmetic where it cannot be used (weights / activations beyond fp16's range): the fp32
        matrix-core kernel for the 3-frame model, torch's fp32 kernels for the single-frame one (no fp32 HIP kernel exists)."""
        return "f32" if self.use_hip_conv_f32 else "torch"

    def forward(self, x: torch.Tensor, arith: str | None = None, fixup: bool = True, sflags: torch.Tensor | None = None,
                out: torch.Tensor | None = None) -> torch.Tensor:
        """x: (B,3,8,8,5) [or (B,8,8,5)] channels-last fp32, as mmw_features writes it.  `arith` overrides the model's for
        this call; `fixup=False, sflags=<new_fixup_list()>` leaves the fp32 repair of out-of-range samples to the caller's
        range_fixup() (another stream, later); `out` (B, 57) fp32 contiguous receives the keypoints of the split-arithmetic path
        directly (Dense-2 writes there: no copy behind it).  A CUDA tensor runs this package's HIP kernels or raises: torch's own convolutions take a GPU batch only
        when arith = "torch" was asked for (CPU tensors always take the torch path -- training exports, CPU tests)."""
        arith = arith or self.arith
        if x.is_cuda and arith != "torch":
            if x.dtype != torch.float32:
                raise TypeError(f"MarsCNN: the HIP kernels take fp32 feature tensors (what mmw_features writes), got {x.dtype}; "
                                "convert the batch, or ask for torch's kernels with arith='torch'")
            if arith == "f32" and not self.use_hip_conv_f32:
                raise ValueError("MarsCNN: the single-frame model has no fp32 HIP kernel; arith='f16x3' (default) runs k_mars_conv16, "
                                 "arith='torch' torch's convolutions")
            with torch.cuda.device(x.device):
                if arith == "f16x3":
                    xc = x.contiguous()
                    h = self._dense1_split(self._hip_convs_split(xc, sflags))
                    if out is not None:
                        if not (out.is_contiguous() and out.dtype == torch.float32 and out.is_cuda and tuple(out.shape) == (xc.shape[0], N_KEYPOINTS)):
                            raise ValueError(f"MarsCNN.forward(out=...): needs a contiguous fp32 CUDA tensor of shape ({xc.shape[0]}, {N_KEYPOINTS}), "
                                             f"got {out.dtype} {tuple(out.shape)} contiguous={out.is_contiguous()}")
                        kp = torch.addmm(self.dense2.bias, h, self.dense2.weight.t(), out=out)
                    else:
                        kp = self.dense2(h).contiguous()
                    if fixup and self.has_range_fixup():
                        self._range_fixup(xc, kp, sflags)   # samples outside fp16's range: their rows again, in fp32 (Keras' arithmetic)
                    return kp
                if out is not None:   # (a caller that passes `out` does not look at the return value: never leave it stale)
                    raise ValueError("MarsCNN.forward(out=...) is the split-arithmetic path's (arith='f16x3'); this call runs arith=%r" % arith)
                h = F.relu(self.dense1_dhwc(self._hip_convs(x)))
            return self.dense2(h)
        if out is not None:
            raise ValueError("MarsCNN.forward(out=...) is the split-arithmetic GPU path's; this call runs torch's operators")
        if self.three_d:
            h = x.permute(0, 4, 1, 2, 3)
        else:
            h = x.permute(0, 3, 1, 2)
        h = F.relu(self.conv1(h))
        h = F.relu(self.conv2(h))
        h = F.relu(self.dense1(h.flatten(1)))
        return self.dense2(h)

    @torch.no_grad()
    def predict_numpy(self, feat: np.ndarray) -> np.ndarray:
        """Keras' model.predict: synchronous, so the range check costs nothing here -- a batch that left fp16's range under the
        split arithmetic is computed again on the fp32 path (Keras' fp32 has no such limit)."""
        dev = next(self.parameters()).device
        x = torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32)).to(dev)
        out = self(x).float().cpu().numpy()
        if self.arith == "f16x3" and x.is_cuda and self.range_overflow():
            self.range_fallbacks += 1
            out = self(x, arith=self.fp32_arith()).float().cpu().numpy()
        return out

    def predict(self, feat, verbose=0):  # Keras-style entry used by estimate_posture (Tracking.py:732)
        return self.predict_numpy(np.asarray(feat))
