"""Scene-level data parallelism over the GPUs of one node.

Scenes are fully independent (nothing in the reference's Tracking.py crosses
scenes), so each rank owns a contiguous block of scenes and runs the whole hot
path locally.  The only exchange is an all-gather of fixed-size per-track
summaries (`mmw_track_summary`, include/mmw.h) once per reporting interval --
RCCL over xGMI when the process group backend is "nccl", gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._lib import SUMMARY_DTYPE
from .shard import shard_range  # noqa: F401  (re-exported)

SUMMARY_WORDS = SUMMARY_DTYPE.itemsize // 4


def summaries_to_tensor(table: np.ndarray, device="cpu") -> torch.Tensor:
    """Structured [S, slots] table -> int32 tensor [S*slots, SUMMARY_WORDS] (bit view)."""
    flat = np.ascontiguousarray(table).view(np.int32).reshape(-1, SUMMARY_WORDS)
    return torch.from_numpy(flat.copy()).to(device)


def tensor_to_summaries(t: torch.Tensor, slots: int) -> np.ndarray:
    a = t.detach().cpu().numpy().astype(np.int32, copy=False)
    return np.ascontiguousarray(a).view(SUMMARY_DTYPE).reshape(-1, slots)


def all_gather_tables(local: torch.Tensor, counts=None) -> torch.Tensor:
    """All-gather the per-rank summary tensors ([rows_r, SUMMARY_WORDS] int32) into the global
    table ordered by rank (= by global scene id).  Ranks may own different row counts:
    shorter shards are padded for the collective and trimmed afterwards.
    `local` must be complete on torch's current stream: when `SceneBatch.track_table_dev` filled it, the context has
    to run on that stream (`SceneBatch.follow_torch_stream`) or be synchronised first."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows)
    all_rows = [int(r.item()) for r in all_rows]
    mx = max(all_rows)
    padded = local
    if local.shape[0] < mx:
        padded = torch.cat([local, torch.zeros((mx - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)])
    out = torch.empty((world * mx, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    if all(r == mx for r in all_rows):
        return out
    return torch.cat([out[r * mx: r * mx + all_rows[r]] for r in range(world)])
