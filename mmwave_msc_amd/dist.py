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


def exchange_row_counts(rows: int, device="cpu"):
    """Every rank's row count (one small all-gather + a host read: done ONCE per table shape by ShardedTracker, then cached)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(rows)]
    world = dist.get_world_size()
    t = torch.tensor([int(rows)], dtype=torch.int64, device=device)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    return [int(r.item()) for r in every]


def all_gather_tables(local: torch.Tensor, counts=None, out=None, pad=None, force_collective: bool = False) -> torch.Tensor:
    """All-gather the per-rank summary tensors ([rows_r, SUMMARY_WORDS] int32) into the global
    table ordered by rank (= by global scene id).  Ranks may own different row counts:
    shorter shards are padded for the collective and trimmed afterwards.
    `counts` = every rank's row count when the caller already knows them (no size exchange, no host read in front of the
    collective); `out` / `pad` = buffers of [world * max(counts), W] / [max(counts), W] to reuse.  `local` must be complete on
    the stream the collective runs on (torch's current stream): `ShardedTracker.gather_table` hands it over with an event
    (`SceneBatch.stream_wait`).  force_collective: run the collective even for a world of one (the one-rank RCCL test)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return local
    world = dist.get_world_size()
    all_rows = list(counts) if counts is not None else exchange_row_counts(local.shape[0], local.device)
    assert len(all_rows) == world and all_rows[dist.get_rank()] == local.shape[0], (all_rows, local.shape)
    mx = max(all_rows)
    padded = local
    if local.shape[0] < mx:
        if pad is None or pad.shape != (mx, local.shape[1]) or pad.device != local.device:
            pad = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]].copy_(local)
        padded = pad
    if out is None or out.shape != (world * mx, local.shape[1]) or out.device != local.device:
        out = torch.empty((world * mx, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    if all(r == mx for r in all_rows):
        return out
    return torch.cat([out[r * mx: r * mx + all_rows[r]] for r in range(world)])


def job_shard(n_scenes: int, rank: int, world: int, scaling: str = "strong"):
    """(lo, hi, total): the global scene ids [lo, hi) of `rank` and the job's scene count.  strong: `n_scenes` is the whole job,
    cut into contiguous blocks (BASELINE configs[2] / [4]); weak: every rank owns `n_scenes` scenes."""
    if scaling == "strong":
        lo, hi = shard_range(n_scenes, rank, world)
        return lo, hi, int(n_scenes)
    return rank * int(n_scenes), (rank + 1) * int(n_scenes), int(n_scenes) * int(world)


class ShardedTracker:
    """One rank's share of a job of many scenes, one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI):
    the rank-local `SceneBatch` for the scenes [lo, hi), optionally the `PosturePipeline` behind it, and the ONE exchange of the
    design -- the all-gather of the fixed-size track table (nothing in Tracking.py crosses scenes, so there is no
    data-path collective).

        st = ShardedTracker(cfg, n_scenes, max_pts, scaling="strong")      # after init_process_group, or single process
        st.step_dev(pts_ptr, n_ptr, dt_ptr, ...)                           # rank-local frames [hi - lo][max_pts][8]
        st.after_step()                                                    # estimate_posture, pipelined (if a model was given)
        table = st.gather_table(slots)                                     # [n_scenes_total * slots, SUMMARY_WORDS], by global scene id

    `batch_factory(cfg, n_local, max_pts, device)` builds the rank-local batch (default: `SceneBatch`); the CPU tests pass a stand-in
    with `track_table_host`, which is all `gather_table` needs off the GPU."""

    def __init__(self, cfg, n_scenes: int, max_pts: int, scaling: str = "strong", device=None, model=None, posture_cap=None,
                 rank=None, world=None, batch_factory=None, stream=None):
        init = dist.is_available() and dist.is_initialized()
        self.rank = int(rank if rank is not None else (dist.get_rank() if init else 0))
        self.world = int(world if world is not None else (dist.get_world_size() if init else 1))
        self.scaling = scaling
        self.lo, self.hi, self.n_total = job_shard(n_scenes, self.rank, self.world, scaling)
        self.S = self.hi - self.lo
        if self.S < 1:
            raise ValueError(f"rank {self.rank}: no scenes to own ({self.n_total} scenes over {self.world} ranks)")
        self.device = device
        if batch_factory is None:
            from .batch import SceneBatch
            batch_factory = lambda c, s, n, d: SceneBatch(c, s, n, device=0 if d is None else int(d))   # noqa: E731
        self.sb = batch_factory(cfg, self.S, int(max_pts), device)
        self._table = None
        self._gather = {}    # per slots: every rank's row count (exchanged once) and the collective's buffers
        self.pipe = None
        if stream is not None:
            self.sb.follow_torch_stream(stream)
        if model is not None:
            from .posture import PosturePipeline
            cap = int(posture_cap if posture_cap is not None else self.S * self.sb.track_cap)
            self.pipe = PosturePipeline(self.sb, model, cap, tracker_stream=stream)

    def step_dev(self, pts_ptr, n_ptr, dt_ptr, assoc_ptr=None, labels_ptr=None, dbn_ptr=None, f32: bool = False):
        (self.sb.step_dev_f32 if f32 else self.sb.step_dev)(pts_ptr, n_ptr, dt_ptr, assoc_ptr, labels_ptr, dbn_ptr)

    def after_step(self):
        if self.pipe is not None:
            self.pipe.after_step()

    def drain(self):
        if self.pipe is not None:
            self.pipe.drain()

    def gather_table(self, slots: int, force_collective: bool = False, reuse_out: bool = False) -> torch.Tensor:
        """Every rank's track summaries (`slots` per scene, scene ids global) all-gathered, ordered by global scene id.
        The table is written on the CONTEXT's stream (k_table); the collective runs on torch's current stream.  Both directions
        of the hand-over are events, whatever stream either side is on -- no host wait, no reliance on the caller having bound
        the context to torch's stream: the collective waits for k_table (`mmw_stream_wait`), and the NEXT call's k_table waits
        for the previous collective (`mmw_wait_stream`: it rewrites the table -- and this call the pad / output buffers -- the
        collective read).  Row counts are exchanged once per table shape and cached, the collective's buffers reused; the
        result is a fresh tensor unless `reuse_out` (then it is the cached output buffer, valid until the next call)."""
        slots = int(slots)
        if hasattr(self.sb, "track_table_dev") and getattr(self.sb, "h", None) is not None:
            dev = torch.device("cuda", self.sb.device)
            if self._table is None or self._table.shape[0] != self.S * slots:
                self._table = torch.zeros((self.S * slots, SUMMARY_WORDS), dtype=torch.int32, device=dev)
                self._gather = {}
            if hasattr(self.sb, "wait_stream"):
                self.sb.wait_stream(torch.cuda.current_stream(dev))   # (write-after-read: the previous gather still reads _table)
            self.sb.track_table_dev(self._table.data_ptr(), slots, scene_base=self.lo)
            if self.world > 1 and dist.get_backend() != "nccl":   # (gloo: host tensors; the table must have been written first)
                self.sb.synchronize()
                local = self._table.cpu()
            else:
                self.sb.stream_wait(torch.cuda.current_stream(dev))
                local = self._table
        else:
            local = summaries_to_tensor(self.sb.track_table_host(slots, scene_base=self.lo))
        if self.world == 1 and not force_collective:
            return local.clone() if (local is self._table and not reuse_out) else local
        g = self._gather.get(slots) if isinstance(getattr(self, "_gather", None), dict) else None
        if g is None or g["device"] != local.device:
            counts = exchange_row_counts(local.shape[0], local.device)
            mx = max(counts)
            g = dict(counts=counts, device=local.device,
                     out=torch.empty((len(counts) * mx, local.shape[1]), dtype=local.dtype, device=local.device),
                     pad=torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device) if local.shape[0] < mx else None)
            if not isinstance(getattr(self, "_gather", None), dict):
                self._gather = {}
            self._gather[slots] = g
        res = all_gather_tables(local, counts=g["counts"], out=g["out"], pad=g["pad"], force_collective=force_collective)
        if not reuse_out and res.data_ptr() == g["out"].data_ptr():
            res = res.clone()     # (a table kept from one call must not change under the caller at the next)
        return res

    def close(self):
        if self.pipe is not None:
            self.pipe.close()
            self.pipe = None
        if self.sb is not None and hasattr(self.sb, "close"):
            self.sb.close()
        self.sb = None


class LocalShardedTracker:
    """The same job in ONE process: G contexts (one per entry of `devices`; several may name the same GPU), each stepped from its
    own host thread -- the form SURVEY.md §8(e) allows beside one process per GPU (the C-ABI takes the device; a context is used
    from one thread at a time, different contexts from different threads concurrently).  `run(fn)` calls fn(g, shard) for every
    shard on its thread; `gather_table` concatenates the shards' tables (no collective: one address space)."""

    def __init__(self, cfg_factory, n_scenes: int, max_pts: int, devices, scaling: str = "strong"):
        from concurrent.futures import ThreadPoolExecutor
        from .batch import SceneBatch
        self.G = len(devices)
        self.shards = []
        for g, dev in enumerate(devices):
            lo, hi, total = job_shard(n_scenes, g, self.G, scaling)
            self.shards.append(dict(g=g, lo=lo, hi=hi, device=int(dev), sb=SceneBatch(cfg_factory(), hi - lo, int(max_pts), device=int(dev))))
        self.n_total = total
        self._pool = ThreadPoolExecutor(max_workers=self.G)

    def run(self, fn):
        """fn(g, shard_dict) on every shard's own thread; returns the results in shard order (exceptions propagate)."""
        return [f.result() for f in [self._pool.submit(fn, sh["g"], sh) for sh in self.shards]]

    def gather_table(self, slots: int) -> np.ndarray:
        tabs = self.run(lambda g, sh: sh["sb"].track_table_host(int(slots), scene_base=sh["lo"]))
        return np.concatenate(tabs, axis=0)

    def close(self):
        for sh in self.shards:
            sh["sb"].close()
        self._pool.shutdown()
