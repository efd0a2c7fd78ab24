"""N > 1 path on CPU: two gloo ranks shard scenes, build their track-summary tables and
all-gather them into the global table (the only exchange step of the design)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_table(lo, hi, slots):
    from mmwave_msc_amd._lib import SUMMARY_DTYPE
    t = np.zeros((hi - lo, slots), dtype=SUMMARY_DTYPE)
    for s in range(lo, hi):
        for j in range(slots):
            t[s - lo, j]["scene"] = s
            t[s - lo, j]["slot"] = j
            t[s - lo, j]["alive"] = int((s + j) % 3 != 0)
            t[s - lo, j]["lifetime"] = 0.1 * j
            t[s - lo, j]["x"] = np.arange(9) + s
            t[s - lo, j]["keypoints"] = np.linspace(0, 1, 57) * (j + 1)
    return t


def _worker(rank, world, port, total, slots, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmwave_msc_amd.dist import all_gather_tables, shard_range, summaries_to_tensor, tensor_to_summaries
    lo, hi = shard_range(total, rank, world)
    local = summaries_to_tensor(_fake_table(lo, hi, slots))
    glob = all_gather_tables(local)
    tab = tensor_to_summaries(glob, slots)
    want = _fake_table(0, total, slots)
    ok = tab.shape == want.shape and all(np.array_equal(tab[n], want[n]) for n in want.dtype.names)
    q.put((rank, bool(ok), int(tab.shape[0])))
    dist.destroy_process_group()


def _run(total, slots):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == total for r in res)


def test_all_gather_track_tables_even_shards():
    _run(total=8, slots=4)


def test_all_gather_track_tables_uneven_shards():
    _run(total=7, slots=3)
