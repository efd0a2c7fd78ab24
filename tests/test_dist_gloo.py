"""N > 1 path on CPU: two gloo ranks shard scenes, build their track-summary tables and
all-gather them into the global table (the only exchange step of the design)."""
import os
import socket

import numpy as np
import pytest
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


def _sharded_worker(rank, world, port, total, slots, scaling, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmwave_msc_amd.dist import ShardedTracker, job_shard, tensor_to_summaries

    class Standin:   # the rank-local batch off the GPU: its table is all gather_table needs
        def __init__(self, cfg, n_local, max_pts, device):
            self.S, self.steps = n_local, 0

        def step_dev(self, *a):
            self.steps += 1

        def track_table_host(self, n_slots, scene_base=0):
            return _fake_table(scene_base, scene_base + self.S, n_slots)

    st = ShardedTracker(None, total, 64, scaling=scaling, batch_factory=Standin)
    lo, hi, n_total = job_shard(total, rank, world, scaling)
    ok = (st.lo, st.hi, st.n_total, st.S) == (lo, hi, n_total, hi - lo) and st.rank == rank and st.world == world
    st.step_dev(0, 0, 0)
    st.after_step()      # (no model: nothing to do)
    tab = tensor_to_summaries(st.gather_table(slots), slots)
    want = _fake_table(0, n_total, slots)
    ok = ok and st.sb.steps == 1 and tab.shape == want.shape and all(np.array_equal(tab[n], want[n]) for n in want.dtype.names)
    st.close()
    q.put((rank, bool(ok), int(tab.shape[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("total,scaling,rows", [(9, "strong", 9), (5, "weak", 10)])
def test_sharded_tracker_object_over_two_gloo_ranks(total, scaling, rows):
    """mmwave_msc_amd.dist.ShardedTracker -- what `bench.py --gpus N` is a client of -- with a stand-in for the rank-local batch:
    shard arithmetic (strong: uneven blocks; weak: `total` scenes per rank), stepping, and the all-gather of the track table
    into the global table ordered by scene id."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, total, 3, scaling, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), res
    assert all(r[2] == rows for r in res), res


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout, whatever the libraries print
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks_strong_scaling():
    """`python bench.py --gpus 2` without a launcher: two ranks are spawned before any GPU call, the 4096 scenes of
    BASELINE configs[2] are sharded over them (strong scaling is the default), the gathered table is the global one."""
    j = _bench("--gpus", "2", "--dry-run")
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["scaling"] == "strong"
    assert j["scenes_total"] == 4096 and j["scenes_rank0"] == 2048 and j["gathered_table_rows"] == 4096 and j["gather_ok"]


def test_bench_weak_scaling_and_uneven_strong_shards():
    j = _bench("--gpus", "2", "--dry-run", "--scaling", "weak", "--scenes", "6")
    assert j["scenes_total"] == 12 and j["scenes_rank0"] == 6 and j["gather_ok"]
    j = _bench("--gpus", "3", "--dry-run", "--scenes", "10")
    assert j["scenes_total"] == 10 and j["scenes_rank0"] == 4 and j["gathered_table_rows"] == 10 and j["gather_ok"]


def test_bench_under_an_external_launcher_uses_the_given_ranks():
    """The driver's form: torch.distributed.run sets RANK/WORLD_SIZE/MASTER_*; bench.py must not spawn again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--scenes", "64"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [json.loads(ln) for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(js) == 1 and js[0]["n_ranks_seen"] == 2 and js[0]["scenes_rank0"] == 32 and js[0]["gather_ok"]


def test_bench_times_every_kernel_and_never_two_in_a_step():
    """bench.profiled_kernels: the thin HIP-event sample of the timed region (an event pair holds the stream for 10-20 us: two
    pairs in every 4th step cost 3.3 % of the headline, profiles/NOTEBOOK.md round 5).  Whatever the number of timed steps the
    driver asks for: never two pairs in one step, k_track -- the roofline kernel -- most often, each of the other three kernels
    of a step at least once, and from the default 40 steps on at most one step in five carries a pair."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for K in (8, 12, 20, 40, 80, 150, 1000):
        seen = {}
        for i in range(K):
            pk = bench.profiled_kernels(i, K)
            assert len(pk) <= 1, (K, i, pk)
            for k in pk:
                seen[k] = seen.get(k, 0) + 1
        assert set(seen) == {0, *bench.OTHER_KERNELS}, (K, seen)          # K_TRACK + K_PREDICT, K_DBSCAN, K_POST
        assert seen[0] >= max(seen[k] for k in bench.OTHER_KERNELS), (K, seen)
        if K >= 40:   # (a short run samples more densely so that every kernel is seen; from the default 40 steps on: one step in five at most)
            assert sum(seen.values()) <= K // 5, (K, seen)
