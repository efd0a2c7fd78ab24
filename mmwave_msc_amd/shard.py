"""Scene sharding arithmetic (no torch import: bench.py needs it before any GPU / process-group set-up)."""


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of the global scene ids owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
