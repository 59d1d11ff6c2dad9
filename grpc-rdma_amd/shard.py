"""Connection sharding across the GPUs of a node.

Connections are independent units (one pair, one ring, one credit word each;
pair.h:323-326, poller.h:65 are the only shared tables in the reference), so the
data path needs no collective: connection c lives on GPU c // (n_conns / world).
The ranks meet only in the barrier / max-over-ranks timing of the bench contract."""
import os


def gpu_of_connection(conn, n_conns, world):
    """BASELINE config 4: 256 connections on 8 GPUs -> 32 per GPU, contiguous blocks."""
    per = -(-n_conns // world)
    return min(conn // per, world - 1)


def connections_for_rank(n_conns, rank, world):
    per = -(-n_conns // world)
    lo = min(rank * per, n_conns)
    hi = min(lo + per, n_conns)
    return list(range(lo, hi))


def rank_env():
    """RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class RankGroup:
    """Barrier + max-over-ranks reduction; backend nccl (= RCCL) on GPUs, gloo on CPU."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = rank_env()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def sum_int(self, value):
        """Exact sum of one integer per rank (< 2^62 each): all-gathered as int64, added as Python ints."""
        if self.dist is None:
            return int(value)
        import torch
        dev = self.device if self.device is not None else "cpu"
        mine = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        every = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(every, mine)
        return sum(int(t.item()) for t in every)

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
