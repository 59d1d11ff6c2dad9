"""Single-large-stream fan-out (BASELINE.json configs[4]).

One stream is inherently serial at ingest (one ring, one head).  After the ingest GPU
has decoded it into its receive arena, the arena is cut into `world` contiguous byte
ranges (on delivered-slice boundaries) and rebalanced with ONE scatter step:
torch.distributed.scatter over RCCL, i.e. grouped ncclSend/ncclRecv, one xGMI link
per destination GPU (7 links x ~153 GB/s in parallel; not a ring collective, which
would be bound by a single link).  This is the only collective on the data path."""
import torch


def partition_slices(slices, world):
    """slices: [(offset, length)] in arena order -> `world` lists of slices whose byte
    ranges are contiguous and as even as slice boundaries allow."""
    total = sum(n for _, n in slices)
    parts, cur, acc, target = [], [], 0, total / float(world)
    for s in slices:
        cur.append(s)
        acc += s[1]
        if len(parts) < world - 1 and acc >= target * (len(parts) + 1):
            parts.append(cur)
            cur = []
    parts.append(cur)
    while len(parts) < world:
        parts.append([])
    return parts


def byte_range(part):
    """(start, end) of the arena bytes spanned by a list of slices (16-byte slice padding
    in between travels along)."""
    if not part:
        return (0, 0)
    return (part[0][0], part[-1][0] + part[-1][1])


def scatter_arena(group, arena, slices, src=0):
    """arena: uint8 tensor on the ingest rank (the stream job's destination buffer),
    slices: delivered slices [(offset, length)] (only needed on `src`).
    Returns (my_tensor, my_slices) where my_slices are rebased to my_tensor."""
    dist = group.dist
    world, rank = group.world, group.rank
    if dist is None or world == 1:
        return arena, list(slices)
    device = arena.device
    if rank == src:
        parts = partition_slices(slices, world)
        ranges = [byte_range(p) for p in parts]
        meta = torch.tensor([[a, b] for a, b in ranges], dtype=torch.int64, device=device)
    else:
        parts, ranges = None, None
        meta = torch.empty((world, 2), dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src)
    ranges = [(int(a), int(b)) for a, b in meta.tolist()]
    width = max(b - a for a, b in ranges)
    width = max(1, (width + 15) // 16 * 16)
    mine = torch.empty(width, dtype=torch.uint8, device=device)
    if rank == src:
        chunks = []
        for a, b in ranges:
            c = torch.zeros(width, dtype=torch.uint8, device=device)
            c[:b - a] = arena[a:b]
            chunks.append(c)
        dist.scatter(mine, scatter_list=chunks, src=src)
    else:
        dist.scatter(mine, scatter_list=None, src=src)
    a, b = ranges[rank]
    # slice tables travel as one small broadcast of (offset, length) rows per rank
    if rank == src:
        counts = torch.tensor([len(p) for p in parts], dtype=torch.int64, device=device)
    else:
        counts = torch.empty(world, dtype=torch.int64, device=device)
    dist.broadcast(counts, src=src)
    nmax = int(counts.max().item())
    table = torch.zeros((world, max(1, nmax), 2), dtype=torch.int64, device=device)
    if rank == src:
        for r, p in enumerate(parts):
            for k, (o, n) in enumerate(p):
                table[r, k, 0], table[r, k, 1] = o - ranges[r][0], n
    dist.broadcast(table, src=src)
    my_slices = [(int(o), int(n)) for o, n in table[rank, :int(counts[rank].item())].tolist()]
    return mine[:b - a], my_slices
