"""Single-large-stream fan-out (BASELINE.json configs[4]).

One stream is inherently serial at ingest (one ring, one head).  After the ingest GPU
has decoded it into its receive arena, the arena is cut into `world` contiguous byte
ranges (on delivered-slice boundaries) and rebalanced with ONE grouped point-to-point
step: torch.distributed.batch_isend_irecv over RCCL = ncclGroupStart / ncclSend x (world-1)
/ ncclGroupEnd on the ingest rank, one ncclRecv on every other rank -- one xGMI link per
destination GPU (7 links x ~153 GB/s in parallel; not a ring collective, which would be
bound by a single link).  The sends are VIEWS of the arena with their exact lengths: nothing
is copied or padded on the source (a scatter needs equal-sized chunks, which cost
2 x stream bytes of extra HBM traffic in round 1).  This is the only collective on the data path."""
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


def stream_checksum(t, slices, start=0):
    """Checksum of the bytes `slices` [(offset, length)] of the uint8 tensor t hold, taken as bytes start, start + 1, ...
    of one stream: (sum of the bytes, sum of byte i x (i mod 65521 + 1)) -- the second one moves when bytes swap places
    or a share lands at another position.  Sums over disjoint shares of a stream add up to the whole stream's, so the
    ranks of a fan-out can each check their share and one all-reduce checks the concatenation.  Computed where t
    lives (one gather, two reductions); exact in int64 up to 2^38 bytes."""
    if not slices:
        return (0, 0)
    flat = torch.cat([t[o:o + n] for o, n in slices]).to(torch.int64)
    w = (torch.arange(start, start + flat.numel(), dtype=torch.int64, device=flat.device) % 65521) + 1
    return (int(flat.sum().item()), int((flat * w).sum().item()))


def scatter_arena(group, arena, slices, src=0, return_start=False):
    """arena: uint8 tensor on the ingest rank (the stream job's destination buffer),
    slices: delivered slices [(offset, length)] (only needed on `src`).
    Returns (my_tensor, my_slices) where my_slices are rebased to my_tensor; with return_start also the position of my
    share's first byte in the delivered stream (what stream_checksum takes)."""
    dist = group.dist
    world, rank = group.world, group.rank
    if dist is None or world == 1:
        return (arena, list(slices), 0) if return_start else (arena, list(slices))
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
    a, b = ranges[rank]
    ops = []
    if rank == src:
        mine = arena[a:b]  # the ingest rank keeps its share in place
        for r, (ra, rb) in enumerate(ranges):
            if r != src and rb > ra:
                ops.append(dist.P2POp(dist.isend, arena[ra:rb], r))
    else:
        mine = torch.empty(max(0, b - a), dtype=torch.uint8, device=device)
        if b > a:
            ops.append(dist.P2POp(dist.irecv, mine, src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    # slice tables travel as one small broadcast of (offset, length) rows per rank
    if rank == src:
        counts = torch.tensor([len(p) for p in parts], dtype=torch.int64, device=device)
    else:
        counts = torch.empty(world, dtype=torch.int64, device=device)
    dist.broadcast(counts, src=src)
    nmax = int(counts.max().item())
    if rank == src:  # built on the host in one go (thousands of rows), then one broadcast
        rows = [[[o - ranges[r][0], n] for o, n in p] + [[0, 0]] * (max(1, nmax) - len(p)) for r, p in enumerate(parts)]
        table = torch.tensor(rows, dtype=torch.int64, device=device).reshape(world, max(1, nmax), 2)
    else:
        table = torch.zeros((world, max(1, nmax), 2), dtype=torch.int64, device=device)
    dist.broadcast(table, src=src)
    my_slices = [(int(o), int(n)) for o, n in table[rank, :int(counts[rank].item())].tolist()]
    if return_start:
        start = int(sum(int(table[r, :int(counts[r].item()), 1].sum().item()) for r in range(rank)))
        return mine, my_slices, start
    return mine, my_slices
