"""CPU, world_size 2 over gloo: the one collective of the data path (single-stream
fan-out, BASELINE configs[4]) -- partition on slice boundaries, scatter, reassemble."""
import json
import os
import socket
import subprocess
import sys

import grpc_rdma_amd  # noqa: F401
from grpc_rdma_amd import fanout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, hashlib
sys.path.insert(0, %r)
import torch
import grpc_rdma_amd
from grpc_rdma_amd import shard, fanout
grp = shard.RankGroup(backend="gloo")
# "arena" of an ingest rank: slices at 16-byte aligned offsets, i %% 251 payload
slices, off = [], 0
for n in [14, 16379] + [9, 16384] * 20 + [9, 9]:
    slices.append((off, n)); off = (off + n + 15) // 16 * 16
arena = torch.zeros(off, dtype=torch.uint8)
if grp.rank == 0:
    for o, n in slices:
        arena[o:o + n] = torch.arange(o, o + n, dtype=torch.int64).remainder(251).to(torch.uint8)
mine, my_slices = fanout.scatter_arena(grp, arena if grp.rank == 0 else torch.zeros(1, dtype=torch.uint8),
                                       slices if grp.rank == 0 else [], src=0)
payload = b"".join(bytes(mine[o:o + n].tolist()) for o, n in my_slices)
print(json.dumps({"rank": grp.rank, "n": len(my_slices), "bytes": len(payload),
                  "sha": hashlib.sha256(payload).hexdigest()}))
grp.close()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_covers_everything_in_order():
    slices = [(i * 32, 9 + (i % 3) * 1000) for i in range(50)]
    for w in (1, 2, 3, 8):
        parts = fanout.partition_slices(slices, w)
        assert len(parts) == w and [s for p in parts for s in p] == slices
        sizes = [sum(n for _, n in p) for p in parts]
        assert max(sizes) - min(sizes) <= 2 * 2009 or w == 1


def test_scatter_two_ranks_gloo(tmp_path):
    import hashlib
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    docs = sorted((json.loads(o[0].strip().splitlines()[-1]) for o in outs), key=lambda d: d["rank"])
    # expected: the same slices, same split, hashed locally
    slices, off = [], 0
    for n in [14, 16379] + [9, 16384] * 20 + [9, 9]:
        slices.append((off, n)); off = (off + n + 15) // 16 * 16
    parts = fanout.partition_slices(slices, 2)
    for r, d in enumerate(docs):
        exp = b"".join(bytes((i % 251) for i in range(o, o + n)) for o, n in parts[r])
        assert d["n"] == len(parts[r]) and d["bytes"] == len(exp)
        assert d["sha"] == hashlib.sha256(exp).hexdigest()
