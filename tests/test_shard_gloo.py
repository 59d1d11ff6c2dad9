"""CPU, world_size 2 over gloo: the N>1 path of the bench (connection sharding, barrier,
max-over-ranks timing, whole-job aggregate) without GPUs."""
import json
import os
import socket
import subprocess
import sys

import grpc_rdma_amd  # noqa: F401
from grpc_rdma_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import grpc_rdma_amd
from grpc_rdma_amd import shard
grp = shard.RankGroup(backend="gloo")
conns = shard.connections_for_rank(256, grp.rank, grp.world)
grp.barrier()
elapsed = 0.010 * (grp.rank + 1)          # the slower rank must set the job time
t = grp.max(elapsed)
total = grp.sum(len(conns) * 65536)       # bytes all ranks "moved"
if grp.rank == 0:
    print(json.dumps({"world": grp.world, "t": t, "total": total, "mine": [conns[0], conns[-1]]}))
grp.close()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_exact():
    for n, w in [(256, 8), (256, 1), (7, 2), (5, 8), (1, 4)]:
        seen = []
        for r in range(w):
            seen += shard.connections_for_rank(n, r, w)
        assert seen == list(range(n))
        for c in range(n):
            assert c in shard.connections_for_rank(n, shard.gpu_of_connection(c, n, w), w)
    assert [shard.gpu_of_connection(c, 256, 8) for c in (0, 31, 32, 255)] == [0, 0, 1, 7]


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    doc = json.loads(outs[0][0].strip().splitlines()[-1])
    assert doc["world"] == 2
    assert abs(doc["t"] - 0.020) < 1e-9            # MAX over ranks
    assert doc["total"] == 256 * 65536             # whole-job aggregate
    assert doc["mine"] == [0, 127]
