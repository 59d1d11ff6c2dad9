"""CPU, world_size 2 over gloo: the N > 1 path of the bench EXECUTING the data plane, not only its arithmetic.

Each rank is a process of its own (as `python -m torch.distributed.run --nproc-per-node N bench.py` makes them) and
runs REAL streaming jobs through the C ABI -- the product sources on the wave emulator (oracle/_build/libgrdma_emu.so,
tests/cc/build_emu.sh; on the GPU box the same code path is libgrdma_amd.so on cuda:LOCAL_RANK):

  * the connections of BASELINE configs[3] are sharded by grpc_rdma_amd.shard (contiguous blocks, no data-path
    collective): every rank runs ONE job over its share -- several links per launch -- and checks its delivered
    byte streams; the ranks meet in the barrier, the max-over-ranks time and the whole-job byte sum of the contract;
  * the single-stream fan-out of configs[4]: rank 0's delivered arena -- the output of its own job, not a synthetic
    tensor -- goes through fanout.scatter_arena (one grouped send / recv), and the concatenation of what the ranks hold
    afterwards is the framed message stream, byte for byte."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

from oracle import pyorc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "oracle", "_build", "libgrdma_emu.so")

WORKER = r'''
import hashlib, json, os, random, sys, time
sys.path.insert(0, %(root)r)
import torch
import grpc_rdma_amd as g
from grpc_rdma_amd import shard, fanout, stream as gs
from oracle import pyorc
grp = shard.RankGroup(backend="gloo")
g.init(0)
N_CONNS, R, MAX_SGE = 4, 1 << 18, 30
mine = shard.connections_for_rank(N_CONNS, grp.rank, grp.world)

def framed(conn, n_msgs, msg_len):
    rng = random.Random(1000 + conn)
    out = []
    for i in range(n_msgs):
        msg = bytes(rng.getrandbits(8) for _ in range(64)) * (msg_len // 64) + bytes(msg_len %% 64)
        wire, lens = pyorc.h2_frame_message(msg, stream_id=2 * i + 1)
        off = 0
        for n in lens:
            out.append(wire[off:off + n]); off += n
    return out

# ---- configs[3] shape: my share of the connections, one job, one op per connection in every launch
links, keep = [], []
for c in mine:
    sl = framed(c, 2, 20000)
    tx, rx = g.Pair(R, MAX_SGE), g.Pair(R, MAX_SGE)
    g.connect_pairs(tx, rx)
    bufs = [g.DeviceBuffer(data=s, offset=(7 * k) %% 16) for k, s in enumerate(sl)]
    nbytes = sum(len(s) for s in sl)
    cap = nbytes + 32 * (2 * len(sl) + 64) + 4096
    dst = g.DeviceBuffer(nbytes=cap)
    links.append((tx, rx, [(b.ptr, len(s)) for b, s in zip(bufs, sl)], dst.ptr, cap, 2 * len(sl) + 64))
    keep.append((sl, bufs, dst, cap, nbytes, tx, rx))
job = gs.MultiStreamJob(links, 12)
grp.barrier()
t0 = time.perf_counter()
r = job.run(gs.RUN_EAGER)
mine_s = time.perf_counter() - t0
grp.barrier()
assert r.done
ok = True
for k, (sl, bufs, dst, cap, nbytes, tx, rx) in enumerate(keep):
    mem = dst.read(cap)
    got = b"".join(mem[o:o + n] for o, n in job.delivered_slices(k))
    ok = ok and got == b"".join(sl) and rx.ring_mem() == bytes(R)
job_s = grp.max(mine_s)
total = grp.sum(sum(x[4] for x in keep))
all_ok = grp.sum(1 if ok else 0)

# ---- configs[4]: rank 0 ingests ONE stream, its delivered arena is fanned out
stream_sl = framed(99, 4, 30000)
arena, slices = torch.zeros(1, dtype=torch.uint8), []
if grp.rank == 0:
    tx, rx = g.Pair(1 << 20, 4095), g.Pair(1 << 20, 4095)
    g.connect_pairs(tx, rx)
    bufs = [g.DeviceBuffer(data=s) for s in stream_sl]
    nbytes = sum(len(s) for s in stream_sl)
    cap = nbytes + 32 * (2 * len(stream_sl) + 64) + 4096
    dst = g.DeviceBuffer(nbytes=cap)
    j2 = gs.MultiStreamJob([(tx, rx, [(b.ptr, len(s)) for b, s in zip(bufs, stream_sl)], dst.ptr, cap, 2 * len(stream_sl) + 64)], 12)
    r2 = j2.run(gs.RUN_EAGER)
    assert r2.done and r2.bytes_delivered == nbytes
    slices = j2.delivered_slices(0)
    arena = torch.frombuffer(bytearray(dst.read(cap)), dtype=torch.uint8)
part, my_slices = fanout.scatter_arena(grp, arena, slices, src=0)
payload = b"".join(bytes(part[o:o + n].tolist()) for o, n in my_slices)
print(json.dumps({"rank": grp.rank, "conns": mine, "ok": bool(ok), "all_ok": all_ok, "job_s": job_s, "total": total,
                  "fan_bytes": len(payload), "fan_sha": hashlib.sha256(payload).hexdigest(),
                  "fan_slices": len(my_slices)}))
grp.close()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def emu_lib():
    if not os.path.exists(EMU_SO):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cc", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return EMU_SO


def test_two_ranks_run_their_share_of_the_connections_and_the_fanout(tmp_path, emu_lib):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GRDMA_LIB_PATH=emu_lib, GRDMA_TEST_ALLOW_EMU="1",
                   GRDMA_COPY_BLOCKS="2")  # (grid size of the copy kernels: the emulator runs a workgroup at a time)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    docs = sorted((json.loads(o[0].strip().splitlines()[-1]) for o in outs), key=lambda d: d["rank"])
    # sharding: contiguous blocks, every connection on exactly one rank, every rank's streams delivered intact
    assert docs[0]["conns"] == [0, 1] and docs[1]["conns"] == [2, 3]
    assert docs[0]["ok"] and docs[1]["ok"] and docs[0]["all_ok"] == 2
    # the contract's aggregates: MAX of the ranks' times, SUM of the ranks' bytes (identical on both ranks)
    assert docs[0]["job_s"] == docs[1]["job_s"] > 0
    per_conn = 2 * (20000 + 5) + 2 * 9 * 2  # 2 messages: payload + 5-byte message header, 2 frames of 9-byte headers
    assert docs[0]["total"] == docs[1]["total"] == 4 * per_conn
    # fan-out: the ranks' shares, in rank order, are the framed stream rank 0 ingested
    import random
    rng = random.Random(1000 + 99)
    want = b""
    for i in range(4):
        msg = bytes(rng.getrandbits(8) for _ in range(64)) * (30000 // 64) + bytes(30000 % 64)
        wire, _lens = pyorc.h2_frame_message(msg, stream_id=2 * i + 1)
        want += wire
    assert docs[0]["fan_bytes"] + docs[1]["fan_bytes"] == len(want)
    a = docs[0]["fan_bytes"]
    assert docs[0]["fan_sha"] == hashlib.sha256(want[:a]).hexdigest()
    assert docs[1]["fan_sha"] == hashlib.sha256(want[a:]).hexdigest()
    assert docs[0]["fan_slices"] > 0 and docs[1]["fan_slices"] > 0
    assert abs(docs[0]["fan_bytes"] - docs[1]["fan_bytes"]) < 40000  # (balanced to within a couple of slices)
