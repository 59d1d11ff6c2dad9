"""GPU: a client PROCESS and a server PROCESS connected the way grpc_rdma_bp_create connects
them (48-byte Address + memory-region exchange over a socket, rdma_bp_posix.cc:640-692,763-784;
pair.cc:143-168), then the endpoint conformance grid of the reference
(test/core/iomgr/endpoint_tests.cc:341-355) echoed byte-identically across the process
boundary.  The rings live in HBM; each end's one-sided writes land in the other process's ring
through the HIP IPC mapping of that allocation."""
import os
import socket
import subprocess
import sys

import pytest

from oracle import pyorc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEER = os.path.join(ROOT, "tests", "two_proc_peer.py")


def run_pair(ring_kib, num_bytes, write_size, slice_size, devs=(0, 0), pair_flags=4, extra_env=None):
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    procs = []
    env = dict(os.environ, GRDMA_TEST_PAIR_FLAGS=str(pair_flags))
    env.update(extra_env or {})
    for role, sock, dev in (("server", a, devs[0]), ("client", b, devs[1])):
        os.set_inheritable(sock.fileno(), True)
        procs.append(subprocess.Popen(
            [sys.executable, PEER, role, str(sock.fileno()), str(dev), str(ring_kib), str(num_bytes),
             str(write_size), str(slice_size)],
            pass_fds=[sock.fileno()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    a.close()
    b.close()
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, "peer failed:\n" + o + e[-3000:]
    return outs


@pytest.mark.parametrize("ring_kib,num_bytes,write_size,slice_size", [
    (4096, 2000000, 100000, 8192),    # endpoint_tests.cc:345 scaled 5x down, reference-default 4 MiB ring
    (4096, 20160, 10000, 1),          # :346 scaled: 1-byte slices, one ring record each
    (64, 500000, 100000, 8192),       # small ring: every write parks on credit several times
    (4096, 40320, 777, 777),          # one point of the write = slice sweep (:350-352)
])
def test_echo_between_two_processes(gpu, ring_kib, num_bytes, write_size, slice_size):
    outs = run_pair(ring_kib, num_bytes, write_size, slice_size)
    assert "ok server" in outs[0][1] and "ok client" in outs[1][1]


def test_echo_between_two_processes_fine_grained_rings(gpu):
    """GRDMA_RING_FINE_GRAINED: ring and status block are fine-grained device memory -- what a
    remote writer (peer process, peer GPU, NIC) needs to see acknowledged stores without cache
    maintenance.  Every pair that is exported to another process must be created this way (the
    export refuses a coarse-grained ring): all cases of this file run with it."""
    outs = run_pair(256, 600000, 100000, 8192, pair_flags=4)
    assert "ok server" in outs[0][1] and "ok client" in outs[1][1]


@pytest.mark.parametrize("ring_kib,num_bytes,write_size,slice_size", [(4096, 2000000, 100000, 8192), (64, 500000, 100000, 8192)])
def test_echo_between_two_processes_direct_wire_into_the_mapped_peer_ring(gpu, ring_kib, num_bytes, write_size, slice_size):
    """GRDMA_WIRE_DIRECT across a process boundary: the gather kernel of one process encodes its records STRAIGHT into
    the other process's ring through the IPC mapping (no staging buffer, no wire step), the credit report comes back into
    this process's connection block the same way.  This is the code path of a pair whose ends sit on two GPUs of a node
    (the peer ring mapped over xGMI instead of over the same GPU's memory: test_echo_between_two_gpus, which needs a second
    GPU and has never had one) -- the stand-in one GPU allows.  The small ring parks every write on credit several times."""
    outs = run_pair(ring_kib, num_bytes, write_size, slice_size, pair_flags=4 | 2)
    assert "ok server" in outs[0][1] and "ok client" in outs[1][1]


def test_credit_return_across_processes_for_ten_seconds(gpu):
    """64 KiB rings: every ~32 KiB consumed the reader zero-fills what it read and posts a credit report into
    the OTHER process's connection block, and the writer reuses that space at once.  Ten seconds of echo
    (tens of thousands of credit cycles each way) -- a zero-fill that became visible after the credit, or a
    record accepted before its payload had landed, shows up as a byte mismatch or as a dirty ring at the end."""
    outs = run_pair(64, 0, 30000, 4096, pair_flags=4, extra_env={"GRDMA_TEST_SECONDS": "10"})
    assert "ok server" in outs[0][1] and "ok client" in outs[1][1]


def test_unary_pingpong_between_two_processes_on_the_arrival_triggered_path(gpu):
    """BASELINE configs[1] across a process boundary: client process and server process, each with its own latency
    engine; every read is a standing order carried out by a watcher workgroup of the READING process's engine when
    the bytes the other process wrote land in its ring (k_watch) -- the path a NIC-fed ring takes too: the reader
    learns of a message from its own ring, as ring_buffer.cc:56-97 / ev_epollex_rdma_bpev_linux.cc:1105-1149 do.
    5000 round trips of [14 B][66 B]; byte sums, hit counts and zero rings checked in both processes (tests/two_proc_peer.py: unary_pingpong)."""
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    env = dict(os.environ, GRDMA_TEST_PAIR_FLAGS="4")
    procs = []
    for role, sock in (("pp_server", a), ("pp_client", b)):
        os.set_inheritable(sock.fileno(), True)
        procs.append(subprocess.Popen([sys.executable, PEER, role, str(sock.fileno()), "0", "1024", "5000", "0", "0"],
                                      pass_fds=[sock.fileno()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    a.close()
    b.close()
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, "peer failed:\n" + o + e[-3000:]
    assert "ok server 5000 round trips" in outs[0][1] and "ok client 5000 round trips" in outs[1][1]
    print(outs[1][1].strip())


def test_killed_peer_turns_the_pair_half_closed(gpu):
    """kill -9 of the peer process: no Disconnect(), no peer_exit word.  The reference notices through
    ibv_query_qp every 500 ms (pair.cc:358-372 => kHalfClosed) and the TCP fd's hang-up; here get_status()
    checks the peer's pid and the bootstrap socket.  The survivor must report kHalfClosed within a second."""
    import signal
    import time
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    env = dict(os.environ, GRDMA_TEST_PAIR_FLAGS="4")
    procs = {}
    for role, sock in (("watcher", a), ("victim", b)):
        os.set_inheritable(sock.fileno(), True)
        procs[role] = subprocess.Popen([sys.executable, PEER, role, str(sock.fileno()), "0", "256", "0", "0", "0"],
                                       pass_fds=[sock.fileno()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    a.close()
    b.close()
    try:
        line = procs["victim"].stdout.readline()   # "connected": both ends are up
        assert "connected" in line, line + procs["victim"].stderr.read()[-2000:]
        line = procs["watcher"].stdout.readline()
        assert "connected" in line, line
        time.sleep(0.3)
        procs["victim"].send_signal(signal.SIGKILL)
        out, err = procs["watcher"].communicate(timeout=60)
        assert procs["watcher"].returncode == 0 and "ok watcher" in out, out + err[-3000:]
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()


def test_echo_between_two_gpus(gpu):
    lib = gpu.load()
    if lib.grdma_device_count() < 2:
        pytest.skip("one GPU visible: the cross-GPU (xGMI) case needs two")
    outs = run_pair(4096, 2000000, 100000, 8192, devs=(0, 1))
    assert "ok server" in outs[0][1] and "ok client" in outs[1][1]


def test_connect_checks_of_the_reference(gpu):
    """Connect() asserts equal tag and equal ring size (pair.cc:146-149); the offsets that come off the socket must be
    this build's; a peer in this process must exist."""
    g = gpu
    coarse = g.Pair(1 << 20, 30)
    with pytest.raises(g.GrdmaError, match="FINE_GRAINED"):   # what another process writes must be fine-grained memory
        coarse.export_address()
    a, b = g.Pair(1 << 20, 30, flags=4), g.Pair(2 << 20, 30, flags=4)
    blob = bytearray(b.export_address())
    assert len(blob) == 208 and blob[32] == 0xA0
    assert int.from_bytes(blob[40:48], "little") == 2 << 20
    with pytest.raises(g.GrdmaError, match="ring sizes differ"):
        a.connect_remote(bytes(blob))
    c = g.Pair(1 << 20, 30, flags=4)
    blob = bytearray(c.export_address())
    blob[32] = 0xA1
    with pytest.raises(g.GrdmaError, match="tag"):
        a.connect_remote(bytes(blob))
    blob[32] = 0xA0
    blob[64 + 16:64 + 24] = (12345).to_bytes(8, "little")   # status_off comes off a socket: it must be this build's
    blob[64 + 8:64 + 16] = (os.getpid() + 1).to_bytes(8, "little")
    with pytest.raises(g.GrdmaError, match="layout"):
        a.connect_remote(bytes(blob))
    # an address that names a pair of THIS process which does not exist (an IPC handle cannot be opened where it was
    # made: a peer in this process is looked up by its serial, addr.qpn)
    blob = bytearray(c.export_address())
    blob[4:8] = (0x7FFFFFF0).to_bytes(4, "little")
    with pytest.raises(g.GrdmaError, match="does not exist"):
        a.connect_remote(bytes(blob))
    assert a.get_status() == 1  # still kInitialized


def test_two_pairs_of_one_process_connect_through_the_bootstrap_path(gpu):
    """Client and server in ONE process (what gRPC's own end2end tests are): both ends export their address and connect
    to the other one's -- grdma_pair_bootstrap_fd's two halves -- and the peer, found to live in this process, is looked
    up by its serial instead of being mapped through an IPC handle.  Then an echo in both directions."""
    g = gpu
    a, b = g.Pair(1 << 18, 30, flags=4), g.Pair(1 << 18, 30, flags=4)
    ba, bb = a.export_address(), b.export_address()
    a.connect_remote(bb)
    b.connect_remote(ba)
    assert a.get_status() == 2 and b.get_status() == 2  # kConnected
    o = pyorc.OracleLink(1 << 18, 30)
    for k, (src, dst, so, do) in enumerate([(a, b, 0, 1), (b, a, 1, 0)] * 2):
        msg = [bytes((7 * i + k) % 251 for i in range(n)) for n in (9, 5000, 14, 70000)]
        bufs = [g.DeviceBuffer(data=m) for m in msg]
        assert src.Send(bufs) == o.send(so, msg)
        got, _ = dst.endpoint_read(max_reads=64)
        exp = []
        while True:
            s_, _a = o.endpoint_read(do)
            if not s_:
                break
            exp.append(s_)
        assert got == exp
    o.close()
