"""The file a maintainer SHIPS, executed: integration/rdma_hip_posix.cc (the drop-in for the reference's
src/core/lib/iomgr/rdma_bp_posix.cc: grpc_rdma_bp_create, the grpc_endpoint vtable, the iomgr traits of
include/grdma_endpoint_impl.hpp -- grpc_slice_new_with_user_data windows, grpc_core::Closure::Run,
grpc_fd_notify_on_read / _write, RefCount, rdma_annotate_error) + integration/ibverbs_facade, linked by oracle/Makefile
against the product library under the SAME driver and iomgr stand-ins that run the reference's own endpoint
(oracle/ref_endpoint_trace.cc, -DGRDMA_HIP_ADAPTER):

  oracle/_ref/ref_endpoint_trace       the reference's rdma_bp_posix.cc + pair.cc, compiled unmodified (software verbs)
  oracle/_ref/hip_endpoint_trace       the shipped endpoint over grpc-rdma_amd/libgrdma_amd.so      (-m gpu)
  oracle/_ref/hip_endpoint_trace_emu   the shipped endpoint over oracle/_build/libgrdma_emu.so      (CPU suite: the
                                       product sources on the wave emulator)

Both replay the same seeded operation lists -- grpc_endpoint_write of slice buffers larger than the ring and longer than
max_sge, the writable edge, endpoint reads that find data, find nothing, find the peer gone, PairPollable::Send
underneath, grpc_endpoint_shutdown + destroy -- and print, per operation, what chttp2 would see: whether the write
callback ran, the read callback's bytes (crc32) and slice, would-block, the peer's readable size, the sender's writable
size, HasPendingWrites, and for a failed operation the error text with its fd / status / target-address annotation.
The two outputs must be IDENTICAL, line by line.

The shipped endpoint runs in its reference-exact configuration: GRPC_RDMA_HIP_READ_AHEAD=1 (one endpoint read per
device pass -- the default reads ahead, up to 1024 reads per pass, which delivers the same bytes in reads sized at
another moment) and GRPC_RDMA_HIP_SEND_BUFFER_KB=0 (a write completes when its last Send has, as rdma_flush's does --
the default completes a write that fits a send buffer once it has been copied).  Its Sends and drains are asynchronous;
the driver fires the edges the event engine would fire (the facade's HasMessage / HasPendingWrites) until the operation
is where the reference's synchronous call returns."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_endpoint_trace")
HIP = os.path.join(ROOT, "oracle", "_ref", "hip_endpoint_trace")
HIP_EMU = os.path.join(ROOT, "oracle", "_ref", "hip_endpoint_trace_emu")


def _ops(seed, ring_kb, max_sge, n_ops, zc=0.0):
    """Writes (some larger than the ring, some of more slices than max_sge), writable edges, endpoint reads on both
    sides, raw Sends.  The list's author does not model the protocol: a write while one is outstanding, a raw Send under
    a waiting write are answered "busy" by the driver (both builds alike).  zc: the share of operations that run the
    zero-copy hook (Z: pool lookup, AllocateSendBuffer, the host's serialisation, SendZerocopy from the cursor)."""
    rng = random.Random(777000 + 1000 * seed + ring_kb + max_sge + (17 if zc else 0))
    ring = ring_kb * 1024
    text = []
    for _ in range(n_ops):
        side = rng.randrange(2)
        r = rng.random()
        if zc:
            # the hook's own mix: messages below / around / above what the ring can take at once, some larger than the
            # buffer (no buffer then); reads that make room; now and then a small write and its edge in between
            if r < zc:
                n = rng.choice([1, 9, 200, 3000, ring // 8, ring // 3, ring // 2, ring - 4096, ring + 1])
                text.append("Z %d %d %d %d" % (side, rng.randrange(1 << 16), rng.choice([0, 5, 14, 14, 300]), n))
            elif r < zc + 0.06:
                lens = [rng.choice([9, 14, 100, 255, 256, 257, 4000]) for _ in range(rng.choice([1, 2, 5, max_sge + 3]))]
                text.append("W %d %d %d %s" % (side, rng.randrange(1 << 16), len(lens), " ".join(map(str, lens))))
            elif r < zc + 0.16:
                text.append("F %d" % side)
            else:
                text.append("E %d" % side)
            continue
        if r < 0.25:
            n = rng.choice([1, 2, 5, max_sge, max_sge + 3, 2 * max_sge + 1])
            lens = [rng.choice([ring // 6, ring // 2, ring]) if rng.random() < 0.12 else
                    rng.choice([9, 14, 100, 255, 256, 257, 4000]) for _ in range(n)]
            text.append("W %d %d %d %s" % (side, rng.randrange(1 << 16), len(lens), " ".join(map(str, lens))))
        elif r < 0.45:
            text.append("F %d" % side)
        elif r < 0.5:
            n = rng.choice([1, 2, 3, max_sge])
            lens = [rng.choice([9, 5, 14, 100, 255, 256, 257, 3000]) for _ in range(n)]
            text.append("S %d 0 %d %d %s" % (side, rng.randrange(1 << 16), len(lens), " ".join(map(str, lens))))
        else:
            text.append("E %d" % side)
    return text


def _run(binary, text, ring_kb, max_sge, extra=None):
    env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge),
               GRPC_RDMA_MAX_SGE=str(max_sge), GRPC_RDMA_HIP_SEND_BUFFER_KB="0", GRPC_RDMA_HIP_READ_AHEAD="1",
               GRPC_RDMA_HIP_PAIR_POOL_MB="0", TRACE_ENGINE_SETTLE="1")
    env.update(extra or {})
    p = subprocess.run([binary], input="\n".join(text) + "\n", capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, "%s: rc %d, %s" % (os.path.basename(binary), p.returncode, p.stderr[-600:])
    return p.stdout.strip().splitlines()


def _compare(binary, seed, ring_kb, max_sge, n_ops, extra=None, zc=0.0):
    text = _ops(seed, ring_kb, max_sge, n_ops, zc)
    # then both directions drained: whatever write still waits is flushed and read out
    tail = []
    for _ in range(3):
        for side in (0, 1):
            tail += ["F %d" % side] + ["E %d" % (1 - side)] * 6
    want = _run(REF, text + tail, ring_kb, max_sge, {"GRPC_RDMA_ZEROCOPY_THRESHOLD_KB": "0"} if zc else None)
    got = _run(binary, text + tail, ring_kb, max_sge, dict(extra or {}, GRPC_RDMA_ZEROCOPY_THRESHOLD_KB="0") if zc else extra)
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "operation %d (%s): shipped endpoint %r, reference endpoint %r" % (k, (text + tail)[k][:48], g, w)
    return want


CONFIGS = [(64, 30), (256, 30), (64, 4), (1024, 64)]


@pytest.mark.parametrize("ring_kb,max_sge", CONFIGS)
@pytest.mark.parametrize("seed", range(5))
def test_shipped_endpoint_equals_the_reference_endpoint_under_the_emulator(seed, ring_kb, max_sge):
    if not (os.path.exists(REF) and os.path.exists(HIP_EMU)):
        pytest.skip("oracle/_ref/ref_endpoint_trace / hip_endpoint_trace_emu not built (no reference tree here)")
    lines = _compare(HIP_EMU, seed, ring_kb, max_sge, 140)
    # (the list did something: data delivered, a write that had to wait for the edge, a read that found nothing)
    assert sum(1 for ln in lines if ln.startswith("E ") and not ln.startswith("E -")) >= 10
    assert any(ln.startswith("W 0 ") for ln in lines) and any(ln.startswith("E -1") for ln in lines)


@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (256, 5)])
@pytest.mark.parametrize("seed", range(3))
def test_zero_copy_hook_of_the_shipped_facade_equals_the_reference_pair_under_the_emulator(seed, ring_kb, max_sge):
    """SURVEY.md 8(f-3), executed on BOTH builds by the same driver code: the body of the reference's
    CoreCodegen::grpc_call_allocate_send_buffer (core_codegen.cc:126-142 -- Config's threshold, PairPool::Get().Get(id),
    get_status() == kConnected, AllocateSendBuffer), the host writing the message through the returned pointer, and
    SendZerocopy from the cursor -- against the reference's own PairPool / PairPollable (pair.cc:305-323, 793-941) and
    against the shipped facade over the library (pinned host buffer, the gather reads it in place), mixed with endpoint
    writes, edges, raw Sends and reads.  Whether the pool knew the id, whether a buffer was handed out, the bytes of
    every SendZerocopy, the sizes behind it and everything the peer then reads: identical, line by line."""
    if not (os.path.exists(REF) and os.path.exists(HIP_EMU)):
        pytest.skip("oracle/_ref/ref_endpoint_trace / hip_endpoint_trace_emu not built (no reference tree here)")
    lines = _compare(HIP_EMU, seed, ring_kb, max_sge, 160, zc=0.3)
    z = [ln for ln in lines if ln.startswith("Z ") and ln != "Z busy"]
    assert sum(1 for ln in z if ln.startswith("Z 1 1 ")) >= 5 and any(ln.startswith("Z 1 0") for ln in z), z
    # (a message that met a ring without room for all of it: SendZerocopy again from the cursor, until one takes nothing)
    assert any(len(ln.split("|")[0].split()) > 4 for ln in z), "no message went out in more than one SendZerocopy: %s" % z


@pytest.mark.parametrize("binary_kind", ["emu"])
def test_errors_of_a_closed_connection_carry_the_reference_text_and_annotation(binary_kind):
    """grpc_endpoint_shutdown + destroy of one side (rdma_free: Disconnect): the other side's read reports "Pair
    closed" and its write "Peer has been exited", both annotated with the endpoint's fd, UNAVAILABLE (14) and the peer
    string (rdma_annotate_error, rdma_bp_posix.cc:86-96, 220-238, 499-518) -- identical text from both endpoints."""
    if not (os.path.exists(REF) and os.path.exists(HIP_EMU)):
        pytest.skip("trace binaries not built")
    text = ["W 0 11 2 300 5000", "E 1", "E 1", "E 1", "E 1", "C 0", "E 1", "W 1 12 3 70000 70000 70000", "E 1"]
    want = _run(REF, text, 64, 30)
    got = _run(HIP_EMU, text, 64, 30)
    assert got == want
    failed = [ln for ln in want if " -2 " in ln]
    assert any("Pair closed" in ln and "status 14" in ln and "fd 1" in ln and "target peer-of-1" in ln for ln in failed), want
    assert any("Peer has been exited" in ln for ln in failed), want


@pytest.mark.gpu
@pytest.mark.parametrize("wire", ["direct", "staged"])
@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (1024, 64), (4096, 30)])
def test_shipped_endpoint_equals_the_reference_endpoint_on_the_gpu(gpu, ring_kb, max_sge, wire):
    """The same comparison with the shipped endpoint over the real library, rings in HBM: every operation of the
    reference endpoint's trace reproduced by integration/rdma_hip_posix.cc on the MI355X."""
    if not (os.path.exists(REF) and os.path.exists(HIP)):
        pytest.skip("oracle/_ref/ref_endpoint_trace / hip_endpoint_trace not built (they are built where the reference tree is)")
    for seed in range(3):
        _compare(HIP, seed, ring_kb, max_sge, 120, {"GRPC_RDMA_HIP_WIRE": wire})


@pytest.mark.gpu
@pytest.mark.parametrize("wire", ["direct", "staged"])
@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (4096, 30)])
def test_zero_copy_hook_of_the_shipped_facade_equals_the_reference_pair_on_the_gpu(gpu, ring_kb, max_sge, wire):
    """The zero-copy hook on the MI355X: the host serialises into the pinned buffer the facade's AllocateSendBuffer
    returned, the gather kernel reads it in place into the peer's HBM ring; line by line what the reference's own
    PairPool / AllocateSendBuffer / SendZerocopy print under the same driver."""
    if not (os.path.exists(REF) and os.path.exists(HIP)):
        pytest.skip("oracle/_ref/ref_endpoint_trace / hip_endpoint_trace not built (they are built where the reference tree is)")
    for seed in range(3):
        lines = _compare(HIP, seed, ring_kb, max_sge, 160, {"GRPC_RDMA_HIP_WIRE": wire}, zc=0.3)
        assert sum(1 for ln in lines if ln.startswith("Z 1 1 ")) >= 3


@pytest.mark.gpu
def test_errors_of_a_closed_connection_on_the_gpu(gpu):
    if not (os.path.exists(REF) and os.path.exists(HIP)):
        pytest.skip("trace binaries not built")
    text = ["W 0 11 2 300 5000", "E 1", "E 1", "E 1", "E 1", "C 0", "E 1", "W 1 12 3 70000 70000 70000", "E 1"]
    assert _run(HIP, text, 64, 30) == _run(REF, text, 64, 30)
