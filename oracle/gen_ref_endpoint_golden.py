"""TEST INFRASTRUCTURE ONLY.  Golden traces produced by the REFERENCE ITSELF: oracle/_ref/ref_endpoint_trace (the reference's
unmodified rdma_bp_posix.cc + pair.cc + ring_buffer.cc over the software verbs of oracle/fakeverbs, oracle/Makefile) replays
seeded operation lists -- PairPollable::Send on one side, endpoint reads on the other -- and what it returned is written to
tests/golden/ref_endpoint_*.json.  The reference tree and its build exist in the build container only; the vectors travel:
tests/test_golden_ref_endpoint.py (CPU: the oracle reproduces them) and tests/test_zz_gpu_golden_ref_endpoint.py (GPU: the HIP
pair reproduces them) need nothing but the JSON.

    python -m oracle.gen_ref_endpoint_golden          (needs /root/reference; rewrites the four files)

A slice of length n with seed s and index i holds bytes ((s * 131 + i * 17 + j * 7 + (j >> 8)) & 0xFF for j < n)."""
import json
import os
import random
import subprocess
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyorc  # noqa: E402

CASES = [  # name, ring KiB, max_sge, ops, seed
    ("r64k_sge30", 64, 30, 140, 11),
    ("r256k_sge30", 256, 30, 140, 12),
    ("r64k_sge4", 64, 4, 140, 13),
    ("r1m_sge64", 1024, 64, 100, 14),
]


def make_ops(rng, ring, max_sge, n_ops):
    ops = []
    for _ in range(n_ops):
        if rng.random() < 0.42:
            n = rng.choice([1, 2, 3, max_sge, max_sge + 2, rng.randrange(1, 2 * max_sge)])
            if rng.random() < 0.5:
                lens = [rng.choice([9, 5, 14, 100, 255, 256, 257]) if k % 2 == 0 else rng.randrange(1, ring // 5)
                        for k in range(n)]
            else:
                lens = [rng.choice([1, 9, 200, 256, 300, 511, 512, 5000]) for _ in range(n)]
            bi = rng.randrange(lens[0]) if rng.random() < 0.25 else 0
            ops.append(["S", bi, rng.randrange(1 << 16), lens])
        else:
            ops.append(["E"])
    ops += [["E"]] * 6
    return ops


def main():
    pyorc.build()
    if not os.path.exists(pyorc.REF_ENDPOINT_TRACE):
        raise SystemExit("oracle/_ref/ref_endpoint_trace is not built (no reference tree here)")
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, ring_kb, max_sge, n_ops, seed in CASES:
        ops = make_ops(random.Random(seed), ring_kb * 1024, max_sge, n_ops)
        text = []
        for op in ops:
            if op[0] == "S":
                text.append("S 0 %d %d %d %s" % (op[1], op[2], len(op[3]), " ".join(map(str, op[3]))))
            else:
                text.append("E 1")
        env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge))
        p = subprocess.run([pyorc.REF_ENDPOINT_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True,
                           timeout=120, env=env, check=True)
        results = []
        for op, line in zip(ops, p.stdout.strip().splitlines()):
            f = line.split()
            if op[0] == "S":
                results.append([int(f[1])])                                   # bytes the Send accepted
            else:
                results.append([int(f[1]), int(f[2]), int(f[4]), int(f[5])])  # delivered (-1 = would block), crc32,
                                                                              # readable left, writable size of the sender
        assert len(results) == len(ops)
        doc = {"generated_by": "oracle/gen_ref_endpoint_golden.py over oracle/_ref/ref_endpoint_trace "
                               "(the reference's rdma_bp_posix.cc + pair.cc, unmodified)",
               "ring_kib": ring_kb, "max_sge": max_sge, "ops": ops, "results": results,
               "crc_of_results": zlib.crc32(json.dumps(results).encode()) & 0xFFFFFFFF}
        with open(os.path.join(out_dir, "ref_endpoint_%s.json" % name), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        print(name, len(ops), "ops")


if __name__ == "__main__":
    main()
