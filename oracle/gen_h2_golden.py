#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: extracts the HTTP/2 byte vectors the reference's own
tests feed to chttp2 (test/core/bad_client/tests/simple_request.cc:28-44,135-139 and
head_of_line_blocking.cc:30-67,117-133) and writes them, as hex, to
tests/golden/h2_bad_client.json together with the frame list each vector must
parse to (written out by hand from the comments in those tests).  Run in the
build container: python oracle/gen_h2_golden.py"""
import json
import os
import re

REF = "/root/reference/test/core/bad_client/tests"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def c_unescape(lit):
    out = bytearray()
    i = 0
    while i < len(lit):
        ch = lit[i]
        if ch != "\\":
            out.append(ord(ch))
            i += 1
            continue
        i += 1
        e = lit[i]
        if e == "x":
            j = i + 1
            while j < len(lit) and lit[j] in "0123456789abcdefABCDEF":
                j += 1
            out.append(int(lit[i + 1:j], 16) & 0xFF)
            i = j
        elif e in "01234567":
            j = i
            while j < len(lit) and j < i + 3 and lit[j] in "01234567":
                j += 1
            out.append(int(lit[i:j], 8))
            i = j
        else:
            out.append({"n": 10, "r": 13, "t": 9, "\\": 92, '"': 34, "'": 39, "0": 0}[e])
            i += 1
    return bytes(out)


def literals(block):
    block = re.sub(r"/\*.*?\*/", "", block, flags=re.S)
    block = re.sub(r"//[^\n]*", "", block)
    return b"".join(c_unescape(m) for m in re.findall(r'"((?:[^"\\]|\\.)*)"', block))


def define_block(src, name):
    m = re.search(r"#define " + name + r"\b(.*?)(?<!\\)\n", src, flags=re.S)
    return m.group(1)


def main():
    simple = open(os.path.join(REF, "simple_request.cc")).read()
    pfx = literals(define_block(simple, "PFX_STR"))
    hol_src = open(os.path.join(REF, "head_of_line_blocking.cc")).read()
    m = re.search(r"static const char prefix\[\] =(.*?);\n", hol_src, flags=re.S)
    hol_prefix = literals(m.group(1))
    frame = bytes([0, 0x03, 0xE8, 0, 0, 0, 0, 0, 3]) + b"a" * 1000  # NUM_FRAMES x FRAME_SIZE, :117-131
    hol = hol_prefix + frame * 10
    vectors = [
        {"name": "simple_request_illegal_grpc_frame",
         "cite": "test/core/bad_client/tests/simple_request.cc:135-139",
         "hex": (pfx + bytes.fromhex("000005000000000001") + bytes.fromhex("3400000000")).hex(),
         "frames": [[4, 0, 0, 0], [1, 4, 1, 0xC9], [0, 0, 1, 5]],
         "stream_error": {"stream": 1, "code": "bad_grpc_frame_type"}, "messages": []},
        {"name": "simple_request_bad_data_flags",
         "cite": "test/core/bad_client/tests/simple_request.cc:141-143",
         "hex": (pfx + bytes.fromhex("000000000200000001")).hex(),
         "frames": [[4, 0, 0, 0], [1, 4, 1, 0xC9], [0, 2, 1, 0]],
         "stream_error": {"stream": 1, "code": "data_flags"}, "messages": []},
        {"name": "head_of_line_blocking",
         "cite": "test/core/bad_client/tests/head_of_line_blocking.cc:30-67,117-133",
         "hex": hol.hex(),
         "frames": [[4, 0, 0, 0], [1, 4, 1, 0xD0], [0, 0, 1, 5], [1, 4, 3, 0xD0], [0, 0, 3, 5]]
                   + [[0, 0, 3, 1000]] * 10,
         "stream_error": None,
         # stream 1 announces a compressed 10000-byte message it never sends;
         # stream 3 announces one and delivers it in 10 x 1000-byte DATA frames
         "messages": [{"stream": 1, "compressed": 1, "length": 10000, "complete": False},
                      {"stream": 3, "compressed": 1, "length": 10000, "complete": True,
                       "payload_byte": "61"}]},
    ]
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "h2_bad_client.json"), "w") as f:
        json.dump({"generator": "oracle/gen_h2_golden.py", "vectors": vectors}, f, indent=1)
    print("prefix lens", len(pfx), len(hol_prefix), "->", os.path.join(OUT, "h2_bad_client.json"))


if __name__ == "__main__":
    main()
