"""CPU: audit of the shipped gfx950 code object.  The co-residency hazard of round 1 came with the SLP
vectorizer's packed-fp32 code: v_pk_* instructions with op_sel / neg operand modifiers and v_pk_mov_b32 in the FFT kernels.
The library is built without that vectorizer; this test keeps it that way by disassembling the device code inside
libbsrnn_hip.so: packed fp32 may only appear in the plain forms the hand-written vector code produces."""
import os
import re
import struct
import subprocess

import pytest

from conftest import REPO

LIB = os.path.join(REPO, "speechseparation_amd", "lib", "libbsrnn_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(path):
    """(triple, bytes) of every entry of the clang offload bundle embedded in the shared library."""
    blob = open(path, "rb").read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            out.append((triple, blob[pos + off:pos + off + size]))
        pos = blob.find(MAGIC, pos + 1)
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_modified_packed_fp32_in_the_shipped_kernels(tmp_path):
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    objs = [(t, b) for t, b in device_code_objects(LIB) if "gfx950" in t and len(b) > 0]
    assert objs, "no gfx950 code object found in %s" % LIB
    total = 0
    for i, (triple, data) in enumerate(objs):
        f = tmp_path / ("dev%d.co" % i)
        f.write_bytes(data)
        asm = subprocess.run([OBJDUMP, "-d", str(f)], capture_output=True, text=True, check=True).stdout
        pk = [ln for ln in asm.splitlines() if re.search(r"\bv_pk_(add|mul|fma|mov)_(f32|b32)\b", ln)]
        total += len(pk)
        bad = [ln.strip() for ln in pk if "v_pk_mov_b32" in ln or re.search(r"op_sel:|neg_lo:|neg_hi:", ln)]
        assert not bad, "packed fp32 with operand modifiers in %s: %s" % (triple, bad[:5])
    assert total > 0      # the disassembly worked: the GEMM split and the LSTM cells do use plain v_pk_mul / add / fma
