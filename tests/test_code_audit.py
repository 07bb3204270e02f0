"""CPU: audit of the shipped gfx950 code object.  The co-residency hazard (NOTEBOOK.md R3.4;
profiles/r03_vpk_hazard.txt): STFT / iSTFT kernels whose butterflies the SLP vectorizer had turned into packed-fp32 code
(v_pk_add / mul / fma_f32, v_pk_mov_b32) returned whole frames of garbage WHILE WAVES OF ANOTHER KERNEL SHARED THEIR CU (beside
GEMM / LSTM launches: 4-20 of 20 runs wrong), never alone, and never when their workgroups were made to own the CU (all of its LDS
claimed: 0 of 80 runs wrong with the same vectorised code).  Forced-zero wait counts do not change it, LDS guard bands do not,
scalar fp32 code does not show it.  The victims were kernels that issue no MFMA of their own; the kernels that do (GEMM, chains,
LSTMs) run hand-written packed fp32 beside MFMA waves all the time and stay bit-stable beside other kernels (tests/test_gpu_coresident.py).
Round 4 bisected the real kernels by instruction class (profiles/r04_vpk_bisect.txt: classes of packed instructions of the SLP listing
rewritten in place into scalar pairs, schedule kept): what goes wrong is v_pk_add_f32 / v_pk_mul_f32 with a non-default `op_sel` - a low
result lane reading the HIGH dword of a source pair, the crosswise reads of complex butterflies; plain forms, neg_lo / neg_hi,
op_sel_hi alone, every v_pk_fma_f32 form and v_pk_mov_b32 were clean in 60-120 runs each.
The library is therefore built without the SLP vectorizer, and this test holds two rules on the code that ships: a kernel that
issues no MFMA contains no packed-fp32 instruction at all (the rule of rounds 2-3, kept), and NO kernel contains a v_pk_add_f32 /
v_pk_mul_f32 with an op_sel modifier (the hand-written packed code of the MFMA kernels uses no modifier or op_sel_hi only)."""
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
def test_packed_fp32_only_in_kernels_that_issue_mfma(tmp_path):
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    objs = [(t, b) for t, b in device_code_objects(LIB) if "gfx950" in t and len(b) > 0]
    assert objs, "no gfx950 code object found in %s" % LIB
    n_kernels = n_pk = 0
    for i, (triple, data) in enumerate(objs):
        f = tmp_path / ("dev%d.co" % i)
        f.write_bytes(data)
        asm = subprocess.run([OBJDUMP, "-d", str(f)], capture_output=True, text=True, check=True).stdout
        cur, stats = None, {}
        for ln in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                cur = m.group(1)
                stats[cur] = [0, 0, 0]
            elif cur is not None:
                stats[cur][0] += "v_mfma" in ln
                stats[cur][1] += bool(re.search(r"\bv_pk_(add|mul|fma)_f32\b|\bv_pk_mov_b32\b", ln))
                stats[cur][2] += bool(re.search(r"\bv_pk_(add|mul)_f32\b.*\bop_sel:\[", ln))
        bad = sorted(k for k, (mfma, pk, sel) in stats.items() if pk and not mfma)
        assert not bad, "packed fp32 in kernels without MFMA (%s): %s" % (triple, bad[:5])
        crosswise = sorted(k for k, (mfma, pk, sel) in stats.items() if sel)
        assert not crosswise, "v_pk_add_f32 / v_pk_mul_f32 with op_sel (crosswise read: profiles/r04_vpk_bisect.txt) in (%s): %s" % (triple, crosswise[:5])
        n_kernels += len(stats)
        n_pk += sum(1 for v in stats.values() if v[1])
    # the disassembly worked: the DSP kernels are there, and the GEMM split / LSTM cells do use packed fp32
    assert n_kernels > 20 and n_pk > 0


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")), reason="llvm-readelf of the ROCm toolchain not found")
def test_no_kernel_uses_scratch(tmp_path):
    """No shipped kernel spills registers or keeps arrays in scratch memory.  Round 3: the time-axis kernel's instantiation with 9-14
    spilled registers (reloaded inside its loops) returned a result a few ulp off about once in 40 calls when a call ran beside the
    first kernels of a second process - never alone, never in the steady state beside a running load, and not at all once the kernel
    fitted its 128 VGPRs (NOTEBOOK.md R3.11, profiles/r03_busy_start.txt).  Whether scratch itself is fragile there is not established;
    the rule costs nothing: every kernel of the library fits its register budget."""
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    n = 0
    offenders = []
    for i, (triple, data) in enumerate(device_code_objects(LIB)):
        if "gfx950" not in triple or not data:
            continue
        f = tmp_path / ("dev%d.co" % i)
        f.write_bytes(data)
        notes = subprocess.run([readelf, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            spilled = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))      # (SGPR spills go to VGPR lanes, not to memory)
            dynamic = re.search(r"\.uses_dynamic_stack:\s+(\w+)", blk).group(1)
            n += 1
            if scratch or spilled or dynamic != "false":
                offenders.append((name, scratch, spilled, dynamic))
    assert n > 20, "no kernel metadata found"
    assert not offenders, "kernels with scratch / spills: %s" % offenders[:5]
