"""Rules the BUILT device code has to obey, checked on the code objects inside libafv_hip.so (no GPU needed: the library is disassembled).

Round 6: packed-fp32 instructions with a broadcast operand (v_pk_mul_f32 ... op_sel) returned wrong lanes 48..63 in k_describe while the MFMA
matcher of another context shared the SIMD (DESIGN_LOG round 6, tools/probes/probe_pk_real.hip).  build.py therefore switches the target
feature off for every translation unit; a kernel that gets them back (a dropped flag, a variant build that replaces the library) fails here."""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "anyfeature-vslam_amd", "libafv_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(path):
    """every gfx950 ELF of every offload bundle embedded in a host library (one bundle per translation unit)"""
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
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    return out


@pytest.fixture(scope="module")
def disassembly():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump")
    objs = device_code_objects(LIB)
    assert len(objs) >= 12, "expected one code object per kernel translation unit (14 in round 6), found %d" % len(objs)
    d = tempfile.mkdtemp()
    text = []
    try:
        for i, o in enumerate(objs):
            f = os.path.join(d, "co%d.elf" % i)
            open(f, "wb").write(o)
            text.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], check=True, capture_output=True, text=True).stdout)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return "\n".join(text)


def test_no_packed_fp32_instruction_in_any_kernel(disassembly):
    # (v_pk_mov_b32, the compiler's 64-bit register copy, stays: both of its op_sel forms were measured clean by the same probe)
    bad = sorted(set(re.findall(r"\b(v_pk_(?:mul|add|fma)_f32)\b", disassembly)))
    assert not bad, "packed-fp32 instructions in the built library: %s (build.py: -target-feature -packed-fp32-ops)" % bad


def test_the_disassembly_is_the_real_thing(disassembly):
    # the rule above must not pass on an empty or host-only listing
    for kernel in ("k_describe", "k_fast_nms", "k_match_topk_mfma", "k_akz_fed_gauss"):
        assert kernel in disassembly, kernel
    assert "v_mfma_i32_32x32x32_i8" in disassembly
