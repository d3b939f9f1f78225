#!/usr/bin/env python3
"""Static VALU instruction-class histogram of the ORB32 / matcher kernels, from the gfx950 ISA hipcc emits for the committed sources
(no GPU needed): python tools/isa_valu_classes.py [out.json]

Why: bench.py's `valu_issue` compares SQ_INSTS_VALU x frames/s with an issue rate, and the rate depends on the op class
(tools/calib_valu.hip: plain 32-bit ops 9.5e11 wave-instr/s, packed i16 6.2e11, v_pk_mad_u16 4.8e11, dot4 6.0e11, popcount 4.6e11).
With the class shares of every kernel (static: every instruction of the kernel text counted once, loops not weighted) and the
per-kernel dynamic counts of the PMC pass, the pipeline gets ONE blended peak: sum_k N_k / sum_k N_k * sum_c share_kc / rate_c.
Classes without a calibration loop (3-source integer ops, f64, transcendental, conversions) are priced at the plain rate, so the
blended peak is an upper bound and the fraction a lower bound.  The file carries the sha of the sources (csrc_sha, tools/csrc_sha.py)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from csrc_sha import csrc_sha  # noqa: E402

SOURCES = ["k_pyramid.hip", "k_fast.hip", "k_harris.hip", "k_select.hip", "k_describe.hip", "k_match.hip", "k_match_mfma.hip"]
KERNELS = ["k_resize_level", "k_fast_nms", "k_retain_score", "k_harris", "k_select_quadtree", "k_describe", "k_match_topk_mfma", "k_match_topk",
           "k_match_resolve"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]


def classify(m):
    if m.startswith("v_mfma") or m.startswith("v_accvgpr"):
        return "mfma"
    if m.startswith("v_pk_mad") or m.startswith("v_pk_mul") or m.startswith("v_pk_fma"):
        return "pk_mad"
    if m.startswith("v_pk_"):
        return "pk_i16"
    if m.startswith("v_dot"):
        return "dot"
    if m.startswith("v_bcnt"):
        return "bcnt"
    if m.endswith("_f64") or "_f64_" in m:
        return "f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", m):
        return "trans"
    return "plain"


def histogram(asm_text):
    out = {}
    cur = None
    for line in asm_text.splitlines():
        mm = re.match(r"^(_Z\w+|k_\w+):", line)
        if mm:
            name = mm.group(1)
            hit = [k for k in KERNELS if re.search(r"\d+%s[A-Z]" % k, name) or name == k]
            cur = max(hit, key=len) if hit else None
            if cur:
                out.setdefault(cur, collections.Counter())
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur:
            t = line.strip().split()
            if t and t[0].startswith("v_"):
                out[cur][classify(t[0])] += 1
    return out


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for s in SOURCES:
            o = os.path.join(td, s + ".s")
            subprocess.run([hipcc] + FLAGS + [os.path.join(ROOT, "anyfeature-vslam_amd", "csrc", s), "-o", o], check=True,
                           stderr=subprocess.DEVNULL)
            for k, c in histogram(open(o).read()).items():
                res.setdefault(k, collections.Counter()).update(c)
    doc = {"csrc_sha": csrc_sha(), "kernels": {}}
    for k, c in sorted(res.items()):
        tot = sum(c.values())
        doc["kernels"][k] = {"static_valu_instructions": tot, "counts": dict(c), "shares": {a: b / tot for a, b in c.items()}}
    doc["note"] = "static counts over the kernel text (every template instance of a kernel summed); loops are not weighted"
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03", "isa_valu_classes.json")
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in doc["kernels"].items():
        print(k.ljust(20), v["static_valu_instructions"], {a: round(b, 3) for a, b in sorted(v["shares"].items())})


if __name__ == "__main__":
    main()
