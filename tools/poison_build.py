#!/usr/bin/env python3
"""Builds the TEST-ONLY poison variant of the library (csrc/afv_poison.h): every allocation filled with a byte, the LDS of every CU
filled before every kernel launch.  Output: anyfeature-vslam_amd/build_exp/libafv_poison.so (git-ignored; travels to the GPU box).
The tests bind it when AFV_TEST_LIB names it (tests/conftest.py); the product loader never looks at the environment."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anyfeature-vslam_amd")
EXP = os.path.join(PKG, "build_exp")


def build(force=False):
    spec = importlib.util.spec_from_file_location("afv_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = os.path.join(EXP, "poison")
    os.makedirs(d, exist_ok=True)
    flags = ["-DAFV_POISON", "-include", os.path.join(PKG, "csrc", "afv_poison.h")]
    out = mod.build(force=force, extra_flags=flags, out=os.path.join(EXP, "libafv_poison.so"), objdir=d)
    # stamp of the sources this library was built from: tools/poison_suite.sh rebuilds when it is not the current one (a stale poison
    # library lacks the entry points added since and fails every test at load time)
    import subprocess
    sha = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "csrc_sha.py")], capture_output=True, text=True).stdout.strip()
    with open(os.path.join(EXP, "libafv_poison.sha"), "w") as fh:
        fh.write(sha + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
