"""Build libafv_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and `python build.py`.

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off   one rounding per float operator (Harris response, fastAtan2, rBRIEF rotation, epipolar tests)
  -packed-fp32-ops    (target feature OFF) no v_pk_*_f32: see FLAGS
  (no -ffast-math)    IEEE division / sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt stays on)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libafv_hip.so")
SOURCES = ["k_pyramid.hip", "k_fast.hip", "k_harris.hip", "k_select.hip", "k_describe.hip", "k_match.hip", "k_match_mfma.hip", "k_project.hip", "k_bow.hip", "k_match_l2.hip", "k_frame.hip",
           "afv_api.hip", "afv_comm.hip", "afv_frame.hip", "k_akaze.hip", "k_akaze_detect.hip", "k_akaze_desc.hip", "akaze_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function",
         # no packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32), in any kernel: with an operand broadcast
         # (op_sel) they return wrong lanes 48..63 while an MFMA kernel of another queue shares the SIMD (round 6, DESIGN_LOG;
         # tools/probes/probe_pk_real.hip).  tests/test_isa_rules.py checks the built code objects.
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


# Translation units built WITHOUT the target-feature switch.  The switch also moves the register allocator: k_match_topk_mfma<false> grows
# from 168 to 177 vector registers = from 3 to 2 wavefronts per SIMD (kernel + 11 %, pairs10k 3.49 -> 3.17 M jobs/s: found in round 6's last
# collection; with __launch_bounds__(256, 3) it spills instead); every other kernel keeps or gains occupancy with the switch.  The file
# holds integer / MFMA code only - no v_pk_*_f32 either way, which tests/test_isa_rules.py checks on the BUILT code objects of the whole
# library, this file's included.
NO_FEATURE_SWITCH = {"k_match_mfma.hip"}
_SWITCH = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _flags_for(source):
    if source in NO_FEATURE_SWITCH:
        assert FLAGS[-4:] == _SWITCH
        return FLAGS[:-4]
    return FLAGS


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run_filtered(cmd):
    """hipcc hands -target-feature to the HOST pass of a .hip file too, where x86 says "'-packed-fp32-ops' is not a recognized feature for
    this target (ignoring feature)" once per function: dropped from what is shown, everything else passes through"""
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    for line in r.stderr.splitlines():
        if "'-packed-fp32-ops' is not a recognized feature for this target" not in line:
            print(line, file=sys.stderr)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)


def build(force=False, verbose=False, extra_flags=(), out=None, objdir=None):
    """extra_flags/out/objdir: kernel experiments (tools/experiments.py) build variant libraries next to the real one"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = objdir or os.path.join(HERE, "build")
    out = out or OUT
    os.makedirs(objdir, exist_ok=True)
    import glob
    # every header / table any translation unit may include: a stale object silently breaks the bit-exact parity tests
    headers = (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
               glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)])
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + _flags_for(s) + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            _run_filtered(cmd)
    if force or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
