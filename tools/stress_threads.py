#!/usr/bin/env python3
"""Round-6 diagnostic: the three-threads scene of tests/test_gpu_match.py::test_three_threads_three_contexts, run for many rounds WITHOUT
stopping at the first mismatch; prints one line per wrong descriptor row (thread, call, row, level, differing bit positions).
  python tools/stress_threads.py [--rounds N] [--lib path] [--threads 3] [--no-match] [--no-close]"""
import argparse
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--no-match", action="store_true", help="extraction only")
    ap.add_argument("--keep-contexts", action="store_true", help="one context per thread for the whole run (no create / destroy while others work)")
    ap.add_argument("--same-image", action="store_true", help="every thread extracts the same two images")
    ap.add_argument("--roles", default=None, help="one letter per thread: e = extraction only, m = matching only, b = both (default: b for all)")
    ap.add_argument("--engine", type=int, default=None, help="afv_set_match_engine of the thread contexts (0 popcount, 1 MFMA)")
    ap.add_argument("--resolve", type=int, default=None, help="afv_set_match_resolve of the thread contexts")
    ap.add_argument("--small", type=int, default=None, help="afv_set_small_batch_path mode of the thread contexts")
    args = ap.parse_args()
    afv = importlib.import_module("anyfeature-vslam_amd")
    if args.lib:
        afv._lib.use_library(args.lib)
    s = afv.synth
    nt = args.threads
    imgs = [s.corners_frame(40 + (0 if args.same_image else i)) for i in range(nt)]
    serial = afv.Context()
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    want = []
    for im in imgs:
        k1, d1 = serial.extract(im)
        k2, d2 = serial.extract(np.roll(im, 3, axis=1))
        m = afv.FeatureMatcher(0.7, True, ctx=serial)
        want.append((k1, d1, k2, d2, m.SearchByBoW(afv.FeatureView(d1, angles=k1["angle"]), afv.FeatureView(d2, angles=k2["angle"]))))
    lock = threading.Lock()
    stats = {"calls": 0, "bad_rows": 0, "bad_kps": 0, "bad_match": 0, "bad_count": 0, "bad_akaze": 0}
    akz_want = None
    if args.roles and "a" in args.roles:  # role a: AKAZE61 extraction (its kernels carried ~1200 packed-fp32 instructions with op_sel before round 6)
        akz = importlib.import_module("anyfeature-vslam_amd.akaze")
        a0 = akz.AkazeContext(akz.default_params(max_width=640, max_height=480))
        akz_want = [a0.extract(im) for im in imgs]
        a0.close()
    ctxs = [afv.Context() for _ in range(nt)] if args.keep_contexts else None

    def check(i, which, k, d, wk, wd, rnd):
        bad = 0
        if len(k) != len(wk):
            with lock:
                stats["bad_count"] += 1
                print("thread %d round %d call %s: count %d / %d" % (i, rnd, which, len(k), len(wk)), flush=True)
            return
        if (k["class_id"] != -1).any():  # the self-checking diagnosis build (-DKD_SELFCHECK) flags rows whose two evaluations disagreed
            rows = np.nonzero(k["class_id"] != -1)[0]
            with lock:
                stats["selfcheck_rows"] = stats.get("selfcheck_rows", 0) + len(rows)
                for r in rows[:8]:
                    print("thread %d round %d call %s: row %d level %d self-check code 0x%04x, final row %s" % (
                        i, rnd, which, r, k["octave"][r], k["class_id"][r] & 0xffff, "right" if np.array_equal(d[r], wd[r]) else "WRONG"), flush=True)
            k = k.copy()
            k["class_id"] = -1
        if k.tobytes() != wk.tobytes():
            with lock:
                stats["bad_kps"] += 1
                print("thread %d round %d call %s: keypoints differ" % (i, rnd, which), flush=True)
        rows = np.nonzero((d != wd).any(axis=1))[0]
        for r in rows:
            x = np.unpackbits(d[r] ^ wd[r], bitorder="little")
            with lock:
                stats["bad_rows"] += 1
                print("thread %d round %d call %s: row %d level %d bits %s" % (i, rnd, which, r, k["octave"][r], np.nonzero(x)[0].tolist()), flush=True)

    def work(i, rnd):
        ctx = ctxs[i] if ctxs else afv.Context()
        if args.engine is not None:
            ctx.set_match_engine(args.engine)
        if args.resolve is not None:
            ctx.set_match_resolve(args.resolve)
        if args.small is not None:
            ctx.set_small_batch_path(args.small)
        role = (args.roles or "b" * nt)[i]
        m = afv.FeatureMatcher(0.7, True, ctx=ctx)
        if role == "a":
            actx = akz.AkazeContext(akz.default_params(max_width=640, max_height=480))
            for _ in range(args.iters):
                k, d = actx.extract(imgs[i])
                wk, wd = akz_want[i]
                if k.tobytes() != wk.tobytes() or not np.array_equal(d, wd):
                    with lock:
                        stats["bad_akaze"] += 1
                        rows = np.nonzero((d != wd).any(axis=1))[0][:8].tolist() if d.shape == wd.shape else None
                        print("thread %d round %d: AKAZE61 extraction differs (counts %d / %d, descriptor rows %s)" % (i, rnd, len(k), len(wk), rows), flush=True)
                with lock:
                    stats["calls"] += 1
            actx.close()
            if not ctxs:
                ctx.close()
            return
        rolled = np.roll(imgs[i], 3, axis=1)
        for _ in range(args.iters):
            if role in "eb":
                k1, d1 = ctx.extract(imgs[i])
                check(i, "frame", k1, d1, want[i][0], want[i][1], rnd)
                k2, d2 = ctx.extract(rolled)
                check(i, "shifted", k2, d2, want[i][2], want[i][3], rnd)
            if not args.no_match and role in "mb":
                r = m.SearchByBoW(afv.FeatureView(want[i][1], angles=want[i][0]["angle"]), afv.FeatureView(want[i][3], angles=want[i][2]["angle"]))
                if not (r[1] == want[i][4][1] and np.array_equal(r[0], want[i][4][0])):
                    with lock:
                        stats["bad_match"] += 1
                        print("thread %d round %d: SearchByBoW differs (%d / %d matches, %d entries differ)" % (
                            i, rnd, r[1], want[i][4][1], int((r[0] != want[i][4][0]).sum())), flush=True)
            with lock:
                stats["calls"] += 2
        if not ctxs:
            ctx.close()

    for rnd in range(args.rounds):
        th = [threading.Thread(target=work, args=(i, rnd)) for i in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
    print("SUMMARY", vars(args), stats, flush=True)


if __name__ == "__main__":
    main()
