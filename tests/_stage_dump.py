"""Per-stage state of the last one-frame extraction of a context (through the afv_debug_get_* getters), and the first stage at which
two such states differ.  Used by the concurrency tests to NAME the kernel that left the oracle, should one ever do so."""
import numpy as np


def stage_state(ctx, frame=0):
    g = ctx.geometry()
    st = {}
    for l in range(g["nlevels"]):
        st["pyramid level %d" % l] = ctx.debug_level(frame, l)
    for l in range(g["nlevels"]):
        x, y, s, r = ctx.debug_candidates(frame, l)
        o = np.lexsort((x, y))
        st["FAST + NMS candidates (x, y, score) level %d" % l] = np.stack([x[o], y[o], s[o]], 1)
        keep = r[o] != 0
        st["retainBest + Harris (x, y, response bits) level %d" % l] = np.stack([x[o][keep], y[o][keep], r[o][keep].view(np.int32)], 1)
    for l in range(g["nlevels"]):
        x, y, r = ctx.debug_selected(frame, l)
        st["quadtree survivors (x, y, response bits, list order) level %d" % l] = np.stack([x, y, r.view(np.int32)], 1)
    return st


def first_difference(a, b):
    """name of the first stage (pipeline order) whose arrays differ, with a short description; None when all agree"""
    for k in a:
        if a[k].shape != b[k].shape:
            return "%s: shapes %s / %s" % (k, a[k].shape, b[k].shape)
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(a[k] != b[k])
            return "%s: %d elements differ, first at %s (%s / %s)" % (k, len(bad), bad[0].tolist(), a[k][tuple(bad[0])], b[k][tuple(bad[0])])
    return None
