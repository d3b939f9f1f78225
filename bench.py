#!/usr/bin/env python3
"""bench.py — ORB32 keypoints extracted + described + brute-force Hamming-matched per second (BASELINE.json metric).

Workload (BASELINE.json configs[1]): synthetic 640x480 'corners' frames (LCG, seed = 1 + global frame index), 1000
keypoints/frame budget, ORB32 defaults (8 levels, scale 1.2, FAST 20).  One STEP = one pass of the hot path over one
batch of B (default 512) frames already resident in HBM: pyramid -> FAST+NMS+Harris -> retainBest x2 + quadtree -> IC + blur +
rBRIEF, then SearchByBoW(KF,KF) brute force (TH_LOW 75, nnratio 0.6, orientation check) of frame t against frame
t-1 (frame 0 against frame B-1) — all on the device, nothing returns to the host inside the timed region.
Unit of work = one output keypoint (extracted, described, matched).  value = keypoints of all ranks / wall time.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), frames sharded by rank, no data-path
collective (weak scaling: B frames per GPU per step); barrier + synchronize around the timed region, max over ranks.

Extra objects on the JSON line:
  roofline      dominant kernel (k_fast_nms): algorithmic bytes (every pyramid pixel read once = 950 532 B/frame at
                640x480) x frames per launch / mean launch duration measured with hipEvents recorded on
                the launch stream inside the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference cannot be built here) timed single-threaded on
                rank 0 on a bounded sample of the same frames, reference-faithful call pattern (variant 1).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "keypoints extracted+matched /sec (ORB32, 640x480)"
_REAL_STDOUT = None


def emit(obj):
    """the ONE JSON line of this run, on the process's original stdout"""
    f = _REAL_STDOUT or sys.stdout
    f.write(json.dumps(obj) + "\n")
    f.flush()

HBM_PEAK_GBS = 8000.0
W, H = 640, 480


def level_pixels(w, h, nlevels=8, scale=1.2):
    tot = 0
    for l in range(nlevels):
        s = np.float32(np.power(np.float64(np.float32(scale)), l))
        inv = np.float32(1.0) / s
        tot += int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv))
    return tot


PROF_EVERY = 4   # stage events on every 4th timed step


def _csrc_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from csrc_sha import csrc_sha
        return csrc_sha(ROOT)
    except Exception:
        return None


def _newest_profile(name):
    """newest committed profiles/r*/<name> (collected by tools/collect_profiles.sh) and whether it still matches the sources:
    every such file carries the sha of csrc/ + include/ it was measured on"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        sha = _csrc_sha()
        d["_path"] = os.path.relpath(p, ROOT)
        d["_stale"] = (d.get("csrc_sha") is None) or (sha is not None and d.get("csrc_sha") != sha)
        return d
    return None


def pmc_traffic(kernel, batch):
    """HBM bytes per FRAME of `kernel` from the newest committed rocprofv3 PMC passes (profiles/r*/traffic_pmc.json: FETCH_SIZE
    and WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read
    from inside a normal run, so the figure is only valid for the source tree it was collected on: (None, path) when stale."""
    d = _newest_profile("traffic_pmc.json")
    if d is None or kernel not in d.get("kernels", {}):
        return None, None, False
    return d["kernels"][kernel]["hbm_bytes_per_frame"], d["_path"], d["_stale"]


def valu_pmc():
    return _newest_profile("valu_pmc.json")


def cpu_baseline(afv, nframes, seed0):
    """oracle (kind "port": the reference cannot be built here or on the GPU box), ONE thread pinned to one core,
    steady clock, same frames as the GPU batch.  Two variants as BASELINE.md section 3 asks:
      A  reference-faithful call pattern (Feature_orb32.cpp:42-53: 1 detect pyramid + 8 cv::ORB::compute passes that each
         rebuild and blur levels 0..L = 36 level builds + 36 blurs per frame)  -> `value`
      B  de-duplicated (8 builds, 8 blurs)                                        -> `dedup_value`
    each followed by the brute-force SearchByBoW(KF,KF) match against the previous frame."""
    import oracle
    try:
        oracle.lib(oracle.build(native=True))  # re-tuned for this host's CPU (gcc -O3 -march=native -ffp-contract=off)
    except Exception:
        oracle.lib()
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    reps = 5                                                    # 1 warm-up + >= 5 repetitions, median
    per_rep = max(nframes // (2 * reps), 2)
    frames = [afv.synth.corners_frame(seed0 + i) for i in range(per_rep)]
    oracle.orb_extract(frames[0])  # warm-up

    def run(variant, fr):
        t0 = time.perf_counter()
        prev, nk = None, 0
        for f in fr:
            kps, desc = oracle.orb_extract(f, variant=variant)
            if prev is not None:
                oracle.search_by_bow_kf_kf(desc, prev[1], angle1=kps["angle"], angle2=prev[0]["angle"], th_low=75.0, nnratio=0.6,
                                           check_orientation=True)
            prev = (kps, desc)
            nk += len(kps)
        dt = time.perf_counter() - t0
        return nk / dt, dt

    ra = sorted(run(1, frames) for _ in range(reps))
    rb = sorted(run(0, frames) for _ in range(reps))
    va, ta = ra[reps // 2]
    vb, tb = rb[reps // 2]
    return {"value": va, "unit": "keypoints/s", "cores": 1, "kind": "port", "dedup_value": vb,
            "min": ra[0][0], "max": ra[-1][0], "repetitions": reps,
            "sample": "median of %d repetitions over the same %d frames 640x480 corners (%.1f s each): variant A = reference call pattern "
                      "(1 detect pyramid + 8 compute passes); dedup_value = variant B (8 builds, 8 blurs, %.1f s per repetition); extraction + "
                      "brute-force match vs previous frame; 1 thread pinned; host %s, %d logical cores"
                      % (reps, per_rep, ta, tb, cpu, os.cpu_count() or 0),
            "ms_per_frame": 1e3 * ta / per_rep, "dedup_ms_per_frame": 1e3 * tb / per_rep}


# ---------------------------------------------------------------------------------------------------------------------
# --workload akaze61 (BASELINE.json configs[4]: AKAZE61, 1280x720, 1 GPU) — a secondary line, not the contract metric
# ---------------------------------------------------------------------------------------------------------------------
def akaze_cpu_baseline(afv, frames, quotas):
    """oracle/akaze.c + the oracle quadtree, one thread (kind "port"; the libAKAZE fork cannot be built here)"""
    from oracle import akaze_binding as ak
    from oracle import binding as ob
    h, w = frames[0].shape
    op = ak.make_plan(w, h)
    t0 = time.perf_counter()
    tot = 0
    for fr in frames:
        levels, _ = ak.full_evolution(fr, op)
        kp = ak.subpixel(op, levels, ak.find_extrema(op, levels))
        chosen = []
        for lvl in range(op.nlevels):
            idx = np.nonzero(kp["class_id"] == lvl)[0]
            if len(idx):
                chosen.append(idx[ob.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(quotas[lvl]), w, h, tiebreak=np.arange(len(idx)))])
        kk, _ = ak.compute_descriptors(op, levels, kp[np.concatenate(chosen)])
        tot += len(kk)
    ct = time.perf_counter() - t0
    return {"value": tot / ct, "unit": "keypoints/s", "cores": 1, "kind": "port", "ms_per_frame": ct / len(frames) * 1e3,
            "sample": "%d frames %dx%d through oracle/akaze.c + oracle quadtree, single thread" % (len(frames), w, h)}


def akaze_algorithmic_bytes(plan, Wa, Ha):
    """Algorithmic HBM bytes per frame of scale space + Hessian, each datum moved once: level 0 reads the u8 frame twice (Gaussian
    and contrast percentile) and writes Lt; every further level reads the previous Lt, writes Lsmooth and Lt; the Hessian reads
    Lsmooth and writes Ldet (round 5: the first derivatives are no longer stored - before, 16 B per pixel with Lx and Ly: 126.3 MB per
    1280 x 720 frame, now 89.4).  The kernel structure moves more (second value: gauss, level kernel, derivative kernel)."""
    px0 = Wa * Ha
    strict = 2 * px0 + 4 * px0
    kern = px0 + 4 * px0 + px0 + 4 * px0 + 4 * px0 + 4 * px0 + 4 * px0
    for i in range(1, plan.nlevels):
        L, Q = plan.lv[i], plan.lv[i - 1]
        n = L.w * L.h
        strict += (4 * Q.w * Q.h if L.octave > Q.octave else 4 * n) + 8 * n
        if L.octave > Q.octave:
            kern += 4 * Q.w * Q.h + 4 * n
        kern += 8 * n + 12 * n
    for i in range(plan.nlevels):
        n = plan.lv[i].w * plan.lv[i].h
        strict += 8 * n
        kern += 8 * n
    return strict, kern


def akaze_main(args):
    import torch
    afv = importlib.import_module("anyfeature-vslam_amd")
    B, steps = args.batch, args.steps
    Wa, Ha = 1280, 720
    ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
    frames_h = afv.synth.corners_batch(1, B, Wa, Ha)
    frames = torch.from_numpy(frames_h).cuda()
    for _ in range(max(args.warmup, 1)):
        ctx.extract_device(frames)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.extract_device(frames)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nk = sum(len(ctx.features(f)[0]) for f in range(B))
    det = sum(len(ctx.keypoints(f)) for f in range(B))
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.scale_space_device(frames)
    ctx.synchronize()
    dt_ss = (time.perf_counter() - t0) / steps
    strict, kern = akaze_algorithmic_bytes(ctx.plan, Wa, Ha)
    out = {"metric": "keypoints extracted+described /sec (AKAZE61, 1280x720)", "value": nk / dt, "unit": "keypoints/s", "n_gpus": 1, "steps": steps,
           "warmup": max(args.warmup, 1), "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": "AKAZE61 1280x720 synthetic corners frames, omax 2 x 4 sublevels, dthreshold 0.0005, 1000-feature quadtree, MLDB-486",
                      "frames_per_gpu_per_step": B, "detected_per_frame": det / B, "described_per_frame": nk / B},
           "frames_per_s": B / dt, "scale_space_ms_per_step": dt_ss * 1e3,
           "roofline": {"bound": "hbm", "kernel": "scale space + Hessian (k_akz_*)", "achieved": strict * B / dt_ss / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": strict * B / dt_ss / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_frame": strict, "kernel_structure_bytes_per_frame": kern,
                        "kernel_structure_GBps": kern * B / dt_ss / 1e9}}
    try:
        out["single_frame"] = akaze_single_frame()
    except Exception as e:
        out["single_frame"] = {"error": str(e)[:200]}
    if args.cpu_frames > 0:
        out["cpu_baseline"] = akaze_cpu_baseline(afv, frames_h[:min(args.cpu_frames, B, 4)], ctx.quotas())
    emit(out)



# ---------------------------------------------------------------------------------------------------------------------
# --workload pairs10k (BASELINE.json configs[3]: 10 000 keyframe-pair Hamming match jobs over a K = 1000 keyframe table,
# sharded over the GPUs of one node, ONE RCCL broadcast of the table) — the path's only exchange step
# ---------------------------------------------------------------------------------------------------------------------
PAIR_ALG_BYTES = 2 * 32000 + 4000     # two 1000 x 32 B descriptor sets read + 1000 x 4 B matches written per job


def pairs_cpu_baseline(table, angles, counts, pa, pb, budget_s=12.0):
    """oracle SearchByBoW(KF,KF) brute force (FeatureMatcher.cc:561-660), one pinned thread, on the first jobs of the list"""
    import oracle
    try:
        oracle.lib(oracle.build(native=True))
    except Exception:
        oracle.lib()
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass

    def one(i):
        a, b = int(pa[i]), int(pb[i])
        return oracle.search_by_bow_kf_kf(table[a, :counts[a]], table[b, :counts[b]], angle1=angles[a, :counts[a]], angle2=angles[b, :counts[b]],
                                          th_low=75.0, nnratio=0.75, check_orientation=True)[1]
    one(0)
    reps = []
    per_rep = None
    for _ in range(5):                      # warm-up + >= 5 repetitions, median
        t0 = time.perf_counter()
        n = 0
        while True:
            one(n % len(pa))
            n += 1
            if per_rep is not None and n >= per_rep:
                break
            if per_rep is None and time.perf_counter() - t0 > budget_s / 5:
                per_rep = n
                break
        reps.append(n / (time.perf_counter() - t0))
    reps.sort()
    return {"value": reps[len(reps) // 2], "unit": "jobs/s", "cores": 1, "kind": "port", "min": reps[0], "max": reps[-1],
            "sample": "median of 5 repetitions of the first %d jobs (1000 x 1000 brute-force SearchByBoW(KF,KF) with orientation check) through "
                      "oracle/afvo.c, 1 pinned thread, %d logical cores on the host" % (per_rep, os.cpu_count() or 0)}


def table_broadcast_bytes(K, cap):
    """what one replication of the keyframe table moves: descriptors + angles + counts (36 004 000 B at K = 1000, cap = 1000)"""
    return int(K * cap * 32 + K * cap * 4 + K * 4)


def multi_gpu_identity(rank, world, local, dev, dist, backend, comm_size=None, single_device=False):
    """who ran where (the first run on real multi-GPU hardware must verify itself): per rank the device ordinal, its PCI
    bus id, name, host and pid, gathered on rank 0, plus the communicator sizes both layers report and the RCCL version"""
    import socket

    import torch
    pr = torch.cuda.get_device_properties(dev)
    bus = None
    try:
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    me = {"rank": rank, "local_rank": local, "device_ordinal": dev.index, "pci_bus_id": bus, "device_name": pr.name,
          "hbm_GiB": round(pr.total_memory / 2 ** 30, 1), "host": socket.gethostname(), "pid": os.getpid()}
    ranks = [me]
    if world > 1:
        g = [None] * world
        dist.all_gather_object(g, me)
        ranks = g
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    devs = {(r["host"], r["pci_bus_id"] if r["pci_bus_id"] else r["device_ordinal"]) for r in ranks}
    return {"world_size": world, "backend": backend if world > 1 else None,
            "rccl_ranks_seen": {"torch_distributed": (dist.get_world_size() if world > 1 else 1), "afv_comm": comm_size},
            "rccl_version": ver, "ranks": ranks, "distinct_devices": len(devs), "single_device_rehearsal": bool(single_device),
            "one_rank_per_device": len(devs) == world}


def pairs_main(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    out = pairs_run(args, rank, world, local, dev, dist, args.steps, args.warmup, not args.no_profile, args.cpu_frames > 0)
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pairs_run(args, rank, world, local, dev, dist, steps, warmup, profile, want_cpu):
    """config #4 on the ranks of an initialised process group (world 1: none): table built on rank 0, ONE broadcast, 10 000 jobs in all
    (strong scaling), block-partitioned.  Returns the result dict on rank 0, None elsewhere."""
    import torch
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    afv = importlib.import_module("anyfeature-vslam_amd")
    tbl_mod = importlib.import_module("anyfeature-vslam_amd.table")
    dmod = importlib.import_module("anyfeature-vslam_amd.dist")
    K, cap, njobs = args.keyframes, 1000, args.jobs
    TH, RATIO = 75.0, 0.75                                  # LoopClosing.cc:255: FeatureMatcher matcher(0.75, true); TH_LOW = matchingTh
    ctx = afv.Context(max_batch=1, device=local)
    if args.match_engine is not None:
        ctx.set_match_engine(args.match_engine)
    if args.match_resolve is not None:
        ctx.set_match_resolve(args.match_resolve)
    table = tbl_mod.DescriptorTable(ctx, K, cap)
    host = None
    if rank == 0:                                           # the table exists on ONE rank before the exchange step
        host = afv.synth.keyframe_table(K, cap)
        table.upload(*host)
    d_desc, d_ang, d_n = table.device_views()
    torch.cuda.synchronize(dev)

    # ---- the exchange step: RCCL broadcast through the C-ABI communicator (afv_comm_*), timed on its own ----
    def exchange_id(ident):
        if world == 1:
            return ident
        t = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(ident), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    bc = {"bytes": table_broadcast_bytes(K, cap), "via": None}
    comm = None
    if args.backend == "nccl":
        try:
            comm = tbl_mod.Communicator(ctx, rank, world, exchange_id)
            bc["via"] = "afv_table_broadcast (ncclBroadcast through the C-ABI communicator)"
        except Exception as e:  # RCCL not initialisable on this box: say so, use torch.distributed for N > 1
            bc["comm_error"] = str(e)[:200]
    times = []
    for _ in range(max(args.bcast_reps, 1)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if comm is not None:
            dev_ms = table.broadcast(comm, root=0)
        else:
            dmod.broadcast_descriptor_table(d_desc, d_n, src=0)
            if world > 1:
                dist.broadcast(d_ang, src=0)
            dev_ms = None
            bc["via"] = "torch.distributed.broadcast (%s)" % args.backend
        torch.cuda.synchronize(dev)
        times.append((time.perf_counter() - t0) * 1e3)
    table.sync_counts()
    bc["wall_ms_first"] = times[0]
    bc["wall_ms_min"] = min(times)
    bc["device_ms_last"] = dev_ms
    bc["GBps_at_min"] = bc["bytes"] / (min(times) * 1e-3) / 1e9
    bc["note"] = "world size 1: the collective degenerates to a local no-op" if world == 1 else "rank 0 -> all ranks over xGMI"

    # ---- jobs: the same LCG list on every rank, block-partitioned ----
    def jobs_for(kind):
        a, b = dmod.lcg_pairs(12345, njobs, K)
        if kind == "covisible":                             # loop candidates close in the keyframe chain: real match load
            b = ((a.astype(np.int64) + 1 + (b.astype(np.int64) % 3)) % K).astype(np.int32)
        return a, b

    def run(kind, steps, warmup, profile):
        a, b = jobs_for(kind)
        lo, hi = tbl_mod.shard_range(njobs, rank, world)
        pa = torch.from_numpy(a[lo:hi].copy()).to(dev)
        pb = torch.from_numpy(b[lo:hi].copy()).to(dev)
        match = torch.empty((hi - lo, cap), dtype=torch.int32, device=dev)
        nm = torch.empty((hi - lo,), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(dev)

        def step():
            with torch.cuda.stream(side):
                table.match_pairs_device(pa, pb, TH, RATIO, True, match=match, nmatches=nm)

        def barrier():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            step()
        barrier()
        if profile:
            ctx.profile_enable(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        stages = ctx.profile_read() if profile else None
        if profile:
            ctx.profile_enable(False)
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # per-job counts of ALL jobs on every rank (the optional gather of SURVEY.md 8e), outside the timed region
        t1 = time.perf_counter()
        if world > 1:
            allnm = dmod.gather_job_results(nm if args.backend == "nccl" else nm.cpu(), njobs).cpu().numpy()
        else:
            allnm = nm.cpu().numpy()
        gather_ms = (time.perf_counter() - t1) * 1e3
        return float(t.item()), stages, allnm, gather_ms, (a, b), match

    dt, stages, allnm, gather_ms, (ja, jb), match = run("uniform", steps, warmup, profile)
    dt_c, _, allnm_c, _, _, _ = run("covisible", max(steps // 2, 1), 1, False)

    comm_size = None
    if comm is not None:
        try:
            comm_size = int(ctx.lib.afv_comm_size(comm.handle))
        except Exception:
            comm_size = None
    out = None
    if rank == 0:
        jobs_s = njobs * steps / dt
        out = {"metric": "keyframe-pair Hamming match jobs /sec (ORB32, 1000x1000 brute-force SearchByBoW(KF,KF))", "value": jobs_s, "unit": "jobs/s",
               "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[3]: %d (a, b) keyframe-pair jobs drawn by LCG over a table of K = %d keyframes x %d x 32 B "
                                      "(keyframe k+1 = keyframe k with 10 %% bit flips and 30 %% rows replaced); brute-force SearchByBoW(KF,KF), TH_LOW 75, "
                                      "nnratio 0.75, orientation check; table built on rank 0 and replicated with one broadcast; jobs block-partitioned"
                                      % (njobs, K, cap), "jobs_per_step": njobs, "jobs_per_gpu_per_step": tbl_mod.shard_range(njobs, 0, world)[1],
                          "parallelism": "jobs sharded x%d, table replicated" % world,
                          "job_ranges": [list(tbl_mod.shard_range(njobs, r, world)) for r in range(world)], "backend": args.backend if world > 1 else None,
                          "matches_per_job": float(allnm.mean()), "jobs_with_matches": int((allnm > 0).sum())},
               "descriptor_pairs_per_s": jobs_s * cap * cap,
               "broadcast": bc, "gather_ms": gather_ms,
               "covisible": {"jobs_per_s": njobs * max(steps // 2, 1) / dt_c, "matches_per_job": float(allnm_c.mean()),
                             "note": "same table and job count, but b = a + 1..3 (loop candidates that really overlap): loads k_match_resolve"}}
        if stages and stages["match_topk"]["launches"]:
            tk, rs = stages["match_topk"], stages["match_resolve"]
            ms = tk["total_ms"] / tk["launches"]
            jobs_per_launch = tk["units"] / tk["launches"]
            pairs_s = jobs_per_launch * cap * cap / (ms * 1e-3)
            engine = 1 if args.match_engine is None else args.match_engine
            if engine == 1:
                # phase 1 on the matrix cores: 2 x 256 integer ops per descriptor pair (the +-64 contraction of k_match_mfma.hip; the ninth,
                # index-injecting MFMA per 32 x 32 block is overhead and not counted).  Peak: the guide lists no spec figure for i8; its
                # measured 32x32x32 ceiling is 4404 TOPS (MI355X_MICROARCH.md, MFMA table).  Launches of the two alternating streams overlap,
                # so the per-launch figure is a lower bound of what the kernel does alone.
                ach = pairs_s * 512 / 1e12
                out["roofline"] = {"bound": "mfma", "kernel": "k_match_topk_mfma", "achieved": ach, "peak": 4404.0, "unit": "TOP/s", "frac": ach / 4404.0,
                                   "traffic": None, "algorithmic_ops_per_launch": 512.0 * jobs_per_launch * cap * cap, "avg_launch_ms": ms,
                                   "jobs_per_launch": jobs_per_launch, "hbm_algorithmic_GBps": PAIR_ALG_BYTES * jobs_per_launch / (ms * 1e-3) / 1e9,
                                   "note": "exact i8 x i8 -> i32 Hamming contraction; the kernel is co-limited by the lane-private top-4 insertion "
                                           "(1 v_min + 3 v_med3 per distance, about 5.3 cycles each; an MFMA hides only ~2/3 of its 32 cycles behind them: "
                                           "tools/probes/probe_mfma_valu.hip), see DESIGN.md section 4"}
            else:
                ach = PAIR_ALG_BYTES * jobs_per_launch / (ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "k_match_topk", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": None, "algorithmic_bytes_per_launch": PAIR_ALG_BYTES * jobs_per_launch, "avg_launch_ms": ms,
                                   "jobs_per_launch": jobs_per_launch,
                                   "note": "68 000 B per job against 8e6 xor+popcount32: this kernel is integer-VALU-bound by four orders of magnitude; see valu_issue"}
                calib = valu_calibration()
                winst = pairs_s * 16 / 64   # 8 v_xor_b32 + 8 v_bcnt_u32_b32 per descriptor pair and lane; a wave instruction covers 64 lanes
                out["valu_issue"] = {"kernel": "k_match_topk", "achieved": winst, "unit": "wave-instr/s (xor + popcount only)",
                                     "peak": calib["peak"], "frac": winst / calib["peak"], "peak_source": calib["source"],
                                     "descriptor_pairs_per_s_in_kernel": pairs_s,
                                     "note": "the remaining issue slots of the kernel go to the key and the branch-free top-4 insertion (1 + 4 VALU per pair: v_min + 3 v_med3)"}
            out["match_engine"] = "mfma_i8" if engine == 1 else "popcount"
            out["stage_event_ms_per_step_summed_over_concurrent_streams"] = {"match_topk": tk["total_ms"] / steps, "match_resolve": rs["total_ms"] / steps}
        if want_cpu and world == 1 and host is not None:
            out["cpu_baseline"] = pairs_cpu_baseline(host[0], host[1], host[2], ja, jb)
        elif want_cpu:
            out["cpu_baseline"] = None
    ident = multi_gpu_identity(rank, world, local, dev, dist, args.backend, comm_size, args.single_device)
    if comm is not None:
        comm.close()
    table.close()
    ctx.close()
    if rank == 0:
        out["multi_gpu"] = ident
        return out
    return None


def valu_calibration():
    """integer-VALU issue peak in wave-instructions/s: the newest committed tools/calib_valu run (profiles/r*/calib_valu.json), else
    the nominal 1024 SIMDs x 2.4 GHz / 4 cycles"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "calib_valu.json")), reverse=True):
        try:
            d = json.load(open(p))
            cls = d.get("classes_winst_per_s", {})
            if "xor+bcnt" in cls:   # the op class the counted instructions of k_match_topk belong to
                return {"peak": float(cls["xor+bcnt"]), "source": os.path.relpath(p, ROOT) + " (xor+bcnt class)", "classes": cls}
            return {"peak": float(d["valu_peak_winst_per_s"]), "source": os.path.relpath(p, ROOT)}
        except Exception:
            pass
    return {"peak": 1024 * 2.4e9 / 4, "source": "nominal: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}



def host_fed(afv, ctx, frames_h, steps, dev_extract_fps):
    """the reference's real boundary (host images in, host keypoints / descriptors out: FeatureExtractor.cpp:111-129,
    createVocabulary.cpp:161-174) through afv_orb_extract_batch's chunk pipeline, extraction only.  Page-locked and pageable
    caller memory; PCIe bytes = frames in + keypoints / descriptors / counts out."""
    import torch
    B = frames_h.shape[0]
    cap = ctx.cap
    out = {}
    for kind in ("pinned", "pageable"):
        fr = torch.from_numpy(frames_h)
        kps = torch.zeros((B, cap, 7), dtype=torch.float32)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8)
        n = torch.zeros((B,), dtype=torch.int32)
        if kind == "pinned":
            fr, kps, desc, n = fr.pin_memory(), kps.pin_memory(), desc.pin_memory(), n.pin_memory()
        for _ in range(2):
            ctx.extract_batch_host(fr, kps, desc, n)  # warm-up (arena / events / streams / first DMA mapping of the pinned pages)
        if kind == "pinned":                           # the link itself, same buffers: one plain H2D of the batch
            dtmp = torch.empty(fr.shape, dtype=fr.dtype, device="cuda")
            dtmp.copy_(fr, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dtmp.copy_(fr, non_blocking=True)
            torch.cuda.synchronize()
            out["h2d_only_GBps"] = 3 * fr.numel() / (time.perf_counter() - t0) / 1e9
            del dtmp
        reps = []
        for _ in range(steps):
            t0 = time.perf_counter()
            ctx.extract_batch_host(fr, kps, desc, n)
            reps.append(time.perf_counter() - t0)
        dt = sorted(reps)[len(reps) // 2]
        nbytes = B * frames_h.shape[1] * frames_h.shape[2] + B * (cap * 60 + 4)
        out[kind] = {"frames_per_s": B / dt, "ms_per_batch": dt * 1e3, "ms_per_batch_all": [round(r * 1e3, 3) for r in reps], "statistic": "median", "pcie_GBps": nbytes / dt / 1e9, "frac_of_63GBps": nbytes / dt / 63e9,
                     "keypoints": int(n.sum())}
    out["device_resident_extract_frames_per_s"] = dev_extract_fps
    out["pinned_vs_device_resident"] = out["pinned"]["frames_per_s"] / dev_extract_fps
    out["note"] = "extraction only (the host API returns descriptors to the host; matching them is a second call); batch of %d frames in chunks of 64" % B
    return out


def overlap_step(afv, device, B=256, steps=20):
    """the same step on frames that really overlap: frame 2i+1 = frame 2i rolled by 3 px, so every second (t, t-1) pair shares most of
    its keypoints (~500 matches) and the ordered greedy resolve (k_match_resolve: claim / replay rounds) carries a real load"""
    import torch
    ctx = afv.Context(max_batch=B, device=device)
    m = afv.FeatureMatcher(0.6, True, ctx=ctx)
    base = afv.synth.corners_batch(9001, B // 2, W, H)
    fr = np.empty((B, H, W), np.uint8)
    fr[0::2] = base
    fr[1::2] = np.roll(base, 3, axis=2)
    frames = torch.from_numpy(fr).cuda(device)
    cap = ctx.cap
    dev = frames.device
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.empty((B,), dtype=torch.int32, device=dev); st = torch.zeros((1,), dtype=torch.int32, device=dev)
    match = torch.empty((B, cap), dtype=torch.int32, device=dev); nm = torch.empty((B,), dtype=torch.int32, device=dev)
    pa = torch.arange(B, dtype=torch.int32, device=dev)
    pb = (pa + (B - 1)) % B
    side = torch.cuda.Stream(dev)

    def step():
        with torch.cuda.stream(side):
            ctx.extract_batch_device(frames, kps, desc, n, st, cap)
            m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True, match=match, nmatches=nm)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.profile_enable(True)  # the stage figures come from their own steps: the event pairs around every launch are not free
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    st_ = ctx.profile_read()
    ctx.profile_enable(False)
    nmh = nm.cpu().numpy()
    out = {"frames_per_step": B, "ms_per_step": dt * 1e3, "keypoints_per_s": float(n.sum().item()) / dt,
           "matches_per_pair_overlapping": float(nmh[1::2].mean()), "matches_per_pair_unrelated": float(nmh[0::2].mean()),
           "match_topk_ms_per_step": st_["match_topk"]["total_ms"] / steps, "match_resolve_ms_per_step": st_["match_resolve"]["total_ms"] / steps}
    ctx.close()
    return out


def standalone_fast_nms(afv, device, B, steps=8, warm=10):
    """k_fast_nms with the chip to itself: the default step runs as chunks alternating over two streams, so its launches share the CUs with
    the other stream's kernels and `roofline.avg_launch_ms` overstates the kernel; here the same batch goes out as ONE chunk on one
    stream and the library's stage events give the duration of the launch alone"""
    import torch
    ctx = afv.Context(max_batch=B, device=device)
    ctx.set_split_threshold(0x7fffffff)
    frames = torch.from_numpy(afv.synth.corners_batch(1, min(B, 64), W, H)).cuda(device)
    if B > 64:
        frames = frames.repeat((B + 63) // 64, 1, 1)[:B].contiguous()
    cap = ctx.cap
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=frames.device)
    desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=frames.device)
    n = torch.empty((B,), dtype=torch.int32, device=frames.device)
    st = torch.zeros((1,), dtype=torch.int32, device=frames.device)
    side = torch.cuda.Stream(frames.device)
    with torch.cuda.stream(side):
        # the chip has been idle while this context was set up: its clocks ramp over the first ~8 launches (a kernel trace of this very
        # sequence, round 5: 1.50, 1.52, 1.49, 1.45, 1.43, 1.40, 1.37, 1.34 ms) - measured behind the ramp, like the continuous run rocprof sees
        for _ in range(warm):
            ctx.extract_batch_device(frames, kps, desc, n, st, cap)
        ctx.profile_enable(True)
        for _ in range(steps):
            ctx.extract_batch_device(frames, kps, desc, n, st, cap)
        torch.cuda.synchronize()
    fh = ctx.profile_read()["fast_nms"]
    ctx.profile_enable(False)
    ctx.close()
    if not fh["launches"]:
        return None
    ms = fh["total_ms"] / fh["launches"]
    ach = level_pixels(W, H) * (fh["units"] / fh["launches"]) / (ms * 1e-3) / 1e9
    return {"avg_launch_ms": ms, "frames_per_launch": fh["units"] / fh["launches"], "achieved": ach, "frac": ach / HBM_PEAK_GBS}


def batch_sweep(afv, device, sizes=(1, 64, 256, 1024)):
    """SURVEY.md 8d batch sizes: the full step (extract + describe + match t vs t-1), device-resident, a few steps each"""
    import torch
    res = {}
    for B in sizes:
        ctx = afv.Context(max_batch=B, device=device)
        m = afv.FeatureMatcher(0.6, True, ctx=ctx)
        frames = torch.from_numpy(afv.synth.corners_batch(1, min(B, 64), W, H)).cuda(device)
        if B > 64:
            frames = frames.repeat((B + 63) // 64, 1, 1)[:B].contiguous()
        cap = ctx.cap
        kps = torch.empty((B, cap, 7), dtype=torch.float32, device=frames.device)
        desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=frames.device)
        n = torch.empty((B,), dtype=torch.int32, device=frames.device)
        st = torch.zeros((1,), dtype=torch.int32, device=frames.device)
        match = torch.empty((B, cap), dtype=torch.int32, device=frames.device)
        nm = torch.empty((B,), dtype=torch.int32, device=frames.device)
        pa = torch.arange(B, dtype=torch.int32, device=frames.device)
        pb = (pa + (B - 1)) % B
        side = torch.cuda.Stream(frames.device)

        def step():
            with torch.cuda.stream(side):
                ctx.extract_batch_device(frames, kps, desc, n, st, cap)
                m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True, match=match, nmatches=nm)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        reps = 40 if B == 1 else (20 if B <= 256 else 6)
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[str(B)] = {"ms_per_step": dt * 1e3, "keypoints_per_s": float(n.sum().item()) / dt}
        ctx.close()
    return res


# ---------------------------------------------------------------------------------------------------------------------
# compact sub-runs that ride in the default --gpus 1 line (BASELINE.json configs[2], [3], [4] on the driver's record); the
# headline `value` never includes them
# ---------------------------------------------------------------------------------------------------------------------
def extra_pairs10k(afv, device, steps=5):
    """configs[3] on one GPU: K = 1000 keyframes x 1000 x 32 B, 10 000 LCG pair jobs, brute-force SearchByBoW(KF,KF) with orientation
    check; phase 1 on the matrix cores (default) and, for the A/B, on the vector ALU"""
    import torch
    tbl_mod = importlib.import_module("anyfeature-vslam_amd.table")
    dmod = importlib.import_module("anyfeature-vslam_amd.dist")
    K, cap, njobs = 1000, 1000, 10000
    ctx = afv.Context(max_batch=1, device=device)
    table = tbl_mod.DescriptorTable(ctx, K, cap)
    table.upload(*afv.synth.keyframe_table(K, cap))
    a, b = dmod.lcg_pairs(12345, njobs, K)
    dev = torch.device("cuda", device)
    pa, pb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    match = torch.empty((njobs, cap), dtype=torch.int32, device=dev)
    nm = torch.empty((njobs,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(dev)
    out = {"jobs": njobs, "keyframes": K, "steps": steps}
    for engine, name in ((1, "mfma"), (0, "popcount")):
        ctx.set_match_engine(engine)

        def step():
            with torch.cuda.stream(side):
                table.match_pairs_device(pa, pb, 75.0, 0.75, True, match=match, nmatches=nm)
        step()
        torch.cuda.synchronize(dev)
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        st = ctx.profile_read()
        ctx.profile_enable(False)
        out[name] = {"jobs_per_s": njobs / dt, "ms_per_step": dt * 1e3, "descriptor_pairs_per_s": njobs * cap * cap / dt,
                     # stage sums of the two alternating streams (they overlap: the sums exceed ms_per_step)
                     "stage_ms_per_step": {"match_topk": st["match_topk"]["total_ms"] / steps, "match_resolve": st["match_resolve"]["total_ms"] / steps}}
    ctx.set_match_engine(1)
    out["matches_per_job"] = float(nm.float().mean().item())
    # whole-step rates (phase 2 included): 2 * 256 integer ops per descriptor pair on the i8 MFMA against the measured 32x32x32
    # ceiling (MI355X_MICROARCH.md: 4404 TOPS); 8 xor + 8 v_bcnt per pair and lane against the measured issue rate of that op class
    out["mfma"]["i8_mfma_frac_of_4404_TOPS"] = out["mfma"]["descriptor_pairs_per_s"] * 512 / 4404e12
    cal = valu_calibration()
    out["popcount"]["xor_bcnt_issue_frac"] = out["popcount"]["descriptor_pairs_per_s"] * 16 / 64 / cal["peak"]
    out["note"] = "the table broadcast is timed by `--workload pairs10k` (world size 1 here: a no-op)"
    table.close()
    ctx.close()
    return out


def extra_l2_sift128(afv, device, reps=20):
    """configs[2]: float descriptors, L2^2 brute force (cv::norm operation order), 1000 x 1000 x 128, host-buffer call"""
    s = afv.synth
    n, dim = 1000, 128
    a = (s.lcg_bytes(1, n * dim).reshape(n, dim).astype(np.float32)) ** 2
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    noise = (s.lcg_bytes(2, n * dim).reshape(n, dim).astype(np.float32) - 128) / 2000.0
    b = np.abs(a + noise).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b = b[np.argsort(s.lcg_states(3, n), kind="stable")].copy()
    ctx = afv.Context(device=device)
    m = afv.FeatureMatcher(0.8, False, ctx=ctx)
    m.match_l2(a, b, 0.5, 0.8)
    ts = []
    gn = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        _, gn = m.match_l2(a, b, 0.5, 0.8)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out = {"n1": n, "n2": n, "dim": dim, "us_per_job_host_to_host": ts[len(ts) // 2] * 1e6, "us_min": ts[0] * 1e6, "matches": int(gn),
           "note": "afv_match_l2 through host buffers (upload 1 MB, three kernels, download); kernel-only times: profiles/"}
    # the throughput form: a device-resident table of K float descriptor sets, LCG pair jobs (the float counterpart of pairs10k)
    import torch
    dmod = importlib.import_module("anyfeature-vslam_amd.dist")
    K, njobs = 64, 2048
    dev = torch.device("cuda", device)
    table = np.empty((K, n, dim), np.float32)
    for k in range(K):  # set k = the base set, every row perturbed a little more than in set k - 1, rows rotated
        nz = (s.lcg_bytes(100 + k, n * dim).reshape(n, dim).astype(np.float32) - 128) / 4000.0
        t = np.abs(a + nz * (1 + k % 4)).astype(np.float32)
        table[k] = np.roll(t / np.linalg.norm(t, axis=1, keepdims=True), 17 * k, axis=0)
    tt = torch.from_numpy(table).to(dev)
    cnt = torch.full((K,), n, dtype=torch.int32, device=dev)
    ja, jb = dmod.lcg_pairs(777, njobs, K)
    pa, pb = torch.from_numpy(ja).to(dev), torch.from_numpy(jb).to(dev)
    match = torch.empty((njobs, n), dtype=torch.int32, device=dev)
    nm = torch.empty((njobs,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        m.match_l2_pairs_device(tt, cnt, pa, pb, 0.5, 0.8, match=match, nmatches=nm)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        with torch.cuda.stream(side):
            m.match_l2_pairs_device(tt, cnt, pa, pb, 0.5, 0.8, match=match, nmatches=nm)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 3
    # roofline of config #3: f64 vector work.  Per descriptor pair 128 x (v_sub_f32, v_cvt_f64_f32, v_mul_f64, v_add_f64): no FMA - cv::norm's
    # operation order is part of the result (-ffp-contract=off) - so 256 f64 flops ride on 512 vector instructions.  Peaks: 78.6 TFLOP/s f64
    # (an FMA per lane and clock: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x 2), 39.3 T vector instructions per second.  `achieved` covers
    # BOTH launches of a job batch (ranking + ordered resolve) by the wall clock: the ranking kernel alone is faster (profiles/r06/kernel_stats_l2.csv).
    pairs_per_s = njobs * n * n / dt
    out["roofline"] = {"bound": "valu_f64", "kernel": "k_l2_topk_pairs", "achieved": pairs_per_s * 2 * dim / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                       "frac": pairs_per_s * 2 * dim / 1e12 / 78.6, "flops_per_descriptor_pair": 2 * dim,
                       "vector_instructions_per_descriptor_pair": 4 * dim, "issue_rate_achieved_T_per_s": pairs_per_s * 4 * dim / 1e12,
                       "issue_rate_peak_T_per_s": 39.3, "issue_frac": pairs_per_s * 4 * dim / 1e12 / 39.3,
                       "note": "no FMA by contract (cv::norm order): the flop fraction cannot pass 0.5 x 2 / 4 = 0.25; the issue fraction is the one to read"}
    out["pairs_device"] = {"keyframes": K, "jobs": njobs, "jobs_per_s": njobs / dt, "us_per_job": dt / njobs * 1e6,
                           "descriptor_pairs_per_s": njobs * n * n / dt, "matches_per_job": float(nm.float().mean().item()),
                           "note": "afv_match_l2_pairs_device: table resident in HBM, one launch pair per 2048 jobs; every distance is 128 float "
                                   "differences squared and summed in double in cv::norm's order (no MFMA: the order is part of the result)"}
    ctx.close()
    return out


def extra_akaze61(afv, device, B=64, steps=3):
    """configs[4]: AKAZE61 at 1280 x 720, device-resident batch"""
    import torch
    Wa, Ha = 1280, 720
    ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
    frames = torch.from_numpy(afv.synth.corners_batch(1, B, Wa, Ha)).cuda(device)
    ctx.extract_device(frames)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.extract_device(frames)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nk = sum(len(ctx.features(f)[0]) for f in range(B))
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.scale_space_device(frames)
    ctx.synchronize()
    dt_ss = (time.perf_counter() - t0) / steps
    strict = akaze_algorithmic_bytes(ctx.plan, Wa, Ha)[0]
    out = {"frames_per_step": B, "steps": steps, "frames_per_s": B / dt, "ms_per_step": dt * 1e3, "keypoints_per_s": nk / dt,
           "described_per_frame": nk / B, "scale_space_ms_per_step": dt_ss * 1e3, "scale_space_GBps": strict * B / dt_ss / 1e9,
           "scale_space_frac_of_hbm_peak": strict * B / dt_ss / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": strict}
    del ctx
    try:
        out["single_frame"] = akaze_single_frame()
    except Exception as e:
        out["single_frame"] = {"error": str(e)[:200]}
    return out


def extra_tracking_frame_akaze61(afv, device, reps=50):
    """the tracking chain of DESIGN.md section 7 on AKAZE61 features: afv_akaze_extract (640 x 480) -> resident 61-byte frame
    (afv_frame_set_features) -> Frame::ComputeBoW on a 61-byte vocabulary (k = 10, L = 4) -> SearchByProjection(cur, last); every stage a
    synchronous call with host results, timed from Python (the ORB32 chain's figures come from the C++ probe: these carry ~10 us of ctypes / numpy
    per call on top)"""
    import numpy as np
    akz = afv.akaze
    ext = akz.AkazeContext(akz.default_params(max_width=640, max_height=480), device)
    ctx = afv.Context(device=device)
    img = afv.synth.corners_frame(8)
    prev_img = np.roll(img, 4, axis=1)
    ka, da = ext.extract(prev_img)
    sf = np.float32(ext.params.scale_factor)
    za = (sf ** ka["class_id"].astype(np.float32)).astype(np.float32)
    voc = afv.Vocabulary.random(9, k=10, L=4, ctx=ctx, desc_bytes=61)
    cur = afv.Frame(ctx, desc_bytes=61)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(143.0)
    m = afv.FeatureMatcher(0.9, True, ctx=ctx)
    Q = afv.ProjectionQueries(da, ka["x"] - np.float32(4), ka["y"], np.float32(15) * za, za / sf, za * sf, angles=ka["angle"])
    t = {"akaze_extract_us": 0.0, "frame_set_features_us": 0.0, "frame_bow_transform_us": 0.0, "frame_projection_lastframe_us": 0.0}
    nm = 0
    for it in range(reps + 5):
        t0 = time.perf_counter()
        kb, db = ext.extract(img)
        t1 = time.perf_counter()
        zb = (sf ** kb["class_id"].astype(np.float32)).astype(np.float32)
        t1b = time.perf_counter()
        cur.set_features(kb, db, sizes=zb)
        t2 = time.perf_counter()
        cur.bow_transform_nodes(voc, levelsup=2)   # the C call (afv_frame_bow_transform); the BowVector / FeatureVector dictionaries of ComputeBoW are host Python
        t3 = time.perf_counter()
        _, nm = cur.SearchByProjection(m, Q, last_frame=True)
        t4 = time.perf_counter()
        if it >= 5:
            t["akaze_extract_us"] += (t1 - t0) * 1e6 / reps
            t["frame_set_features_us"] += (t2 - t1b) * 1e6 / reps
            t["frame_bow_transform_us"] += (t3 - t2) * 1e6 / reps
            t["frame_projection_lastframe_us"] += (t4 - t3) * 1e6 / reps
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    out = dict(t)
    out["chain_us"] = sum(t.values())
    out["keypoints"] = int(len(ka))
    out["matches"] = int(nm)
    out["stages"] = "afv_akaze_extract 640x480 -> afv_frame_set_features (61-byte rows) -> ComputeBoW (k = 10, L = 4, 61-byte words) -> SearchByProjection(cur, last)"
    cur.close(); voc.close(); ctx.close(); ext.close()
    return out


def extra_tracking_frame_float128(afv, device, reps=50):
    """the same chain on float descriptors (BASELINE config #3's kind of feature: 128 floats, L2^2 - SIFT128; the SIFT extractor itself is
    SiftGPU / OpenGL, out of scope): resident float frame (afv_frame_set_features, 512-byte rows) -> Frame::ComputeBoW on a float vocabulary
    (k = 10, L = 4) -> SearchByProjection(cur, last) with L2^2 distances (ordered-walk engine); keypoints of an ORB frame, SIFT-like unit rows;
    synchronous calls timed from Python like tracking_frame_akaze61 (ComputeBoW = the C call afv_frame_bow_transform; the Python containers are not timed)"""
    import numpy as np
    ctx = afv.Context(device=device)
    s = afv.synth
    img = s.corners_frame(8)
    kb, _ = ctx.extract(img)
    zb, _, _ = ctx.size_sigma(kb)

    def unit(a):
        return np.ascontiguousarray(a / np.linalg.norm(a, axis=1, keepdims=True), np.float32)

    n = len(kb)
    db = unit(s.lcg_bytes(31, n * 128).reshape(n, 128).astype(np.float32) ** 2)
    # the last frame's view of the same points: the rows perturbed, the positions two pixels off (what the motion model predicts)
    da = unit(np.abs(db + (s.lcg_bytes(33, n * 128).reshape(n, 128).astype(np.float32) - 128) / 3000.0))
    voc = afv.Vocabulary.random_float(9, k=10, L=4, ctx=ctx, dim=128)
    cur = afv.Frame(ctx, float_dim=128)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(0.5)
    m = afv.FeatureMatcher(0.9, True, ctx=ctx)
    sf = np.float32(1.2)
    Q = afv.ProjectionQueries(da, kb["x"] + np.float32(2), kb["y"] - np.float32(1), np.float32(15) * zb, zb / sf, zb * sf, angles=kb["angle"])
    t = {"frame_set_features_us": 0.0, "frame_bow_transform_us": 0.0, "frame_projection_lastframe_us": 0.0}
    nm = 0
    for it in range(reps + 5):
        t1 = time.perf_counter()
        cur.set_features(kb, db)
        t2 = time.perf_counter()
        cur.bow_transform_nodes(voc, levelsup=2)   # the C call (afv_frame_bow_transform); the BowVector / FeatureVector dictionaries of ComputeBoW are host Python
        t3 = time.perf_counter()
        _, nm = cur.SearchByProjection(m, Q, last_frame=True)
        t4 = time.perf_counter()
        if it >= 5:
            t["frame_set_features_us"] += (t2 - t1) * 1e6 / reps
            t["frame_bow_transform_us"] += (t3 - t2) * 1e6 / reps
            t["frame_projection_lastframe_us"] += (t4 - t3) * 1e6 / reps
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    out = dict(t)
    out["chain_us"] = sum(t.values())
    out["keypoints"] = int(len(kb))
    out["matches"] = int(nm)
    out["stages"] = "afv_frame_set_features (128-float rows) -> ComputeBoW (float vocabulary k = 10, L = 4) -> SearchByProjection(cur, last), L2^2"
    cur.close(); voc.close(); ctx.close()
    return out


def akaze_single_frame():
    """FeatureExtractor_akaze61::detectAndCompute for ONE 1280 x 720 frame per call, host to host through the C-ABI (tools/akaze_latency.cpp):
    the reference's per-frame operator() (FeatureExtractor.cpp:111-121)"""
    exe = os.path.join(ROOT, "tools", "akaze_latency")
    if not os.path.exists(exe):
        return {"error": "tools/akaze_latency is not built"}
    r = subprocess.run([exe, "100"], capture_output=True, text=True, timeout=120)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            return {"ms_per_call": d["afv_akaze_extract_1280x720_us"] / 1e3, "keypoints": d["keypoints"], "calls": d["reps"],
                    "note": "one synchronous afv_akaze_extract call per frame: pageable host image in, host keypoints + 61-byte descriptors out"}
    return {"error": (r.stderr or r.stdout)[-300:]}


def extra_host_api():
    """the reference's call shape measured from C++ (tools/host_latency.cpp, built by __graft_entry__.build()): afv_orb_extract host to
    host, with a pageable and with a page-locked image, and the brute-force afv_match_bow of two frames through host buffers"""
    exe = os.path.join(ROOT, "tools", "host_latency")
    if not os.path.exists(exe):
        return {"error": "tools/host_latency is not built"}
    r = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=120)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    return {"error": (r.stderr or r.stdout)[-300:]}


def tracking_keys(host_api):
    """The per-frame matcher path of Tracking as driver-visible keys, host to host through the C-ABI (tools/host_latency.cpp,
    300 calls each): `projection` / `initialization` / `fuse` / `bow_transform` against a device-resident Frame (afv_frame_*), the same
    calls through the host-array entry points beside them, and the whole tracking step of one frame.  Device time per kernel:
    profiles/rNN/kernel_stats_tracking.json (rocprofv3 of the same program), attached when it matches the sources."""
    t = (host_api or {}).get("tracking_frame")
    if not t:
        return None
    out = {
        "tracking_frame": {"chain_us": t["chain_us"], "chain_queries_by_reference_us": t["chain_queries_by_reference_us"],
                           "host_array_chain_us": t["host_array_chain_us"], "stages": t["stages"],
                           "frame_extract_us": t["frame_extract_us"], "promote_us": t["promote_us"]},
        "projection": {"resident_frame_lastframe_1000q_us": t["frame_projection_lastframe_1000q_us"],
                       "resident_frame_lastframe_1000q_by_reference_us": t["frame_projection_lastframe_1000q_by_reference_us"],
                       "resident_frame_localmap_2000q_us": t["frame_projection_localmap_2000q_us"],
                       "host_arrays_1000q_us": t["host_array_projection_1000q_us"], "host_arrays_2000q_us": t["host_array_projection_2000q_us"],
                       "matches_lastframe": t["matches_lastframe"], "matches_localmap": t["matches_localmap"],
                       "reference": "SearchByProjection, FeatureMatcher.cc:73-154, :1291-1402"},
        "initialization": {"resident_frames_us": t["initialization_resident_frames_us"], "host_arrays_us": t["initialization_host_arrays_us"],
                           "matches": t["matches_initialization"], "reference": "SearchForInitialization, FeatureMatcher.cc:399-557"},
        "fuse": {"resident_frame_2000q_us": t.get("frame_fuse_2000q_us"), "found": t.get("matches_fuse"), "reference": "Fuse, FeatureMatcher.cc:794-940"},
        "bow_transform": {"resident_frame_us": t["frame_bow_transform_us"], "host_arrays_us": t["host_array_bow_transform_us"],
                          "vocabulary": "k = 10, L = 6, %d nodes (random tree of the shipped shape)" % t["vocabulary_nodes"], "descriptors": host_api.get("keypoints"),
                          "search_by_bow_kf_f_us": t["frame_search_by_bow_kf_f_us"], "reference": "Vocabulary.cpp:156-206, Frame.cc:397-401"},
    }
    ks = _newest_profile("kernel_stats_tracking.json")
    if ks:
        out["device_kernels_us"] = None if ks["_stale"] else {k: round(v["avg_ns"] / 1e3, 2) for k, v in ks["kernels"].items()}
        out["device_kernels_source"] = ks["_path"]
        out["device_kernels_stale"] = bool(ks["_stale"])
    return out


def extra_single_frame(afv, device, reps=200):
    """The plugin shape (Frame.cc:186: ONE frame per call; Tracking.cc:78,84 keeps two extractor instances): latency of one
    640 x 480 frame extracted + matched against its predecessor on one context, and the frame rate when 2 / 4 contexts (each with its own
    streams and scratch) take the frames of a video round robin — every frame still is its own extract + match call, the calls of
    different contexts overlap on the device"""
    import torch
    dev = torch.device("cuda", device)
    frames_h = afv.synth.corners_batch(7001, 8, W, H)
    out = {}
    for nctx, mode in ((1, 1), (2, 1), (4, 1), (1, 0)):   # the last run: the same calls served by the batch kernels (the round-3 figure)
        ctxs = [afv.Context(max_batch=2, device=device) for _ in range(nctx)]
        for c in ctxs:
            c.set_small_batch_path(mode)
        ms = [afv.FeatureMatcher(0.6, True, ctx=c) for c in ctxs]
        cap = ctxs[0].cap
        st = []
        for c in ctxs:   # per context: a two-frame ring (previous + current) and its outputs
            st.append(dict(fr=torch.from_numpy(frames_h[:2].copy()).to(dev), kps=torch.empty((2, cap, 7), dtype=torch.float32, device=dev),
                           desc=torch.empty((2, cap, 32), dtype=torch.uint8, device=dev), n=torch.empty((2,), dtype=torch.int32, device=dev),
                           s=torch.zeros((1,), dtype=torch.int32, device=dev), match=torch.empty((1, cap), dtype=torch.int32, device=dev),
                           nm=torch.empty((1,), dtype=torch.int32, device=dev), pa=torch.tensor([1], dtype=torch.int32, device=dev),
                           pb=torch.tensor([0], dtype=torch.int32, device=dev), stream=torch.cuda.Stream(dev)))
        for c, b in zip(ctxs, st):  # the "previous" frame of every ring
            with torch.cuda.stream(b["stream"]):
                c.extract_batch_device(b["fr"][0:1], b["kps"][0:1], b["desc"][0:1], b["n"][0:1], b["s"], cap)

        def one(i):
            c, m, b = ctxs[i % nctx], ms[i % nctx], st[i % nctx]
            with torch.cuda.stream(b["stream"]):
                c.extract_batch_device(b["fr"][1:2], b["kps"][1:2], b["desc"][1:2], b["n"][1:2], b["s"], cap)
                m.match_pairs_device(b["desc"], b["kps"], b["n"], b["pa"], b["pb"], th_low=75.0, check_orientation=True, match=b["match"], nmatches=b["nm"])
        for i in range(4 * nctx):
            one(i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(reps):
            one(i)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        out["contexts_%d" % nctx if mode else "contexts_1_batch_kernels"] = {"frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3}
        for c in ctxs:
            c.close()
    out["note"] = "one extract + match call per frame; contexts_1 = back-to-back latency, contexts_2 / _4 = calls of different contexts overlap"
    return out


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU"""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=5, help="timed blocks of --steps steps each; the line reports the median block (value_min / value_max: the others)")
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage hipEvents")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-fed pipeline figure and the batch-size sweep (N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank control flow on a 1-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--workload", default="orb32", choices=["orb32", "akaze61", "pairs10k", "l2sift128"],
                    help="orb32 = the BASELINE.json metric (default); akaze61 = configs[4], 1280x720, single GPU (use --batch 64); "
                         "pairs10k = configs[3], 10 000 keyframe-pair match jobs over a K = 1000 table, RCCL broadcast timed separately")
    ap.add_argument("--match-engine", type=int, default=None, choices=[0, 1], help="phase 1 of the pair matcher: 1 = matrix cores (library "
                    "default), 0 = popcount on the vector ALU (A/B measurement; identical results)")
    ap.add_argument("--lib", default=None, help="measurement tooling: bind this build of libafv_hip.so (tools/experiments.py variants)")
    ap.add_argument("--split-chunks", type=int, default=0, help="orb32: chunks a batch is split into over the two streams (0 = automatic)")
    ap.add_argument("--no-split", action="store_true", help="orb32: one stream, one chunk (per-kernel timelines)")
    ap.add_argument("--match-resolve", type=int, default=None, choices=[0, 1, 2], help="phase 2 of the pair matcher: 0 = one-wavefront walk, 1 = workgroup-wide "
                    "fixed point, 2 = by call size (library default)")
    ap.add_argument("--small-path", type=int, default=None, choices=[0, 1, 2], help="orb32: small-batch kernels 0 = never, 1 = calls of <= 4 "
                    "frames / pairs (library default), 2 = always (afv_set_small_batch_path)")
    ap.add_argument("--keyframes", type=int, default=1000, help="pairs10k: keyframes in the table")
    ap.add_argument("--jobs", type=int, default=10000, help="pairs10k: pair jobs per step (whole job, all GPUs)")
    ap.add_argument("--bcast-reps", type=int, default=3, help="pairs10k: repetitions of the table broadcast")
    args = ap.parse_args()
    if args.lib:
        importlib.import_module("anyfeature-vslam_amd._lib").use_library(args.lib)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.workload != "akaze61":
        _self_launch(args)
    # RCCL writes its banner / warnings to the C-level stdout whenever it initialises; the driver reads ONE JSON line from stdout.
    # So fd 1 is pointed at stderr for the whole run and the JSON line goes out through a private copy of the original stdout.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.workload == "akaze61":
        return akaze_main(args)
    if args.workload == "pairs10k":
        return pairs_main(args)
    if args.workload == "l2sift128":   # configs[2] alone (profiling runs: profiles/rNN/kernel_stats_l2.csv)
        afv = importlib.import_module("anyfeature-vslam_amd")
        out = {"metric": "SIFT128 L2 descriptor pairs /sec", "config": {"workload": "configs[2]: float descriptors 1000 x 1000 x 128, device table of 64 sets, 2048 pair jobs"}}
        out.update(extra_l2_sift128(afv, 0))
        out["value"] = out["pairs_device"]["descriptor_pairs_per_s"]
        out["unit"] = "descriptor pairs/s"
        print(json.dumps(out), file=_REAL_STDOUT, flush=True)
        return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the two scalar reductions live

    afv = importlib.import_module("anyfeature-vslam_amd")
    B = args.batch
    ctx = afv.Context(nfeatures=1000, nlevels=8, scale_factor=1.2, fast_threshold=20, max_width=W, max_height=H, max_batch=B,
                      device=local)
    if args.match_engine is not None:
        ctx.set_match_engine(args.match_engine)
    if args.split_chunks:
        ctx.set_split_chunks(args.split_chunks)
    if args.no_split:
        ctx.set_split_threshold(0x7fffffff)
    if args.small_path is not None:
        ctx.set_small_batch_path(args.small_path)
    if args.match_resolve is not None:
        ctx.set_match_resolve(args.match_resolve)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    matcher = afv.FeatureMatcher(0.6, True, ctx=ctx)

    # synthetic frames, resident in HBM before the timed region; seed = 1 + global frame index
    seed0 = 1 + rank * B
    frames = torch.from_numpy(afv.synth.corners_batch(seed0, B, W, H)).to(dev)
    cap = ctx.cap
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    n_out = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    match = torch.empty((B, cap), dtype=torch.int32, device=dev)
    nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
    pair_a = torch.arange(B, dtype=torch.int32, device=dev)
    pair_b = (pair_a + (B - 1)) % B  # t-1, frame 0 pairs with frame B-1

    side = torch.cuda.Stream(dev)    # a real stream: the library enqueues on it (the default stream would be event-bridged)

    def step():
        with torch.cuda.stream(side):
            ctx.extract_batch_device(frames, kps, desc, n_out, status, cap)
            matcher.match_pairs_device(desc, kps, n_out, pair_a, pair_b, th_low=75.0, check_orientation=True, match=match,
                                       nmatches=nmatch)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    assert int(status.item()) == 0
    # the library's stage events (roofline.achieved, stage_ms_per_step) are recorded on every PROF_EVERY-th step of the timed region:
    # an event pair per stage and chunk costs about 3 % of a step when every step carries them
    prof_steps = (args.steps + PROF_EVERY - 1) // PROF_EVERY
    if not args.no_profile:
        ctx.profile_enable(True, every=PROF_EVERY)
    # The timed region, `repeat` times over: every block is EXACTLY `steps` steps between two barriers (the contract's bracket), the
    # reported value is the MEDIAN block's (one 51 ms window moves by a few per cent with a clock-ramp hiccup), the slowest and
    # fastest block are printed next to it.  The stage events of the first block only feed stage_ms / the two-stream roofline note.
    block_dt = []
    stages = None
    for blk in range(max(args.repeat, 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        block_dt.append(time.perf_counter() - t0)
        if blk == 0 and not args.no_profile:
            stages = ctx.profile_read()
            ctx.profile_enable(False)

    kp_step = int(n_out.sum().item())
    nm_step = int(nmatch.sum().item())
    t = torch.tensor(block_dt, dtype=torch.float64, device=red_dev)
    k = torch.tensor([kp_step], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # per block: the slowest rank
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
    block_dt = sorted(float(v) for v in t.tolist())
    dt = block_dt[(len(block_dt) - 1) // 2]        # median block (the lower one of an even count)
    total_kp = int(k.item()) * args.steps
    # who did what (the first multi-GPU run should be boring): every rank's frame seeds and keypoint count
    per_rank = [{"rank": rank, "first_seed": seed0, "frames": B, "keypoints_per_step": kp_step}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    # collective, every rank: who ran where, and config #4 as the STRONG-scaling companion of this weak-scaling line (10 000 jobs in all,
    # table broadcast from rank 0 through the C-ABI communicator) - at N = 1 the same code path gives the single-GPU figure
    ident = multi_gpu_identity(rank, world, local, dev, dist, args.backend, None, args.single_device)
    strong = None
    if world > 1 and not args.no_extras:   # (N = 1: after the other sub-runs - initialising RCCL ahead of the host-fed pipeline figure
        try:                               # costs that figure a third of its PCIe rate on this box: 4.0 -> 6.0 ms per 512 frames, measured)
            strong = pairs_run(args, rank, world, local, dev, dist, 5, 2, False, False)
        except Exception as e:
            strong = {"error": repr(e)[:300]}
    if rank == 0:
        out = {
            "metric": METRIC, "value": total_kp / dt, "unit": "keypoints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "timed_blocks": len(block_dt), "value_min": total_kp / block_dt[-1], "value_max": total_kp / block_dt[0],
            "value_is": "median of %d timed blocks of %d steps each (every block bracketed by barrier + synchronize)" % (len(block_dt), args.steps),
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ORB32 640x480 synthetic corners frames (LCG), 1000 kp/frame budget, 8 levels x1.2, FAST 20; "
                                   "extract+describe on device, brute-force Hamming match frame t vs t-1 (TH 75, ratio 0.6, "
                                   "orientation check)", "frames_per_gpu_per_step": B, "global_frames_per_step": B * world,
                       "keypoints_per_frame": kp_step / B, "matches_per_frame": nm_step / B, "parallelism": "frames sharded x%d" % world,
                       "per_rank": per_rank, "backend": args.backend if world > 1 else None},
            "keypoints_per_ms": total_kp / dt / 1e3,
            "frames_per_s": B * world * args.steps / dt,
            "csrc_sha": _csrc_sha(),
            "multi_gpu": ident,
        }
        if stages:
            px = level_pixels(W, H)
            fh = stages["fast_nms"]
            if fh["launches"]:
                # The dominant kernel, measured so that rocprof sees the same thing: ONE launch over the whole batch on
                # ONE stream with the chip otherwise idle, HIP events on the launch stream (standalone_fast_nms) - the launch shape of
                # profiles/rNN/kernel_stats_single_stream.csv (bench.py --no-split under rocprofv3 --kernel-trace --stats).  The figures of
                # the timed region itself - chunks of about 85 frames alternating over two streams, so an event pair around a launch also
                # spans the other stream's kernels - stay under `timed_region_two_streams`.
                ms2 = fh["total_ms"] / fh["launches"]
                fpl2 = fh["units"] / fh["launches"]
                ach2 = px * fpl2 / (ms2 * 1e-3) / 1e9
                two = {"avg_launch_ms_spanning_both_streams": ms2, "frames_per_launch": fpl2, "achieved": ach2, "frac": ach2 / HBM_PEAK_GBS,
                       "note": "hipEvent pairs around the launches of the timed region; a launch shares the chip with the other stream's "
                               "kernels, so this is an upper bound of the kernel's own duration, not a kernel duration"}
                traffic, tpath, tstale = pmc_traffic("k_fast_nms", B)
                sa = None
                if world == 1:
                    try:
                        sa = standalone_fast_nms(afv, local, B)
                    except Exception as e:
                        sa = None
                        two["standalone_error"] = str(e)[:200]
                ms, fpl, ach = (sa["avg_launch_ms"], sa["frames_per_launch"], sa["achieved"]) if sa else (ms2, fpl2, ach2)
                out["roofline"] = {"bound": "hbm", "kernel": "k_fast_nms", "achieved": ach, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": None if (traffic is None or tstale) else traffic * fpl,
                                   "traffic_source": tpath, "traffic_stale": bool(tstale),
                                   "algorithmic_bytes_per_launch": px * fpl, "avg_launch_ms": ms,
                                   "frames_per_launch": fpl,
                                   "measured": ("one launch over the whole batch, one stream, chip otherwise idle; HIP events on the launch stream"
                                                if sa else "timed region (two streams): see timed_region_two_streams"),
                                   "timed_region_two_streams": two,
                                   "note": "integer-VALU-bound kernel (FAST ring tests, exact scores, NMS): the HBM fraction is low by "
                                           "construction (DESIGN.md section 4); traffic = committed PMC pass x frames_per_launch, null when "
                                           "the sources changed since (traffic_stale)"}
                ks = _newest_profile("kernel_stats_single_stream.json")
                if ks and "k_fast_nms" in ks.get("kernels", {}):   # the committed rocprofv3 summary of the same launch shape
                    e = ks["kernels"]["k_fast_nms"]
                    pf = px * ks.get("frames_per_launch", fpl) / (e["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS
                    out["roofline"]["rocprof"] = {"avg_launch_ms": e["avg_ns"] / 1e6, "frames_per_launch": ks.get("frames_per_launch"), "frac": pf,
                                                  "source": ks["_path"], "stale": bool(ks["_stale"])}
            # whole-pipeline algorithmic bytes (resize 1 569 878 + FAST read 950 532 + blur 1 901 064 + outputs
            # 60 000 = 4 481 534 B per 640x480 frame; the blur never touches HBM here, the figure is the reference's data flow)
            out["roofline_pipeline"] = {"bound": "hbm", "achieved": out["frames_per_s"] * 4481534 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": out["frames_per_s"] * 4481534 / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_frame": 4481534}
            vp = valu_pmc()
            if vp:
                stale = bool(vp["_stale"])
                ach = vp["valu_winst_per_frame_total"] * out["frames_per_s"]
                classes = {}
                try:
                    classes = json.load(open(os.path.join(ROOT, os.path.dirname(vp["_path"]), "calib_valu.json")))["classes_winst_per_s"]
                except Exception:
                    pass
                pk = classes.get("v_pk_min_i16+add")
                # ONE blended peak: static op-class shares of every kernel (tools/isa_valu_classes.py, from the ISA) weighted with
                # the kernels' dynamic instruction counts (PMC), every class at its measured issue rate
                isa = _newest_profile("isa_valu_classes.json")
                mix_peak = None
                if isa and not isa["_stale"] and classes.get("v_add_u32"):
                    plain = classes["v_add_u32"]
                    rate = {"plain": plain, "pk_i16": pk or plain, "pk_mad": classes.get("v_pk_mad_u16", plain),
                            "dot": classes.get("v_dot4_u32_u8", plain)}
                    xb = classes.get("xor+bcnt")
                    rate["bcnt"] = 1.0 / (2.0 / xb - 1.0 / plain) if xb else plain  # the calibration loop alternates xor and bcnt
                    n_tot, t_tot = 0.0, 0.0
                    for k, v in vp["kernels"].items():
                        sh = isa["kernels"].get(k, {}).get("shares")
                        if not sh:
                            continue
                        n = v["valu_winst_per_frame"]
                        n_tot += n
                        t_tot += n * sum(share / rate.get(c, plain) for c, share in sh.items())
                    mix_peak = n_tot / t_tot if t_tot > 0 else None
                out["valu_issue"] = {"achieved": None if stale else ach, "peak": vp["valu_peak_winst_per_s"], "unit": "wave-instr/s",
                                     "frac": None if stale else ach / vp["valu_peak_winst_per_s"], "stale": stale, "source": vp["_path"],
                                     "peak_by_op_class": classes,
                                     "frac_vs_packed_i16_class": None if (stale or not pk) else ach / pk,
                                     "peak_isa_mix": mix_peak, "frac_vs_isa_mix": None if (stale or not mix_peak) else ach / mix_peak,
                                     "kernels_winst_per_frame": {k: v["valu_winst_per_frame"] for k, v in vp["kernels"].items()},
                                     "note": "whole pipeline: SQ_INSTS_VALU per frame (committed PMC pass of tools/collect_profiles.sh, stamped with the "
                                             "sha of the sources it ran on) x live frames/s vs the integer-VALU issue rates measured by tools/calib_valu.hip "
                                             "on the same box: peak = the fastest op class (plain 32-bit adds); the packed-i16 / dot4 / popcount ops that "
                                             "make up most of this pipeline issue at the packed class rate; peak_isa_mix blends the class rates with every kernel's static "
                                             "op-class shares (profiles/r*/isa_valu_classes.json) and dynamic counts - uncalibrated classes at the plain rate, so "
                                             "frac_vs_isa_mix is a lower bound.  stale = the sources changed since the pass, "
                                             "the fractions are withheld"}
            cp = _newest_profile("cache_pmc.json")
            if cp:   # north_star: "L2/LDS hit rate on the BRIEF + Hamming pass"; same staleness rule as valu_issue / roofline.traffic
                keep = ("k_describe", "k_match_topk_mfma", "k_match_resolve", "k_fast_nms", "k_harris", "k_resize_level", "k_select_quadtree")
                out["cache"] = {"stale": cp["_stale"], "source": cp["_path"],
                                "kernels": None if cp["_stale"] else {k: {f: (round(v, 4) if isinstance(v, float) else v) for f, v in e.items() if f in
                                                                           ("l2_hit", "lds_conflict_cycles_per_inst", "lds_wait_cycles_per_inst")}
                                                                       for k, e in cp["kernels"].items() if k.split("<")[0] in keep},
                                "note": "l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS), LDS bank-conflict and wait cycles per LDS instruction: committed PMC passes "
                                        "(tools/collect_profiles.sh -> tools/pmc_cache.py), withheld when the sources changed since"}
            sq = _newest_profile("sq_activity.json")
            if sq:   # why the HBM fraction of the dominant kernel is low: its vector issue slots are full (SQ activity counters, same staleness rule)
                keys = ("valu_busy_frac", "lds_array_busy_frac", "scalar_busy_frac", "resident_waves_per_simd", "sclk_GHz_while_running",
                        "wave_cycles_issuing_frac", "wave_cycles_waiting_on_waitcnt_frac", "wave_cycles_waiting_to_issue_frac")
                out["issue_activity"] = {"stale": sq["_stale"], "source": sq["_path"],
                                         "kernels": None if sq["_stale"] else {k: {f: round(v[f], 4) for f in keys if isinstance(v.get(f), (int, float))}
                                                                               for k, v in sq.items() if isinstance(v, dict) and "valu_busy_frac" in v},
                                         "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES (one vector issue slot per SIMD and 4 cycles; 1.0 = never idle, "
                                                 "above 1.0 = co-issued instructions): committed PMC passes of tools/collect_profiles.sh, withheld when the sources changed since"}
            # event time of each stage summed over BOTH streams of a step: the two streams run concurrently, so an entry (and their sum)
            # may exceed ms_per_step - these are not kernel durations (rocprof: profiles/rNN/kernel_stats_default.csv)
            out["stage_event_ms_per_step_summed_over_concurrent_streams"] = {kk: (v["total_ms"] / prof_steps) for kk, v in stages.items()}
            out["stage_events_on_steps"] = "%d of %d timed steps (every %d-th)" % (prof_steps, args.steps, PROF_EVERY)
        if world == 1 and not args.no_extras:
            # extraction alone, device-resident (the yardstick of the host-fed pipeline)
            barrier()
            t1 = time.perf_counter()
            for _ in range(max(args.steps // 2, 2)):
                with torch.cuda.stream(side):
                    ctx.extract_batch_device(frames, kps, desc, n_out, status, cap)
            barrier()
            dev_fps = B * max(args.steps // 2, 2) / (time.perf_counter() - t1)
            out["host_fed"] = host_fed(afv, ctx, frames.cpu().numpy(), max(args.steps // 3, 5), dev_fps)
            out["batch_sweep"] = batch_sweep(afv, local)
            out["overlap_match"] = overlap_step(afv, local)
            try:
                out["host_api"] = extra_host_api()
                tk = tracking_keys(out["host_api"])
                if tk:   # the per-frame tracking calls as keys of the driver's line
                    out.update(tk)
            except Exception as e:
                out["host_api"] = {"error": str(e)[:200]}
            for key, fn in (("single_frame", extra_single_frame), ("pairs10k", extra_pairs10k), ("l2_sift128", extra_l2_sift128), ("akaze61", extra_akaze61),
                            ("tracking_frame_akaze61", extra_tracking_frame_akaze61), ("tracking_frame_float128", extra_tracking_frame_float128)):
                try:
                    out[key] = fn(afv, local)
                except Exception as e:  # a secondary figure must never cost the headline line
                    out[key] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_extras:
            try:
                strong = pairs_run(args, rank, world, local, dev, dist, 5, 2, False, False)
            except Exception as e:
                strong = {"error": repr(e)[:300]}
        if strong:
            out["pairs10k_strong"] = strong if "error" in strong else {
                "metric": strong["metric"], "value": strong["value"], "unit": strong["unit"], "n_gpus": strong["n_gpus"], "scaling": "strong",
                "ms_per_step": strong["ms_per_step"], "jobs_per_step": strong["config"]["jobs_per_step"], "job_ranges": strong["config"]["job_ranges"],
                "matches_per_job": strong["config"]["matches_per_job"], "broadcast": strong["broadcast"], "covisible": strong["covisible"],
                "rccl_ranks_seen": strong["multi_gpu"]["rccl_ranks_seen"],
                "note": "BASELINE.json configs[3] on the same ranks: total work fixed, table replicated with one broadcast from rank 0"}
        if args.cpu_frames > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(afv, args.cpu_frames, seed0)
        elif args.cpu_frames > 0:
            out["cpu_baseline"] = None
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
