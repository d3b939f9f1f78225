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
  roofline      dominant kernel (k_fast_harris): algorithmic bytes (every pyramid pixel read once = 950 532 B/frame at
                640x480) x frames per launch / mean launch duration measured with hipEvents recorded on
                the launch stream inside the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference cannot be built here) timed single-threaded on
                rank 0 on a bounded sample of the same frames, reference-faithful call pattern (variant 1).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "keypoints extracted+matched /sec (ORB32, 640x480)"
HBM_PEAK_GBS = 8000.0
W, H = 640, 480


def level_pixels(w, h, nlevels=8, scale=1.2):
    tot = 0
    for l in range(nlevels):
        s = np.float32(np.power(np.float64(np.float32(scale)), l))
        inv = np.float32(1.0) / s
        tot += int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv))
    return tot


def pmc_traffic(kernel, batch):
    """HBM bytes per FRAME of `kernel` from the committed rocprofv3 PMC passes (profiles/r*/traffic_pmc.json: FETCH_SIZE
    and WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes); None if no pass was
    taken at this batch size.  PMC counters cannot be read from inside a normal run."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_pmc.json")), reverse=True):
        try:
            d = json.load(open(p))
            if kernel in d["kernels"]:  # per-frame figure: independent of the batch size
                return d["kernels"][kernel]["hbm_bytes_per_frame"]
        except Exception:
            pass
    return None


def valu_pmc():
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "valu_pmc.json")), reverse=True):
        try:
            return json.load(open(p))
        except Exception:
            pass
    return None


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
    frames = [afv.synth.corners_frame(seed0 + i) for i in range(nframes)]
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

    half = max(nframes // 2, 1)
    va, ta = run(1, frames[:half])
    vb, tb = run(0, frames[half:] or frames[:half])
    return {"value": va, "unit": "keypoints/s", "cores": 1, "kind": "port", "dedup_value": vb,
            "sample": "variant A (reference call pattern): %d frames 640x480 corners in %.1f s; variant B (de-duplicated): %d frames "
                      "in %.1f s; extraction + brute-force match vs previous frame; 1 thread pinned; host %s, %d logical cores"
                      % (half, ta, len(frames[half:] or frames[:half]), tb, cpu, os.cpu_count() or 0),
            "ms_per_frame": 1e3 * ta / half, "dedup_ms_per_frame": 1e3 * tb / max(len(frames[half:] or frames[:half]), 1)}


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
    plan = ctx.plan
    px0 = Wa * Ha
    # Algorithmic HBM bytes per frame of scale space + Hessian, each datum moved once: level 0 reads the u8 frame twice (Gaussian
    # and contrast percentile) and writes Lt; every further level reads the previous Lt, writes Lsmooth and Lt; the Hessian reads
    # Lsmooth and writes Lx, Ly, Ldet.  The kernel structure moves more ("kernel_structure": gauss, level kernel, 2 derivative kernels).
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
        strict += 16 * n
        kern += 32 * n
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
    if args.cpu_frames > 0:
        out["cpu_baseline"] = akaze_cpu_baseline(afv, frames_h[:min(args.cpu_frames, B, 4)], ctx.quotas())
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage hipEvents")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank control flow on a 1-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--workload", default="orb32", choices=["orb32", "akaze61"],
                    help="orb32 = the BASELINE.json metric (default); akaze61 = configs[4], 1280x720, single GPU (use --batch 64)")
    args = ap.parse_args()
    if args.workload == "akaze61":
        return akaze_main(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
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

    def step():
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
    if not args.no_profile:
        ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stages = None if args.no_profile else ctx.profile_read()
    if not args.no_profile:
        ctx.profile_enable(False)

    kp_step = int(n_out.sum().item())
    nm_step = int(nmatch.sum().item())
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    k = torch.tensor([kp_step], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_kp = int(k.item()) * args.steps

    if rank == 0:
        out = {
            "metric": METRIC, "value": total_kp / dt, "unit": "keypoints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ORB32 640x480 synthetic corners frames (LCG), 1000 kp/frame budget, 8 levels x1.2, FAST 20; "
                                   "extract+describe on device, brute-force Hamming match frame t vs t-1 (TH 75, ratio 0.6, "
                                   "orientation check)", "frames_per_gpu_per_step": B, "global_frames_per_step": B * world,
                       "keypoints_per_frame": kp_step / B, "matches_per_frame": nm_step / B, "parallelism": "frames sharded x%d" % world},
            "keypoints_per_ms": total_kp / dt / 1e3,
            "frames_per_s": B * world * args.steps / dt,
        }
        if stages:
            px = level_pixels(W, H)
            fh = stages["fast_harris"]
            if fh["launches"]:
                ms = fh["total_ms"] / fh["launches"]                 # mean launch duration (hipEvents on the launch stream)
                frames_per_launch = fh["units"] / fh["launches"]     # the runtime splits a batch over two streams
                achieved = px * frames_per_launch / (ms * 1e-3) / 1e9
                traffic = pmc_traffic("k_fast_harris", B)
                out["roofline"] = {"bound": "hbm", "kernel": "k_fast_harris", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                   "traffic": None if traffic is None else traffic * frames_per_launch,
                                   "algorithmic_bytes_per_launch": px * frames_per_launch, "avg_launch_ms": ms,
                                   "frames_per_launch": frames_per_launch,
                                   "note": "integer-VALU-bound kernel (FAST ring tests + Harris): the HBM fraction is low by "
                                           "construction; the runtime runs a step as four quarter-batch launches per kernel over two "
                                           "streams, so a launch shares the chip with the other stream's kernels (DESIGN.md section 4)"}
            # whole-pipeline algorithmic bytes (resize 1 569 878 + FAST read 950 532 + blur 1 901 064 + outputs
            # 60 000 = 4 481 534 B per 640x480 frame; the blur never touches HBM here, the figure is the reference's data flow)
            out["roofline_pipeline"] = {"bound": "hbm", "achieved": out["frames_per_s"] * 4481534 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": out["frames_per_s"] * 4481534 / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_frame": 4481534}
            vp = valu_pmc()
            if vp:
                ach = vp["valu_winst_per_frame_total"] * out["frames_per_s"]
                out["valu_issue"] = {"achieved": ach, "peak": vp["valu_peak_winst_per_s"], "unit": "wave-instr/s",
                                     "frac": ach / vp["valu_peak_winst_per_s"],
                                     "note": "whole pipeline: SQ_INSTS_VALU per frame (committed PMC pass, profiles/r*/valu_pmc.json) x measured "
                                             "frames/s vs the measured integer-VALU issue peak (tools/calib_valu.hip) - the bound that actually "
                                             "limits this integer/byte path"}
            out["stage_ms_per_step"] = {kk: (v["total_ms"] / args.steps) for kk, v in stages.items()}  # overlapping streams: sums exceed ms_per_step
        if args.cpu_frames > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(afv, args.cpu_frames, seed0)
        elif args.cpu_frames > 0:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
