"""CPU, world_size 2, gloo: the N > 1 path — frame sharding and the config-#4 exchange (broadcast of the keyframe
descriptor table, sharded pair jobs, gather of the per-job results).  The device matcher is injected; here the CPU
oracle stands in for it so the distributed logic can be checked against a single-process run."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_match_fn(table, counts, pa, pb):
    import oracle
    out = []
    t = table.numpy(); c = counts.numpy()
    for a, b in zip(pa.tolist(), pb.tolist()):
        _, n = oracle.search_by_bow_kf_kf(t[a, :c[a]], t[b, :c[b]], th_low=75.0, nnratio=0.6)
        out.append(n)
    return torch.tensor(out, dtype=torch.int32)


def _make_table(synth, K=12, cap=64):
    table = np.zeros((K, cap, 32), np.uint8)
    counts = np.zeros(K, np.int32)
    d = synth.random_descriptors(1, cap)
    for k in range(K):
        n = cap - (k % 5)
        table[k, :n] = d[:n]
        counts[k] = n
        d = synth.perturbed_descriptors(d, 100 + k)
    return table, counts


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("anyfeature-vslam_amd.dist")
    synth = importlib.import_module("anyfeature-vslam_amd.synth")
    K, cap = 12, 64
    if rank == 0:
        t, c = _make_table(synth, K, cap)
        table, counts = torch.from_numpy(t), torch.from_numpy(c)
    else:  # other ranks start with garbage: the broadcast must overwrite it
        table = torch.full((K, cap, 32), 7, dtype=torch.uint8)
        counts = torch.zeros(K, dtype=torch.int32)
    a, b = d.lcg_pairs(5, 37, K)
    res = d.match_jobs_sharded(table, counts, torch.from_numpy(a), torch.from_numpy(b), _oracle_match_fn, src=0)
    lo, hi = d.shard_range(37, rank, world)
    # frame sharding of a batch: every frame index is owned by exactly one rank
    owned = torch.zeros(100, dtype=torch.int32)
    flo, fhi = d.shard_range(100, rank, world)
    owned[flo:fhi] = 1
    dist.all_reduce(owned)
    q.put((rank, res.tolist(), table.numpy().tobytes()[:64], (lo, hi), owned.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_config4_broadcast_and_sharded_jobs_world2():
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("anyfeature-vslam_amd.synth")
    d = importlib.import_module("anyfeature-vslam_amd.dist")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    t, c = _make_table(synth)
    a, b = d.lcg_pairs(5, 37, 12)
    want = _oracle_match_fn(torch.from_numpy(t), torch.from_numpy(c), torch.from_numpy(a), torch.from_numpy(b)).tolist()
    results.sort()
    assert results[0][1] == want and results[1][1] == want          # every rank holds the full, ordered result
    assert results[0][2] == results[1][2] == t.tobytes()[:64]       # the table really was replicated
    assert results[0][3][1] == results[1][3][0] and results[1][3][1] == 37
    assert results[0][4] == [1] * 100                               # frames: exactly one owner each
    assert sum(want) > 0
