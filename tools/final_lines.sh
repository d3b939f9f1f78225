#!/bin/bash
# Regenerates the bench lines committed under profiles/<tag>/ AFTER the PMC passes of collect_profiles.sh were copied there (so that
# roofline.traffic / valu_issue are live, not stale), plus the N = 2 rehearsal of the distributed path on one device and the
# per-kernel busy / overlap breakdown.  Run on the GPU box: bash tools/final_lines.sh r02
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p "$OUT"
cd "$ROOT"
# same GPU call as collect_profiles.sh: put its PMC / calibration files where bench.py looks for them (profiles/<tag>/)
if [ -d "$ROOT/gpurun_out/profiles_$TAG" ]; then mkdir -p "$ROOT/profiles/$TAG" && cp "$ROOT"/gpurun_out/profiles_$TAG/*.json "$ROOT/profiles/$TAG/" 2>/dev/null; fi
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --workload pairs10k 2> "$OUT/bench_pairs10k.err" | grep "^{" > "$OUT/bench_pairs10k.json"
python bench.py --workload akaze61 --batch 64 --steps 5 2> "$OUT/bench_akaze61.err" | grep "^{" > "$OUT/bench_akaze61.json"
python tools/timeline.py --no-split > "$OUT/timeline_single_stream.json" 2>/dev/null
python tools/timeline.py > "$OUT/timeline_default.json" 2>/dev/null
for W in orb32 pairs10k; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 \
      --backend gloo --single-device --workload $W --cpu-frames 0 > "$OUT/rehearsal_gloo2_$W.json" 2> "$OUT/rehearsal_gloo2_$W.err"
done
python tools/timeline.py --batch 1 --no-split > "$OUT/timeline_single_frame.json" 2>/dev/null
python tools/bench_single_frame.py > "$OUT/single_frame.txt" 2>/dev/null
# the per-frame plugin call (round 4): kernel-level trace of bench.py's single_frame.contexts_1 workload with the small-batch kernels (1) and
# with the batch kernels (0), and the host-to-host latencies measured from C++
for M in 1 0; do python tools/single_frame_trace.py $M > /dev/null 2>&1; cp gpurun_out/single_frame_trace/summary_mode$M.json "$OUT/single_frame_trace_mode$M.json" 2>/dev/null; done
./tools/host_latency 300 > "$OUT/host_latency.json" 2>&1
AFV_TRACE_HOST=1 ./tools/host_latency 300 2>&1 | grep "afv_orb_extract (us" | tail -2 > "$OUT/host_latency_breakdown.txt"
python tools/probes/probe_pcie.py > "$OUT/pcie_probe.txt" 2>&1
for P in probe_cvt_pk_u8 probe_mfma_valu probe_stream13 probe_u8_tiles; do
  hipcc --offload-arch=gfx950 -O3 tools/probes/$P.hip -o /tmp/$P 2>/dev/null && /tmp/$P > "$OUT/$P.txt" 2>&1
done
./tools/akaze_recip_check > "$OUT/akaze_recip_check.txt" 2>&1
python tools/overlap_trace.py 2 > "$OUT/overlap_trace.txt" 2>/dev/null
python tools/probes/probe_resolve_pair.py 2>/dev/null | grep "^pairs" > "$OUT/resolve_pairs.txt"
# round 6: the instruction fault's reproducer, the stress scenes and the poison differential on the final sources
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/probes/probe_pk_real.hip -Iinclude -Lanyfeature-vslam_amd -lafv_hip -Wl,-rpath,$ROOT/anyfeature-vslam_amd \
    -lpthread -o /tmp/probe_pk_real 2>/dev/null && /tmp/probe_pk_real 4 single > "$OUT/probe_pk_real_single_instructions.txt" 2>&1
/tmp/probe_pk_real 4 2>&1 | head -4 > "$OUT/probe_pk_real_forms.txt"
python tools/stress_threads.py --rounds 400 --roles emm 2>&1 | tail -1 > "$OUT/stress_threads.txt"
python tools/stress_threads.py --rounds 400 --roles mmm 2>&1 | tail -1 >> "$OUT/stress_threads.txt"
python tools/stress_threads.py --rounds 400 2>&1 | tail -1 >> "$OUT/stress_threads.txt"
python tools/poison_build.py > /dev/null 2>&1; bash tools/poison_suite.sh 60 > "$OUT/poison_suite.txt" 2>&1
ls -la "$OUT"
