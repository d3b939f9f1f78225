#!/bin/bash
# One command that regenerates the profile evidence bench.py and DESIGN.md quote.  Run it ON the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r03'
# then copy gpurun_out/profiles_r03/* into profiles/r03/ and commit.  Counter passes are separate rocprofv3 runs with
# --kernel-trace only (no other trace domain), as gpurun requires.
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
RAW=$ROOT/gpurun_out/profiles_${TAG}_raw     # raw counter CSVs stay in scratch: only the summaries derived from them are copied to profiles/
mkdir -p "$OUT" "$RAW"
cd /tmp && export TMPDIR=/tmp
PY=python
B=256; STEPS=3; WARM=1
BENCHARGS="--batch $B --steps $STEPS --warmup $WARM --repeat 1 --cpu-frames 0 --no-profile --no-extras"   # ONE timed block: the per-frame figures divide by B x (STEPS + WARM)

echo "== calibration: integer VALU issue peak (tools/calib_valu.hip)"
hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/calib_valu.hip" -o /tmp/calib_valu && /tmp/calib_valu | tee "$OUT/calib_valu.txt"
$PY - "$OUT" <<'PYEOF'
import json, re, sys
out = sys.argv[1]
rates = {}
for line in open(out + "/calib_valu.txt"):
    m = re.match(r"(\S+)\s+([\d.]+) ms\s+([\d.e+]+) wave-instr/s", line)
    if m:
        rates[m.group(1)] = float(m.group(3))
json.dump({"classes_winst_per_s": rates, "valu_peak_winst_per_s": max(rates.values()) if rates else None,
           "note": "tools/calib_valu.hip: 8 waves/SIMD, ITER x 32 independent ops on 8 accumulators per lane; peak = best class"},
          open(out + "/calib_valu.json", "w"), indent=1)
PYEOF

echo "== calibration: FETCH_SIZE / WRITE_SIZE against known byte counts (tools/calib_fetch.hip)"
hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/calib_fetch.hip" -o /tmp/calib_fetch
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/calib_$C" -o c --output-format csv -- /tmp/calib_fetch > /dev/null 2>&1
done
$PY - "$OUT" <<'PYEOF'
import csv, glob, json, sys
out = sys.argv[1]
BYTES = 1024 << 20
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = {}
    for p in glob.glob(out + "/calib_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == C and k.startswith("k_"):
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    res[C] = {k: sum(v) / len(v) * 1024 / BYTES for k, v in acc.items()}
json.dump({"ratio_counter_x1024_over_bytes": res, "fetch_ratio_4B": res["FETCH_SIZE"].get("k_read4"), "fetch_ratio_16B": res["FETCH_SIZE"].get("k_read16"),
           "write_ratio_4B": res["WRITE_SIZE"].get("k_write4"),
           "note": "1 GiB coalesced streams (larger than the Infinity Cache); ratio = counter * 1024 / bytes really moved"},
          open(out + "/calib_fetch.json", "w"), indent=1)
print(json.dumps(res))
PYEOF

echo "== default bench (the BENCH line) + kernel stats of the same command"
cd "$ROOT" && $PY bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_default" -o st --output-format csv -- $PY "$ROOT/bench.py" --cpu-frames 0 --no-extras > /dev/null 2>&1
cp "$OUT"/stats_default/*kernel_stats.csv "$OUT/kernel_stats_default.csv" 2>/dev/null || cp "$OUT"/stats_default/*/*kernel_stats.csv "$OUT/kernel_stats_default.csv"

echo "== the dominant kernel in the launch shape bench.py's roofline quotes: one stream, one chunk (bench.py --no-split), rocprof beside the hipEvents"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_single" -o st --output-format csv -- $PY "$ROOT/bench.py" --no-split --cpu-frames 0 --no-extras --no-profile > /dev/null 2>&1
cp "$(find "$OUT/stats_single" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_single_stream.csv"
cd "$ROOT" && $PY tools/kernel_stats_json.py "$OUT/kernel_stats_single_stream.csv" "$OUT/kernel_stats_single_stream.json" 512; cd /tmp
rm -rf "$OUT/stats_single"

echo "== the tracking step of one frame (tools/host_latency: resident Frame, projection searches, BoW, initialization): kernel stats + counters"
"$ROOT/tools/host_latency" 300 > "$OUT/host_latency.json" 2> "$OUT/host_latency.err"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_tracking" -o st --output-format csv -- "$ROOT/tools/host_latency" 100 > /dev/null 2>&1
cp "$(find "$OUT/stats_tracking" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_tracking.csv"
cd "$ROOT" && $PY tools/kernel_stats_json.py "$OUT/kernel_stats_tracking.csv" "$OUT/kernel_stats_tracking.json" - k_proj,k_init,k_match_fuse,k_bow,k_frame,k_featvec,k_table_promote,k_match_bow; cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CU_CYCLES -d "$OUT/pmc_tracking" -o p --output-format csv -- "$ROOT/tools/host_latency" 20 > /dev/null 2>&1
cp "$(find "$OUT/pmc_tracking" -name "*counter_collection.csv" | head -1)" "$RAW/pmc_tracking_counter_collection.csv" 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tracking_l2" -o p --output-format csv -- "$ROOT/tools/host_latency" 20 > /dev/null 2>&1
cp "$(find "$OUT/pmc_tracking_l2" -name "*counter_collection.csv" | head -1)" "$RAW/pmc_tracking_l2_counter_collection.csv" 2>/dev/null
$PY "$ROOT/tools/pmc_tracking.py" "$RAW/pmc_tracking_counter_collection.csv" "$RAW/pmc_tracking_l2_counter_collection.csv" "$OUT/pmc_tracking.json" > "$OUT/pmc_tracking.txt" 2>&1
rm -rf "$OUT/stats_tracking" "$OUT/pmc_tracking" "$OUT/pmc_tracking_l2"

echo "== PMC passes on: bench.py $BENCHARGS"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -o p --output-format csv -- $PY "$ROOT/bench.py" $BENCHARGS > /dev/null 2>&1
  F=$(find "$OUT/pmc_$C" -name "*counter_collection.csv" | head -1); cp "$F" "$RAW/pmc_${C}_counter_collection.csv"
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d "$OUT/pmc_LDS" -o p --output-format csv -- $PY "$ROOT/bench.py" $BENCHARGS > /dev/null 2>&1
cp "$(find "$OUT/pmc_LDS" -name "*counter_collection.csv" | head -1)" "$RAW/pmc_lds_counter_collection.csv"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_L2" -o p --output-format csv -- $PY "$ROOT/bench.py" $BENCHARGS > /dev/null 2>&1
cp "$(find "$OUT/pmc_L2" -name "*counter_collection.csv" | head -1)" "$RAW/pmc_l2_counter_collection.csv" 2>/dev/null
cd "$ROOT" && $PY tools/pmc_per_frame.py $B $((STEPS + WARM)) "$OUT" "$RAW/pmc_FETCH_SIZE_counter_collection.csv" "$RAW/pmc_WRITE_SIZE_counter_collection.csv" \
    "$RAW/pmc_SQ_INSTS_VALU_counter_collection.csv" "$OUT/calib_fetch.json" "$OUT/calib_valu.json" > "$OUT/pmc_per_frame.txt"
$PY tools/pmc_summary.py "$RAW/pmc_lds_counter_collection.csv" "$RAW/pmc_l2_counter_collection.csv" > "$OUT/pmc_lds_l2_summary.txt" 2>/dev/null
$PY tools/pmc_cache.py "$OUT" "$RAW/pmc_lds_counter_collection.csv" "$RAW/pmc_l2_counter_collection.csv" > "$OUT/pmc_cache.txt" 2>/dev/null
cd /tmp

echo "== what the two big kernels wait for: SQ activity / wait counters (three passes per kernel; tools/experiments.py prints per-frame sums)"
( cd "$ROOT" && for K in k_fast_nms k_describe; do
    AFV_EXP_KERNEL=$K AFV_EXP_PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $PY tools/experiments.py run base
    AFV_EXP_KERNEL=$K AFV_EXP_PMC="SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_WAVES" $PY tools/experiments.py run base
    AFV_EXP_KERNEL=$K AFV_EXP_PMC="SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_CYCLES" $PY tools/experiments.py run base
  done ) > "$OUT/sq_activity.jsonl" 2>/dev/null
$PY - "$OUT" <<'PYEOF'
import json, sys
out = sys.argv[1]
acc = {}
for line in open(out + "/sq_activity.jsonl"):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    k = acc.setdefault(d["kernel"], {"us_per_launch": []})
    k["us_per_launch"].append(d["us_per_launch"])
    k["frames_per_launch_assumed"] = d["frames_per_launch"]   # frames of the run / launches of the kernel (tools/experiments.py)
    for key, v in d.items():
        if key.endswith("_per_frame"):
            k[key[:-10]] = v
res = {}
for name, k in acc.items():
    us = sum(k["us_per_launch"]) / len(k["us_per_launch"])
    cu_cycles = k.get("SQ_BUSY_CU_CYCLES", 0.0) * k["frames_per_launch_assumed"] / 256.0   # per CU and launch
    res[name] = {
        "us_per_launch": us,
        "sclk_GHz_while_running": cu_cycles / us / 1e3 if us else None,
        "valu_busy_frac": 4.0 * k.get("SQ_ACTIVE_INST_VALU", 0.0) / (4.0 * k["SQ_BUSY_CU_CYCLES"]) if k.get("SQ_BUSY_CU_CYCLES") else None,
        "lds_array_busy_frac": k.get("SQ_LDS_IDX_ACTIVE", 0.0) / k["SQ_BUSY_CU_CYCLES"] if k.get("SQ_BUSY_CU_CYCLES") else None,
        "scalar_busy_frac": k.get("SQ_INST_CYCLES_SALU", 0.0) / k["SQ_BUSY_CU_CYCLES"] if k.get("SQ_BUSY_CU_CYCLES") else None,
        "wave_cycles_waiting_on_waitcnt_frac": k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") else None,
        "wave_cycles_waiting_to_issue_frac": k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") else None,
        "wave_cycles_issuing_frac": k.get("SQ_ACTIVE_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") else None,
        "resident_waves_per_simd": k.get("SQ_WAVE_CYCLES", 0.0) / k["SQ_BUSY_CU_CYCLES"] if k.get("SQ_BUSY_CU_CYCLES") else None,
        "counters_per_frame": {c: v for c, v in k.items() if c.startswith("SQ_")},
    }
res["note"] = ("SQ_ACTIVE_INST_VALU counts 4-cycle issue slots per SIMD (one per vector instruction): valu_busy_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES "
               "(4 SIMDs x 1 slot per 4 cycles = 1 per CU cycle) - 1.0 means the vector issue port is never idle; SQ_WAVE_CYCLES likewise in units of 4 cycles "
               "(resident_waves_per_simd = 8 is full occupancy).  sclk from SQ_BUSY_CU_CYCLES over the kernel's duration in the trace of the same pass.")
try:
    sys.path.insert(0, out + "/../../tools")
    import csrc_sha
    res["csrc_sha"] = csrc_sha.csrc_sha()
except Exception:
    pass
json.dump(res, open(out + "/sq_activity.json", "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters_per_frame"} for k, v in res.items() if isinstance(v, dict)}, indent=1))
PYEOF

echo "== static VALU op-class shares of the kernels (from the ISA; feeds valu_issue.peak_isa_mix)"
cd "$ROOT" && $PY tools/isa_valu_classes.py "$OUT/isa_valu_classes.json" > "$OUT/isa_valu_classes.txt" 2>&1; cd /tmp

echo "== config #4: pairs10k bench + kernel stats"
cd "$ROOT" && $PY bench.py --workload pairs10k > "$OUT/bench_pairs10k.json" 2> "$OUT/bench_pairs10k.err"; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_pairs" -o st --output-format csv -- $PY "$ROOT/bench.py" --workload pairs10k --steps 5 --cpu-frames 0 > /dev/null 2>&1
cp "$(find "$OUT/stats_pairs" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_pairs10k.csv"

timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/pmc_pairs" -o p --output-format csv -- $PY "$ROOT/bench.py" --workload pairs10k --steps 3 --cpu-frames 0 --no-profile > /dev/null 2>&1
cp "$(find "$OUT/pmc_pairs" -name "*counter_collection.csv" | head -1)" "$RAW/pmc_pairs10k_counter_collection.csv" 2>/dev/null
$PY "$ROOT/tools/pmc_summary.py" "$RAW/pmc_pairs10k_counter_collection.csv" > "$OUT/pmc_pairs10k_summary.txt" 2>/dev/null

echo "== config #3: SIFT128 L2 pair matcher: bench line + kernel stats (the roofline of bench.py's l2_sift128 key quotes k_l2_topk_pairs)"
cd "$ROOT" && $PY bench.py --workload l2sift128 > "$OUT/bench_l2sift128.json" 2> "$OUT/bench_l2sift128.err"; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_l2" -o st --output-format csv -- $PY "$ROOT/bench.py" --workload l2sift128 > /dev/null 2>&1
cp "$(find "$OUT/stats_l2" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_l2.csv"
rm -rf "$OUT/stats_l2"

echo "== config #5: AKAZE61 bench + kernel stats"
cd "$ROOT" && $PY bench.py --workload akaze61 --batch 64 --steps 5 > "$OUT/bench_akaze61.json" 2> "$OUT/bench_akaze61.err"; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_akaze" -o st --output-format csv -- $PY "$ROOT/bench.py" --workload akaze61 --batch 64 --steps 3 --cpu-frames 0 > /dev/null 2>&1
cp "$(find "$OUT/stats_akaze" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_akaze61.csv"

echo "== AKAZE scale space + Hessian: fabric traffic per frame (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated as above)"
AKB=16
for C in FETCH_SIZE WRITE_SIZE; do
  cd "$ROOT" && timeout 400 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_akz_$C" -o p --output-format csv -- $PY tools/akaze_scale_space_only.py $AKB 2 > /dev/null 2>&1; cd /tmp
done
$PY - "$OUT" $AKB 2 <<'PYEOF'
import csv, glob, json, sys
out, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cal = json.load(open(out + "/calib_fetch.json"))
tot, per_kernel = {}, {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(out + "/pmc_akz_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == C and "k_akz_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per_kernel.setdefault(k, {}).setdefault(C, 0.0)
                per_kernel[k][C] += float(r["Counter_Value"])
                tot[C] = tot.get(C, 0.0) + float(r["Counter_Value"])
fr, wr = cal.get("fetch_ratio_4B") or 0.5, cal.get("write_ratio_4B") or 1.0
def mb(d):
    return (d.get("FETCH_SIZE", 0.0) * 1024 / fr + d.get("WRITE_SIZE", 0.0) * 1024 / wr) / (B * steps) / 1e6
res = {"frames": B, "scale_space_calls": steps, "hbm_MB_per_frame": mb(tot), "algorithmic_MB_per_frame": 89.3952,
       "per_kernel_MB_per_frame": {k: round(mb(v), 2) for k, v in sorted(per_kernel.items())}}
res["ratio_to_algorithmic"] = res["hbm_MB_per_frame"] / res["algorithmic_MB_per_frame"]
json.dump(res, open(out + "/akaze_traffic_pmc.json", "w"), indent=1)
print(json.dumps(res))
PYEOF

rm -rf "$OUT"/pmc_akz_FETCH_SIZE "$OUT"/pmc_akz_WRITE_SIZE "$OUT"/calib_FETCH_SIZE "$OUT"/calib_WRITE_SIZE "$OUT"/stats_default "$OUT"/stats_pairs "$OUT"/stats_akaze "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ_INSTS_VALU "$OUT"/pmc_LDS "$OUT"/pmc_L2 "$OUT"/pmc_pairs
ls -la "$OUT"
