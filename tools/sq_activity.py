import csv, glob, sys, json
out = sys.argv[1]
acc = {}
for p in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith(sys.argv[2] if len(sys.argv) > 2 else "k_akz"): continue
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = {}
for p in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(sys.argv[2] if len(sys.argv) > 2 else "k_akz"): dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in acc.items():
    s = {n: sum(v) for n, v in c.items()}
    tot_us = sum(dur.get(k, [0]))
    if tot_us < 50: continue
    busy = s.get("SQ_BUSY_CU_CYCLES", 0)
    rows.append((tot_us, k, {"us_total": round(tot_us), "sclk_GHz": round(busy / 256 / tot_us / 1e3, 2) if tot_us else None,
        "valu_busy": round(s.get("SQ_ACTIVE_INST_VALU", 0) / busy, 3) if busy else None,
        "lds_busy": round(s.get("SQ_LDS_IDX_ACTIVE", 0) / busy, 3) if busy else None,
        "salu_busy": round(s.get("SQ_INST_CYCLES_SALU", 0) / busy, 3) if busy else None,
        "waves_per_simd": round(s.get("SQ_WAVE_CYCLES", 0) / busy, 2) if busy else None,
        "wait_cnt": round(s.get("SQ_WAIT_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1), 3)}))
for _, k, d in sorted(rows, reverse=True): print(k[:40], d)
