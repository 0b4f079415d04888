"""Condense scripts/collect_config5.sh output: per-kernel time of one config-5 forward (rocpd kernel trace) joined with the
PMC passes (HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB on gfx950, MFMA busy fraction) -> JSON on stdout.
usage: python scripts/config5_summary.py gpurun_out/prof_<tag>_c5 > profiles/<tag>_config5.json"""
import collections
import csv
import glob
import json
import os
import re
import sqlite3
import sys

root = sys.argv[1]
FORWARDS = 3.0       # --trace: 1 warm-up + 2 timed forwards
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
dbs = glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True)
db = sqlite3.connect(dbs[0])
rows = list(db.execute("select name, end-start from kernels order by start"))
# the warm-up forward (weight packs, allocations) is dropped: a forward ends with Lower_Net's head_fk_fwd_kernel<1>
LAST = "head_fk_fwd_kernel<1>"
ends = [i for i, (n, _) in enumerate(rows) if LAST in n]
if len(ends) >= 2:
    rows = rows[ends[0] + 1:]
    FORWARDS = float(len(ends) - 1)
agg = collections.defaultdict(lambda: [0, 0])
for n, d in rows:
    agg[short(n)][0] += 1
    agg[short(n)][1] += d
tot = sum(v[1] for v in agg.values())

pmc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    recs = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    first_end = next((int(r["Dispatch_Id"]) for r in recs if LAST in r["Kernel_Name"]), -1)
    for r in recs:
        if int(r["Dispatch_Id"]) <= first_end:
            continue
        k = short(r["Kernel_Name"])
        pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

kern = []
for k, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    e = {"kernel": k, "launches_per_forward": cnt / FORWARDS, "avg_us": ns / cnt / 1e3, "ms_per_forward": ns / FORWARDS / 1e6,
         "share_of_kernel_time": ns / tot}
    c = pmc.get(k)
    if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        by = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / FORWARDS
        e["hbm_GB_per_forward"] = by / 1e9
        e["hbm_TBps_in_trace"] = by / (ns / FORWARDS * 1e-9) / 1e12          # PMC bytes over the traced duration
        e["hbm_fraction_of_8TBps"] = e["hbm_TBps_in_trace"] / 8.0
    if c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    kern.append(e)
out = {"config": "BASELINE config 5: IMU_Net -> Upper_Net -> Lower_Net forward, B=2048 T=16 N=256, precision = 'bf16' on all three nets "
                 "(r04: IMU_Net's BiLSTM products, Upper_Net's PointNet / GlobalPointNet stages, Lower_Net's ST-GCN products on bf16 "
                 "operands with fp32 accumulation; r02 / r03 files: the IMU products only)",
       "kernel_time_ms_per_forward": tot / FORWARDS / 1e6, "kernels": kern,
       "note": "rocprofv3 --kernel-trace of `scripts/bench_config5.py --trace` (the forwards behind the warm-up one); PMC in separate passes (FETCH_SIZE | "
               "WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), FETCH doubled for gfx950 per MI355X_MICROARCH.md"}
txt = os.path.join(root, "bench_config5.txt")
if os.path.exists(txt):
    out["bench_config5_stdout"] = [l.rstrip() for l in open(txt) if "amdgpu.ids" not in l]
print(json.dumps(out, indent=1))
