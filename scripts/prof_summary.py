"""Summarise a rocprofv3 rocpd database (--kernel-trace) of bench.py: per-kernel and per-(kernel, grid) time per step.
usage: python scripts/prof_summary.py <results.db> [steps_divisor]   (default: launches of the projection product / 4)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, end-start from kernels order by start"))
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
# 4 batched input projections per U+L step, whichever kernel runs them (r06: gemm_tile_big / s3_gemm_big)
nper = sum(1 for r in rows if any(t in r[0] for t in ("gemm_tile_persistent_kernel<true, true", "gemm_tile_big_kernel", "s3_gemm_big_kernel"))) / 4.0
pos = [a for a in sys.argv[2:] if not a.startswith("--")]
steps = float(pos[0]) if pos else (nper or 1.0)
tot = sum(r[5] for r in rows)
print("steps %.1f  kernel time %.3f ms/step  launches/step %.0f" % (steps, tot / steps / 1e6, len(rows) / steps))
agg = collections.defaultdict(lambda: [0, 0])
agg2 = collections.defaultdict(lambda: [0, 0])
for n, gx, gy, gz, wx, d in rows:
    agg[short(n)][0] += 1
    agg[short(n)][1] += d
    k = (short(n), gx // max(wx, 1), gy, gz)
    agg2[k][0] += 1
    agg2[k][1] += d
print("\nName,LaunchesPerStep,AvgUs,MsPerStep,Percent")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%s,%.1f,%.1f,%.3f,%.1f" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e6, 100.0 * v[1] / tot))
if "--grids" in sys.argv:
    print("\nName,WGs,LaunchesPerStep,AvgUs,MsPerStep")
    for k, v in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%s,(%d;%d;%d),%.1f,%.1f,%.3f" % (k[0], k[1], k[2], k[3], v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e6))
