"""Ordered timeline of the LAST step in a rocprofv3 --kernel-trace database of `bench.py --trace-only [--sequential]`:
per kernel its start offset, duration and the gap since the previous kernel's end (all in us).
usage: python scripts/trace_timeline.py <results.db> [--marker KERNEL_SUBSTRING [--per-step N]]
(--marker: a step starts at every N-th launch of the kernels matching the substring -- e.g. `--marker adam_kernel` for
`bench.py --wlocal-only --trace-only`, whose step ends with its Adam launch)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, workgroup_x, start, end from kernels order by start"))
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
# a step starts at the first persistent projection GEMM after a non-IMU kernel: find the starts of the 4-GEMM groups
marker, per_step, shift = "gemm_tile_persistent_kernel<true, true", 4, 0
if "--marker" in sys.argv:
    marker, per_step, shift = sys.argv[sys.argv.index("--marker") + 1], 1, 1      # (the marker is a step's LAST launch)
if "--per-step" in sys.argv:
    per_step = int(sys.argv[sys.argv.index("--per-step") + 1])
gem = [i + shift for i, r in enumerate(rows) if marker in r[0]]
starts = gem[::per_step]
if len(starts) < 2:
    sys.exit("not enough steps in the trace")
# walk back from a step's first projection GEMM to the first kernel after the previous step's Adam
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = seg[0][3]
prev_end = t0
print("%-52s %9s %8s %7s" % ("kernel (WGs)", "start_us", "dur_us", "gap_us"))
for n, gx, wx, s, e in seg:
    print("%-52s %9.1f %8.1f %7.1f" % ((short(n)[:42] + " (%d)" % (gx // max(wx, 1))), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("segment: %d launches, %.3f ms" % (len(seg), (prev_end - t0) / 1e6))
