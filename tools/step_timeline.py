"""One training step as the GPU saw it: from a rocprofv3 --kernel-trace database, the last complete steps' kernels in launch
order with duration and the idle gap before each (a step starts at the first kernel after k_weights_update_spans).
   python tools/step_timeline.py <results.db> [steps-from-the-end=3]"""
import re
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, start, end, grid_x / workgroup_x, grid_y / workgroup_y, grid_z / workgroup_z from kernels order by start"))


def short(n):
    n = n.replace("mc::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", n)


ends = [i for i, r in enumerate(rows) if "k_weights_update_spans" in r[0]]
if len(ends) < back + 1:
    sys.exit("fewer than %d steps in the trace" % (back + 1))
a, b = ends[-back - 1] + 1, ends[-back] + 1
step = rows[a:b]
busy = sum(r[2] - r[1] for r in step) / 1e3
span = (step[-1][2] - rows[a - 1][2]) / 1e3
print("step of %d launches: span %.1f us, kernels %.1f us, idle %.1f us" % (len(step), span, busy, span - busy))
prev = rows[a - 1][2]
for r in step:
    print("%7.1f us  gap %5.1f  %-86s (%d,%d,%d)" % ((r[2] - r[1]) / 1e3, (r[1] - prev) / 1e3, short(r[0])[:86], r[3], r[4], r[5]))
    prev = r[2]
