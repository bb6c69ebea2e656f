"""Turns a rocprofv3 (rocpd sqlite) result into the text kernel-stats summary kept under profiles/:
   python tools/rocprof_summary.py gpurun_out/prof_r01/nin_results.db profiles/r01_....txt "command line" """
import re
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))


def short(name):
    name = name.replace("mc::(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)          # drop the argument list
    return name.replace("void ", "")


with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# command: %s\n" % cmd)
    f.write("%-100s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        f.write("%-100s %8d %14.1f %12.3f %6.2f%%\n" % (short(name)[:100], calls, total, avg, pct))
print(open(out).read()[:3000])
