"""Turns a rocprofv3 --kernel-trace --stats result (rocpd sqlite) into the text summary kept under profiles/:
   python tools/rocprof_summary.py gpurun_out/prof_r01/trace/nin_results.db profiles/r01_....txt "command line"
Part 1 is rocprofv3's own per-kernel statistics (top_kernels).  Part 2 splits every kernel by launch
shape (grid in workgroups), because one mc_igemm instantiation serves several layers: NIN's conv2
forward is mc_igemm<MatNatAddr, PatchNatAddr, ...> with grid (256,2,1) -- the launch bench.py's
`roofline` object times."""
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
    f.write("\n# the same trace split by launch shape (grid in workgroups); registers and LDS per workgroup\n")
    f.write("%-72s %-16s %6s %12s %12s %5s %7s\n" % ("kernel", "grid", "calls", "avg_us", "total_us", "vgpr", "lds"))
    # One instantiation serves several layers, and persistent column workers cap grid.x at the resident-workgroup
    # count, so two layers can even share a grid: launches of one (kernel, grid) are split further into shapes
    # wherever the sorted durations jump by more than 1.6x.
    q = ("select name, grid_x / workgroup_x, grid_y / workgroup_y, grid_z / workgroup_z, duration, vgpr_count, lds_size from kernels")
    groups = {}
    for name, gx, gy, gz, dur, vgpr, lds in cur.execute(q):
        groups.setdefault((name, gx, gy, gz), []).append((dur, vgpr, lds))
    lines = []
    for (name, gx, gy, gz), ls in groups.items():
        ls.sort()
        shapes, lo = [], None
        for e in ls:
            if lo is None or e[0] > 1.6 * lo:
                shapes.append([])
                lo = e[0]
            shapes[-1].append(e)
        for si, sh in enumerate(shapes):
            total = sum(e[0] for e in sh) / 1e3
            grid = "(%d,%d,%d)" % (gx, gy, gz) + ("" if len(shapes) == 1 else " #%d" % (si + 1))
            lines.append((total, "%-72s %-16s %6d %12.3f %12.1f %5d %7d\n" % (short(name)[:72], grid, len(sh), total / len(sh), total,
                                                                            max(e[1] for e in sh), max(e[2] for e in sh))))
    for _, line in sorted(lines, reverse=True):
        f.write(line)
print(open(out).read()[:6000])
