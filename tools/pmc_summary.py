"""Per-kernel (and per launch shape) means of the counters in a rocprofv3 --pmc result (rocpd sqlite):
   python tools/pmc_summary.py results.db [kernel-substring] [--out file]
Launches are grouped by (kernel, grid, workgroup) so that instances of one template that serve
different layers (conv1 / conv2 / conv3 forward all run mc_igemm<MatNatAddr, PatchNatAddr, ...>) are
listed apart.  FETCH_SIZE / WRITE_SIZE are converted to bytes: rocprofv3 reports KiB, and on gfx950
FETCH_SIZE tallies each 128-byte request as 64 bytes (MI355X_MICROARCH.md, section HBM), so it is doubled."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = name.replace("mc::(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void ", "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = sys.stdout
    if "--out" in sys.argv:
        out = open(sys.argv[sys.argv.index("--out") + 1], "w")
        args = [a for a in args if a != sys.argv[sys.argv.index("--out") + 1]]
    cur = sqlite3.connect(args[0]).cursor()
    sub = args[1] if len(args) > 1 else ""
    agg = collections.OrderedDict()
    q = ("select kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, vgpr_count, accum_vgpr_count, "
         "lds_block_size, counter_name, value, duration, dispatch_id from counters_collection order by dispatch_id")
    for name, gx, gy, gz, wx, vg, ag, lds, cn, val, dur, did in cur.execute(q):
        if sub and sub not in name:
            continue
        key = (short(name)[:80], gx // max(wx, 1), gy, gz, vg, ag, lds)
        d = agg.setdefault(key, {"n": set(), "dur": {}, "c": collections.defaultdict(list)})
        d["n"].add(did)
        d["dur"][did] = dur
        d["c"][cn].append(val)
    for key, d in agg.items():
        durs = list(d["dur"].values())
        out.write("%s  grid=(%d,%d,%d) vgpr=%d agpr=%d lds=%d  launches=%d  avg_us(profiled)=%.1f\n"
                  % (key + (len(d["n"]), sum(durs) / len(durs) / 1e3)))
        for c, v in sorted(d["c"].items()):
            mean = sum(v) / len(v)
            extra = ""
            if c == "FETCH_SIZE":
                extra = "  -> %.3f MB per launch (KiB x 1024 x 2, gfx950 correction)" % (mean * 1024 * 2 / 1e6)
            elif c == "WRITE_SIZE":
                extra = "  -> %.3f MB per launch (KiB x 1024)" % (mean * 1024 / 1e6)
            out.write("   %-28s mean=%.6g%s\n" % (c, mean, extra))


if __name__ == "__main__":
    main()
