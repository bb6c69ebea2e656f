"""Builds profiles/pmc_traffic.json (bench.py's roofline.traffic) from the PMC summaries of one profile round:
   python tools/make_pmc_traffic.py <pmc_fetch.txt> <pmc_write.txt> <pmc_sq.txt> <bench.json> <kernel_stats.txt>  > pmc_traffic.json
The launch looked up is the one bench.py reports in roofline.launch (instantiation + grid of NIN's conv2 forward); one
instantiation can serve several layers on one grid, so among the "shapes" the summaries list for it the one whose duration is
closest to bench.py's event time is taken.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 counts a 128-byte
read request as 64, MI355X_MICROARCH.md)."""
import json
import re
import sys

fetch_f, write_f, sq_f, bench_f, stats_f = sys.argv[1:6]
merge_f = sys.argv[6] if len(sys.argv) > 6 else None      # an existing pmc_traffic.json to add this launch to
bench = json.load(open(bench_f))
launch = bench["roofline"]["launch"]
m = re.match(r"(mc_igemm<[^>]*>) grid=\((\d+),(\d+),(\d+)\)", launch)
kernel, grid = m.group(1), "grid=(%s,%s,%s)" % m.group(2, 3, 4)
want_us = bench["roofline"]["avg_launch_ms"] * 1e3


def entries(path):
    """[(avg_us, {counter: mean})] of the launch's shapes in one summary file"""
    out, cur = [], None
    for line in open(path):
        if not line.startswith(" "):
            cur = None
            if line.startswith(kernel + " ") and grid in line:
                cur = (float(re.search(r"avg_us\(profiled\)=([\d.]+)", line).group(1)), {})
                out.append(cur)
        elif cur is not None:
            mm = re.match(r"\s+(\S+)\s+mean=([-\d.e+]+)", line)
            if mm:
                cur[1][mm.group(1)] = float(mm.group(2))
    return out


def pick(path):
    es = entries(path)
    if not es:
        sys.exit("make_pmc_traffic: %s has no entry for %s %s" % (path, kernel, grid))
    return min(es, key=lambda e: abs(e[0] - want_us))


f_us, f = pick(fetch_f)
w_us, w = pick(write_f)
s_us, sq = pick(sq_f)
trace_us = None
for line in open(stats_f):
    if line.startswith(kernel) and ("(%s)" % grid[6:-1]) in line.replace(" ", ""):
        parts = line.split()
        # kernel ... grid calls avg_us total_us vgpr lds : the shape closest to the event time
        try:
            avg = float(parts[-4])
        except ValueError:
            continue
        if trace_us is None or abs(avg - want_us) < abs(trace_us - want_us):
            trace_us = avg
hbm = int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024)
busy = None
if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "SQ_BUSY_CYCLES" in sq and sq["SQ_BUSY_CYCLES"] > 0:
    busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
entry = {
    "launch": launch,
    "what": "NIN conv2 forward (192 x 32768 x 2400 implicit GEMM + bias + relu + relu bit mask), the launch bench.py's roofline times",
    "fetch_size_kib": f["FETCH_SIZE"], "write_size_kib": w["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
    "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests as 64 B)",
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes"],
    "mfma_busy_fraction": busy,
    "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 SE x 1024 SIMD), profiled run",
    "profiled_avg_us": {"fetch_pass": f_us, "write_pass": w_us, "sq_pass": s_us},
    "kernel_trace_avg_us": trace_us, "bench_event_avg_us": want_us,
    "source": [fetch_f.split("/")[-1], write_f.split("/")[-1], sq_f.split("/")[-1], stats_f.split("/")[-1]],
}
# The autotuner's pick for conv2 forward moves between two tiles that time within a few per cent of each other, so the file
# holds one entry per launch it may make; bench.py reports the traffic of the one it actually timed.
out = {"launches": {}}
if merge_f:
    try:
        out = json.load(open(merge_f))
        if "launches" not in out:
            out = {"launches": {out["launch"]: out}} if "launch" in out else {"launches": {}}
    except (OSError, ValueError):
        pass
out["launches"][launch] = entry
json.dump(out, sys.stdout, indent=1)
