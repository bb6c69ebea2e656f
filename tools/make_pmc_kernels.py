"""Adds the "kernels" section of profiles/pmc_traffic.json: HBM bytes per launch of the kernels behind bench.py's other roofline
rows (roofline_worst, roofline_hbm and every row of its "kernels" list), from the single-stream (MC_NO_STEP_FORK=1) runs of one
profile round:
   python tools/make_pmc_kernels.py <event_table.json> <kernel_stats.txt> <pmc_fetch.txt> <pmc_write.txt> <pmc_traffic.json>
event_table.json is bench.py --kernel-table of the kernel-trace run (profile key -> event time); kernel_stats.txt its rocprofv3
summary split by launch shape; the two pmc files the FETCH_SIZE / WRITE_SIZE passes.  A profile key is tied to a traced launch by
the kernel family its entry point launches (table below) and, inside the family, by the steady-state shape (>= 30 calls) whose
traced duration is closest to the key's event time -- within 25 %, or the key gets no entry; keys under 25 us are left out (an
event pair adds ~4.5 us to what it times, too much to tell small launches apart).  HBM bytes = (2 * FETCH_SIZE +
WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md).  bench.py reports an entry only while its own event time for the key
agrees with the recorded duration."""
import json
import re
import sys

table_f, stats_f, fetch_f, write_f, out_f = sys.argv[1:6]

FAMILY = [   # (profile-key regex, kernel-name regex)
    (r"/fwd:mc_conv_filter_acts", r"mc_igemm<MatNatAddr, PatchNat"),
    (r"/bwd:mc_conv_img_acts", r"mc_igemm<MatKcAddr, HidNat"),
    (r"/wgrad:mc_conv_weight_grad", r"mc_igemm<PatchKc\w*, MatKc"),
    (r"/fwd:mc_gemm", r"mc_igemm<MatKcAddr, MatNatAddr"),    # W (out x in, k-contiguous) . x
    (r"/bwd:mc_gemm", r"mc_igemm<MatNatAddr, MatNatAddr"),   # W^T . grad: both operands row-contiguous
    (r"/wgrad:mc_gemm", r"mc_igemm<MatKc\w*, MatKc"),
    (r":mc_conv_local_pool", r"k_pool"),
    (r":mc_conv_local_max_undo", r"k_max_undo"),
    (r":mc_conv_local_avg_undo", r"k_avg_undo"),
    (r":mc_dropout", r"k_dropout"),
]


def shapes(path):
    """[(kernel, grid, calls, avg_us)] of the by-shape half of a kernel-stats summary"""
    out, on = [], False
    for line in open(path):
        if line.startswith("# the same trace split by launch shape"):
            on = True
            continue
        if not on or line.startswith("kernel ") or not line.strip():
            continue
        m = re.match(r"(.*?)\s+\((\d+),(\d+),(\d+)\)(?: #\d+)?\s+(\d+)\s+([\d.]+)\s", line)
        if m:
            out.append((m.group(1).strip(), "grid=(%s,%s,%s)" % m.group(2, 3, 4), int(m.group(5)), float(m.group(6))))
    return out


def pmc_entries(path):
    """[(kernel, grid, avg_us, {counter: mean})]"""
    out, cur = [], None
    for line in open(path):
        if not line.startswith(" "):
            cur = None
            m = re.match(r"(.*?)\s+(grid=\(\d+,\d+,\d+\)).*avg_us\(profiled\)=([\d.]+)", line)
            if m:
                cur = (m.group(1).strip(), m.group(2), float(m.group(3)), {})
                out.append(cur)
        elif cur is not None:
            mm = re.match(r"\s+(\S+)\s+mean=([-\d.e+]+)", line)
            if mm:
                cur[3][mm.group(1)] = float(mm.group(2))
    return out


sh, fe, wr = shapes(stats_f), pmc_entries(fetch_f), pmc_entries(write_f)
kernels, skipped = {}, []
steady = sorted(s[2] for s in sh if s[2] >= 30)
per_step = steady[len(steady) // 2] if steady else 1            # calls of a once-a-step launch in the trace (the median count)
left = dict((i, max(1, int(round(s[2] / float(per_step))))) for i, s in enumerate(sh))   # how many keys a traced shape can stand for
for row in sorted(json.load(open(table_f)), key=lambda r: -r["avg_ms"]):
    key, us = row["kernel"], row["avg_ms"] * 1e3
    fam = next((k for p, k in FAMILY if re.search(p, key)), None)
    if fam is None or us < 25.0:      # below that the event pair itself (~4.5 us) blurs which traced shape a key is
        continue
    cand = [i for i, s in enumerate(sh) if re.match(fam, s[0]) and s[2] >= 30 and left[i] > 0]
    if not cand:
        skipped.append((key, "no traced launch of its family left"))
        continue
    bi = min(cand, key=lambda i: abs(sh[i][3] - us))
    best = sh[bi]
    if abs(best[3] - us) > 0.25 * us:
        skipped.append((key, "event %.1f us, nearest traced shape %.1f us" % (us, best[3])))
        continue
    def near(entries):
        es = [e for e in entries if e[0] == best[0] and e[1] == best[1]]
        return min(es, key=lambda e: abs(e[2] - best[3])) if es else None
    f, w = near(fe), near(wr)
    if f is None or w is None or "FETCH_SIZE" not in f[3] or "WRITE_SIZE" not in w[3]:
        skipped.append((key, "no counter entry for %s %s" % (best[0], best[1])))
        continue
    e = {"launch": "%s %s" % (best[0], best[1]), "fetch_size_kib": f[3]["FETCH_SIZE"], "write_size_kib": w[3]["WRITE_SIZE"],
         "hbm_bytes_per_launch": int((2 * f[3]["FETCH_SIZE"] + w[3]["WRITE_SIZE"]) * 1024),
         "kernel_trace_avg_us": best[3], "event_avg_us_single_stream": us, "profiled_avg_us": {"fetch_pass": f[2], "write_pass": w[2]}}
    if row.get("bound") == "hbm":
        e["algorithmic_bytes_per_launch"] = int(row["achieved"] * 1e9 * row["avg_ms"] * 1e-3)
    kernels[key] = e
    left[bi] -= 1

try:
    out = json.load(open(out_f))
except (OSError, ValueError):
    out = {"launches": {}}
out["kernels"] = kernels
out["kernels_source"] = [p.split("/")[-1] for p in (table_f, stats_f, fetch_f, write_f)]
json.dump(out, open(out_f, "w"), indent=1)
for k, why in skipped:
    sys.stderr.write("make_pmc_kernels: %s: %s\n" % (k, why))
print("make_pmc_kernels: %d keys" % len(kernels))
