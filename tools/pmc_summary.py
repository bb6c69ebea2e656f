"""Per-kernel PMC counter means from a rocprofv3 --pmc result (rocpd sqlite): python tools/pmc_summary.py db [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = list(cur.execute("select * from counters_collection"))
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix["kernel_name"]] if "kernel_name" in ix else str(r[ix.get("kernel_id", 0)])
    if sub and sub not in name: continue
    agg[name[:90]][r[ix["counter_name"]]].append(r[ix["value"]])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
