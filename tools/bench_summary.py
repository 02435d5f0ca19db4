"""One line per record of a bench.py line (headline + sub-records): python tools/bench_summary.py bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = d.get("roofline") or {}
print("headline %s MB/s  encode %s ms  decode %s ms  bit_exact %s  frac %s  traffic %s (%s GB/s)" % (
    d["value"], d.get("encode_ms"), d.get("decode_ms"), d["bit_exact"], rf.get("frac"), rf.get("traffic"), rf.get("traffic_GBps")))
for k in ("rough", "lsop", "canon", "dem1024", "float256_lsop", "float256", "compact"):
    r = d.get(k)
    if r:
        print("  %-14s" % k, {x: r[x] for x in r if x in ("encode_ms", "decode_ms", "MBps", "bit_exact", "compact_ms", "checked")},
              "frac", (r.get("roofline") or {}).get("frac"))
cb = d.get("cpu_baseline") or {}
print("  cpu_baseline", cb.get("value"), cb.get("unit"), "cores", cb.get("cores"), "| all cores", (cb.get("all_cores") or {}).get("value"))
