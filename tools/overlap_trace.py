"""Do the fast decode kernel's two runs overlap?  Reads a rocprofv3 --kernel-trace csv and prints, for the last decode of the
run, start / end of every dispatch relative to the pre-pass (us) with its stream / queue.
usage: python tools/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "parse_trees" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:last + 6]:
    name = r["Kernel_Name"].split("(")[0][-40:]
    print("%-42s queue %-4s start %9.1f us  end %9.1f us  grid %s lds %s" % (
        name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
        r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?")))
