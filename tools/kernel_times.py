"""Prints kernel name, calls and average duration (us) from a rocprofv3 kernel_stats.csv."""
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", row["Name"]).split("(")[0]
    print("%-40s calls %5s  avg %10.1f us" % (name[-40:], row["Calls"], float(row["AverageNs"]) / 1e3))
