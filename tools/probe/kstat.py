"""Print the average duration of the kernels whose name contains one of the given substrings, from a rocprofv3 kernel_stats csv
found under a directory:  python tools/probe/kstat.py <dir> <substr> [<substr> ...]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[0]
for r in csv.DictReader(open(f)):
    if any(s in r["Name"] for s in sys.argv[2:]):
        n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print("%s %.1f us;" % (n, float(r["AverageNs"]) / 1e3), end=" ")
print()
