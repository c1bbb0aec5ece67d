"""Condensed per-kernel table of a rocprofv3 `--kernel-trace --stats --output-format csv` run: python scripts/kstats.py <dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|ahip::|void ", "", r["Name"])
    n = re.sub(r"\(.*", "", n)
    print("%-72s calls=%6s avg_us=%9.1f min=%8.1f max=%8.1f" % (n[:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
