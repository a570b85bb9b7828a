"""Summarise rocprofv3 --pmc CSV output per kernel (sums over dispatches). usage: python tools/pmc_summary.py DIR [out.txt]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", "")).replace("void ", "").replace("fdgs::", "")
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row.get("Dispatch_Id"))
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["%-44s %6s " % ("kernel", "calls") + " ".join("%22s" % c[-22:] for c in counters)]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        n = max(len(calls[k]), 1)
        lines.append("%-44s %6d " % (k[:44], n) + " ".join("%22.4g" % (v.get(c, 0.0) / n) for c in counters))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
