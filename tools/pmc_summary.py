#!/usr/bin/env python3
"""Mean of one PMC counter per kernel from a rocprofv3 --pmc ... --output-format csv run.

    python tools/pmc_summary.py <dir> <counter> [name-substring ...]
"""
import collections
import csv
import glob
import sys


def main(d, counter, subs):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            if subs and not any(s in name for s in subs):
                continue
            key = next((x for x in subs if x in name), None) or name[:60]
            if key == "Cijk":
                key = name[:18] + ".." + name[name.find("_MT"):name.find("_MT") + 22]
            acc[key].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v2 = v[len(v) // 2:]        # second half of the dispatches (steady state)
        print(f"{counter} {k:62s} n={len(v):4d} mean={sum(v2) / len(v2):12.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
