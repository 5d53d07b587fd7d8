#!/usr/bin/env python3
"""Per-kernel medians of a rocprofv3 --pmc pass (counter_collection.csv) -> small CSV for profiles/.
usage: python tools/summarize_pmc.py FETCH_DIR WRITE_DIR OUT.csv"""
import collections
import csv
import glob
import os
import statistics
import sys


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return out


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    with open(sys.argv[3], "w") as fh:
        fh.write("kernel,launches,median_FETCH_SIZE_KB,max_FETCH_SIZE_KB,median_WRITE_SIZE_KB,hbm_bytes_per_working_launch=(2*FETCH+WRITE)*1024\n")
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            f = fe[k]; w = wr.get(k, [0.0])
            work = [v for v in f if v > 0.5 * max(f)] or f
            fh.write('"%s",%d,%.3f,%.3f,%.3f,%.0f\n' % (k, len(f), statistics.median(work), max(f), statistics.median(w),
                                                       (2 * statistics.median(work) + statistics.median(w)) * 1024))


if __name__ == "__main__":
    main()
