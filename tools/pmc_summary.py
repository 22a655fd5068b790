#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch per kernel."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "render_kernel" not in k and "prepare" not in k: continue
            print(k)
            for c, v in sorted(cs.items()):
                # counter rows may be split per dimension: sum rows per dispatch = total/num dispatches
                print(f"   {c:28s} mean/row={sum(v)/len(v):.4g} rows={len(v)} total={sum(v):.6g}")
