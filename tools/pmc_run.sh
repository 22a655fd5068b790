#!/bin/bash
# usage: tools/pmc_run.sh <outdir-name> <counters...>   (on the GPU box; run from the repo root)
# collects rocprofv3 PMC counters for a short bench.py run and prints the render kernel's means
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/$out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/$out/pmc_counter_collection.csv')):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if 'render' not in k: continue
    for c,v in sorted(cs.items()): print(f"{c:32s} {sum(v)/len(v):.5g}")
PY
