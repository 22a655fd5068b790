#!/bin/bash
# usage: tools/pmc_variants.sh <variant...>  — VALU/SALU/LDS instruction counts per wave of the render
# kernel for diagnostic builds (libfr_raster_var_<name>.so; "main" = the product library), 2048 glyphs of C3
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster_var_$v.so
  [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so
  FR_RASTER_LIB=$lib rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/pmcv_$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --glyphs 2048 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/pmcv_$v/pmc_counter_collection.csv')):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if 'render' not in k: continue
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"])
    print("$v", " ".join(f"{c[9:]}={sum(x)/len(x)/w:.0f}" for c,x in sorted(cs.items()) if c!="SQ_WAVES"), f"waves={w:.0f}")
PY
done
