#!/bin/bash
# usage: tools/c4_ablate_wl.sh <workload> <variant...> — kernel time + VALU/SALU/LDS per wave for a workload
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster_var_$v.so
  [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so
  ms=$(FR_RASTER_LIB=$lib python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  FR_RASTER_LIB=$lib rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/c4wl_$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/c4wl_$v/pmc_counter_collection.csv')):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if 'cov4' not in k and 'win1' not in k: continue
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"])
    print("$wl $v", "kernel_ms=$ms", " ".join(f"{c[9:]}={sum(x)/len(x)/w:.0f}" for c,x in sorted(cs.items()) if c!="SQ_WAVES"), f"waves={w:.0f}")
PY
done
