#!/bin/bash
# usage: [WORKLOAD=<bench workload>] tools/c4_ablate.sh <variant...>  ("main" = the product library; else libfr_raster_var_<name>.so):
# kernel time on the full workload (default: the headline, C3) and VALU / SALU / LDS instructions per wave (2048 glyphs) of cov4_kernel
W=${WORKLOAD:-c3_cjk21k_256px_s128_16spp}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster_var_$v.so
  [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so
  ms=$(FR_RASTER_LIB=$lib python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  FR_RASTER_LIB=$lib rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/c4ab_$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 1 --warmup 0 --glyphs 2048 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/c4ab_$v/pmc_counter_collection.csv')):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if 'cov4' not in k and 'render' not in k: continue
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"])
    print("$W", "$v", k[:36], "kernel_ms=$ms", " ".join(f"{c[9:]}={sum(x)/len(x)/w:.0f}" for c,x in sorted(cs.items()) if c!="SQ_WAVES"), f"waves={w:.0f}")
PY
done
