#!/bin/bash
# usage: tools/kprof.sh <tag> <workload> [lib] — per-kernel average duration (rocprofv3 --kernel-trace --stats) and
# SQ instruction counts per wave (a separate --pmc pass) of one bench workload; summary to gpurun_out/kprof_<tag>.txt
tag=$1; wl=$2; lib=${3:-$GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so}
out=$GRAFT_REPO_ROOT/gpurun_out/kprof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export FR_RASTER_LIB=$lib
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out/p1 -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
if [ -n "$KPROF_FULL" ]; then
  i=1
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --output-format csv --pmc $set -d $out/p$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
  done
fi
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/kprof_$tag.txt
import csv, collections, glob
for f in glob.glob("$out/kt/**/kt_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.2f} pct {r['Percentage']}")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"])
    print(k, f"waves={w:.0f}", " ".join(f"{c[3:]}={sum(x)/len(x)/w:.1f}" for c,x in sorted(cs.items()) if c!="SQ_WAVES"))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/kprof_$tag.txt
