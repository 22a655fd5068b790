#!/bin/bash
# usage: tools/pmc_ab.sh <tag> [bench.py args...] — SQ counters per wave of every render kernel of one bench
# configuration (separate --pmc passes), summary to gpurun_out/pmc_<tag>.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcab_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d $out/p$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --glyphs 4096 "$@" > /dev/null 2>&1
done
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
import csv, collections, glob
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if 'render' not in k and 'cov4' not in k: continue
    w=sum(cs["SQ_WAVES"])/len(cs["SQ_WAVES"])
    print(k, f"waves={w:.0f}")
    for c,x in sorted(cs.items()):
        if c!="SQ_WAVES": print(f"   {c:26s} per wave {sum(x)/len(x)/w:12.1f}   total {sum(x)/len(x):.4g}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
