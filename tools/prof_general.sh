#!/bin/bash
# what is left on the general render_kernel (glyphs of > 768 segments): bench line, rocprofv3 kernel stats, SQ counters
out=$GRAFT_REPO_ROOT/gpurun_out/prof_general; mkdir -p $out
cd $GRAFT_REPO_ROOT
W=huge_s1024_512cells_256px_16spp
python3 bench.py --workload $W --steps 20 --warmup 5 --cpu-seconds 3 > $out/${W}_bench.json 2> $out/bench.err
python3 tools/show_bench.py $out/${W}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $out/kt/kt_kernel_stats.csv $out/${W}_kernel_stats.csv; rm -rf $out/kt
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
python3 - $out/pmc/pmc_counter_collection.csv <<'PY' > $out/${W}_sq.txt
import csv, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if "render_kernel" not in k and "prepare" not in k: continue
    for c,v in sorted(cs.items()): print(f"{k:60s} {c:22s} {sum(v)/len(v):.6g} x{len(v)}")
PY
rm -rf $out/pmc; cat $out/${W}_sq.txt; head -4 $out/${W}_kernel_stats.csv
