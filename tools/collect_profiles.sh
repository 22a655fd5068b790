#!/bin/bash
# Run on the GPU box from the repo root (gpurun): final bench lines, rocprofv3 kernel-trace stats and the
# PMC passes behind profiles/<round>/ and profiles/traffic.json.  usage: tools/collect_profiles.sh <tag>
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/${tag}_c3_bench.json 2> $out/${tag}_c3_bench.err || exit 1
python3 bench.py --workload c4_bmp_shard_128px_s32_16spp --steps 100 --warmup 10 > $out/${tag}_c4shard_bench.json 2>/dev/null
python3 bench.py --workload c2_ascii95_128px_s32_16spp --steps 200 --warmup 20 > $out/${tag}_c2_bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $out/kt/kt_kernel_stats.csv $out/${tag}_c3_kernel_stats.csv 2>/dev/null
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --output-format csv --pmc $c -d $out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
done
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc_SQ -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY > $out/${tag}_c3_pmc.txt
import csv, collections, glob
print("rocprofv3 --pmc (separate passes), mean per dispatch of fr::render_kernel<3,4,32>, workload c3 (20992 glyphs x 256^2, S=128, 16 spp)")
for f in sorted(glob.glob("$out/pmc_*/pmc_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel<3" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in sorted(acc.items()): print(f"{c:28s} {sum(v)/len(v):.6g}")
PY
cat $out/${tag}_c3_pmc.txt
head -4 $out/${tag}_c3_kernel_stats.csv
cat $out/${tag}_c3_bench.json
