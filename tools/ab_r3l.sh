#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "sdf or SDF" > $out/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_sdf.log
for w in c5_sdf_shard_512px_s64 real_dejavuserif_italic_whole_font_512px_sdf; do
  timeout -k 10 200 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
  python tools/show_bench.py $out/bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
for w in c5_sdf_shard_512px_s64 real_dejavuserif_italic_whole_font_512px_sdf; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d $out/pmc_${w}_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    python3 - $out/pmc_${w}_$c/pmc_counter_collection.csv <<'PY'
import csv, collections, sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Kernel_Name"] for t in ("win1_kernel<4", "win1_kernel<3", "sdf_kernel")): acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
  done
done
