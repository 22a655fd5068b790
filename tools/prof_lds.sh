#!/bin/bash
# LDS pipeline counters of a bench workload (one --pmc pass)
out=$GRAFT_REPO_ROOT/gpurun_out/prof_lds; mkdir -p $out
W=${1:-c3_cjk21k_256px_s128_16spp}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $out/pmc -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
python3 - $out/pmc/pmc_counter_collection.csv <<'PY' > $out/${W}_lds.txt
import csv, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if not any(t in k for t in ("cov4","win1","sdf_kernel","render_kernel")): continue
    for c,v in sorted(cs.items()): print(f"{k:60s} {c:24s} {sum(v)/len(v):.6g} x{len(v)}")
PY
rm -rf $out/pmc; cat $out/${W}_lds.txt
