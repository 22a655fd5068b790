#!/bin/bash
# kernel time of the headline workload at 4 / 3 / 2 workgroups per CU (extra dynamic LDS per workgroup)
for pad in 0 14000 28000; do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --opt lds_pad=$pad "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds_pad', $pad, d['roofline']['kernel_ms'])"
done
