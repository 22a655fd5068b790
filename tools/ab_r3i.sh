#!/bin/bash
# same-box A/B: the round-2 tree against this tree, alternating.  Set-up (here, before gpurun; untracked, travels with the
# snapshot):  mkdir .ab_r02 && git archive 9a0dd80 | tar -x -C .ab_r02 && make -C .ab_r02/font-renderer_amd/csrc -j6 && make -C .ab_r02/oracle
out=gpurun_out/r3i; mkdir -p $out
for rep in 1; do
for w in c3_cjk21k_256px_s128_16spp c3_cjk21k_256px_s128_gray_debug c3_strokes21k_256px_s128_16spp c3_cjk21k_256px_s128_winding_i16; do
  (cd .ab_r02 && timeout -k 10 120 python bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > ../$out/r02_${rep}_$w.json 2> ../$out/r02_${rep}_$w.err)
  python - $out/r02_${rep}_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("r02", d["config"]["workload"], "step", d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"])
PY
  timeout -k 10 120 python bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > $out/r03_${rep}_$w.json 2> $out/r03_${rep}_$w.err
  python tools/show_bench.py $out/r03_${rep}_$w.json | head -1 | sed "s/^/r03 /"
done
done
