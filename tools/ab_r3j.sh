#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
run() { name=$1; lib=$2; w=$3; shift 3
  FR_RASTER_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline "$@" > $out/${name}_$w.json 2> $out/${name}_$w.err
  python tools/show_bench.py $out/${name}_$w.json | head -1 | sed "s/^/$name /"; }
D=font-renderer_amd/libfr_raster.so
for w in c3_cjk21k_256px_s128_gray_debug c3_cjk21k_256px_s128_16spp; do
  (cd .ab_r02 && timeout -k 10 120 python bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > ../$out/r02_$w.json 2>/dev/null)
  python - $out/r02_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("r02", d["config"]["workload"], "step", d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"])
PY
  run r03 $D $w; run norag font-renderer_amd/libfr_raster_var_norag.so $w
done
