#!/bin/bash
# option "graph" (a plan's launches replayed as one hipGraph) against plain launches, same box
out=$GRAFT_REPO_ROOT/gpurun_out/graph; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_16spp \
  real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp \
  c2_ascii95_128px_s32_16spp real_dejavuserif_italic_whole_font_256px_16spp c5_sdf_shard_512px_s64 c3_cjk21k_256px_s128_16spp; do
  for rep in 1 2; do
  for v in graph=0 graph=1; do
    timeout -k 10 100 python bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline --opt $v > $out/${v}_${rep}_$w.json 2>$out/err_${v}_$w.txt
    python tools/show_bench.py $out/${v}_${rep}_$w.json | head -1 | sed "s/^/$v /"
  done
  done
done
