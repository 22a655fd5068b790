#!/bin/bash
out=gpurun_out/r3h; mkdir -p $out
run() { name=$1; lib=$2; w=$3; shift 3
  FR_RASTER_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 30 --no-cpu-baseline "$@" > $out/${name}_$w.json 2> $out/${name}_$w.err
  python tools/show_bench.py $out/${name}_$w.json | head -1 | sed "s/^/$name /"; }
D=font-renderer_amd/libfr_raster.so
for w in real_dejavuserif_italic_whole_font_256px_gray_debug c4_bmp_shard_128px_s32_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug; do
  run w1occ6 $D $w; run w1occ4 font-renderer_amd/libfr_raster_var_w1occ4.so $w
done
for w in c3_cjk21k_256px_s128_16spp c3_cjk21k_256px_s32_16spp c3_cjk21k_256px_s16_16spp c3_cjk21k_256px_s64_16spp real_dejavuserif_italic_whole_font_256px_16spp c4_bmp_shard_128px_s32_16spp c3_cjk21k_256px_s256_16spp c3_cjk21k_256px_s128_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_16spp; do
  run new $D $w
done
