#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out
run() { name=$1; lib=$2; w=$3; shift 3
  FR_RASTER_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 30 --no-cpu-baseline "$@" > $out/${name}_$w.json 2> $out/${name}_$w.err
  python tools/show_bench.py $out/${name}_$w.json | head -1 | sed "s/^/$name /"; }
D=font-renderer_amd/libfr_raster.so
for w in c3_cjk21k_256px_s32_16spp c3_cjk21k_256px_s16_16spp real_dejavuserif_italic_whole_font_256px_16spp c3_cjk21k_256px_s64_16spp; do
  run base $D $w; run nowd font-renderer_amd/libfr_raster_var_nowd.so $w
done
for w in real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_16spp real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp; do
  run ovl1 $D $w; run ovl0 $D $w --opt overlap=0
done
