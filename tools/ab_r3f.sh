#!/bin/bash
# A/B on the GPU box: waves per SIMD of the small-strip instances of win1_kernel (4 / 6 / 8) and cov4_kernel (4 / 6)
out=gpurun_out/r3f; mkdir -p $out
run() { name=$1; lib=$2; w=$3; shift 3
  FR_RASTER_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 30 --no-cpu-baseline "$@" > $out/${name}_$w.json 2> $out/${name}_$w.err
  python tools/show_bench.py $out/${name}_$w.json | head -1 | sed "s/^/$name /"; }
D=font-renderer_amd/libfr_raster.so
for w in c4_bmp_shard_128px_s32_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug; do
  run w1occ6 $D $w; run w1occ4 font-renderer_amd/libfr_raster_var_w1occ4.so $w; run w1occ8 font-renderer_amd/libfr_raster_var_w1occ8.so $w
done
for w in c4_bmp_shard_128px_s32_16spp real_dejavuserif_italic_renderglyph_dims_size64_16spp real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp c2_ascii95_128px_s32_16spp; do
  run c4occ6 $D $w; run c4occ4 font-renderer_amd/libfr_raster_var_c4occ4.so $w
done
