#!/bin/bash
# same-box A/B of the shipped library (main) against a variant library (BASE=name: libfr_raster_var_<name>.so)
out=$GRAFT_REPO_ROOT/gpurun_out/dyn; mkdir -p $out
cd $GRAFT_REPO_ROOT
WL=${WL:-"c3_cjk21k_256px_s128_16spp c4_bmp_shard_128px_s32_16spp c3_cjk21k_256px_s32_16spp c3_strokes21k_256px_s128_16spp real_dejavuserif_italic_whole_font_256px_16spp \
  real_dejavuserif_italic_whole_font_256px_gray_debug c3_cjk21k_256px_s128_gray_debug c4_bmp_shard_128px_s32_gray_debug c2_ascii95_128px_s32_16spp_x64pages \
  real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug c3_cjk21k_256px_s256_16spp c5_sdf_shard_512px_s64"}
for w in $WL; do
  for rep in 1 2; do
  for v in main ${BASE:-dyn0}; do
    lib=font-renderer_amd/libfr_raster_var_$v.so; [ $v = main ] && lib=font-renderer_amd/libfr_raster.so
    FR_RASTER_LIB=$lib timeout -k 10 100 python bench.py --workload $w --steps 200 --warmup 50 --no-cpu-baseline > $out/${v}_${rep}_$w.json 2>/dev/null
    python tools/show_bench.py $out/${v}_${rep}_$w.json | head -1 | sed "s/^/$v /"
  done
  done
done
