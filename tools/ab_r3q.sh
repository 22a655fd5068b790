#!/bin/bash
# cov4 phase ablations (timing-only builds) on four workloads + the 2-wave-workgroup experiment on the small-image batches
out=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in c3_cjk21k_256px_s128_16spp c3_strokes21k_256px_s128_16spp c3_cjk21k_256px_s256_16spp c4_bmp_shard_128px_s32_16spp; do
  WORKLOAD=$w tools/c4_ablate.sh main c4a1 c4a2 c4a4 c4a5 c4a6 > $out/c4_ablate_$w.txt 2>&1
  cat $out/c4_ablate_$w.txt
done
cd $GRAFT_REPO_ROOT
for w in real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp real_dejavuserif_italic_renderglyph_dims_size64_gray_debug c4_bmp_shard_128px_s32_16spp; do
  for v in main nw2; do
    lib=font-renderer_amd/libfr_raster_var_$v.so; [ $v = main ] && lib=font-renderer_amd/libfr_raster.so
    FR_RASTER_LIB=$lib timeout -k 10 100 python bench.py --workload $w --steps 100 --warmup 30 --no-cpu-baseline > $out/${v}_$w.json 2>/dev/null
    python tools/show_bench.py $out/${v}_$w.json | head -1 | sed "s/^/$v /"
  done
done
