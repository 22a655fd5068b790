#!/bin/bash
# A/B on the GPU box: occupancy of the 16-crossing instances (list rows CAP + 4 slots) and the dense gather of over-full rows.
# Variant libraries: make -C font-renderer_amd/csrc variant NAME=occ6 DEFS=-DFR_C4_OCC_SMALL=6 (4 / 8 likewise; FR_W1_OCC_SMALL for win1)
out=gpurun_out/r3e; mkdir -p $out
run() { # name lib workload extra...
  name=$1; lib=$2; w=$3; shift 3
  FR_RASTER_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 30 --no-cpu-baseline "$@" > $out/${name}_$w.json 2> $out/${name}_$w.err
  python tools/show_bench.py $out/${name}_$w.json | head -1 | sed "s/^/$name /"
}
D=font-renderer_amd/libfr_raster.so
V=font-renderer_amd/libfr_raster_var_occ6.so
for w in c4_bmp_shard_128px_s32_16spp c3_cjk21k_256px_s32_16spp c3_cjk21k_256px_s16_16spp c2_ascii95_128px_s32_16spp_x64pages real_dejavuserif_italic_whole_font_256px_16spp; do
  run occ5 $D $w
  run occ4 $D $w --opt lds_pad=8192
  run occ6 $V $w
done
for w in c3_cjk21k_256px_s128_16spp c3_cjk21k_256px_s256_16spp c3_strokes21k_256px_s128_16spp c3_cjk21k_256px_s64_16spp; do
  run base $D $w
done
