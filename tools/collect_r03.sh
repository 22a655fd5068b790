#!/bin/bash
# Run on the GPU box from the repo root (gpurun): round-3 evidence — one bench JSON per workload, rocprofv3 kernel-trace
# stats for the key ones, WRITE_SIZE / FETCH_SIZE passes (separate --pmc runs) and SQ counters.
# usage: tools/collect_r03.sh [tag] [part]   -> gpurun_out/prof_<tag>/   (copy what is to be judged into profiles/r03/)
#   part: bench | kt | pmc | all (default)
tag=${1:-r03}; part=${2:-all}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
ALL="c3_cjk21k_256px_s256_16spp c3_cjk21k_256px_s64_16spp c3_cjk21k_256px_s32_16spp c3_cjk21k_256px_s16_16spp c3_strokes21k_256px_s128_16spp \
 real_dejavuserif_italic_whole_font_256px_16spp real_dejavuserif_italic_whole_font_256px_gray_debug c4_bmp_shard_128px_s32_16spp c4_bmp_shard_128px_s32_gray_debug \
 c3_cjk21k_256px_s128_gray_debug c3_cjk21k_256px_s128_winding_i16 c3_cjk21k_256px_s128_4spp big_s512_2048cells_256px_16spp \
 real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_size64_16spp \
 real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp \
 c2_ascii95_128px_s32_16spp_x64pages"
KEY="c3_cjk21k_256px_s128_16spp c3_cjk21k_256px_s256_16spp c3_strokes21k_256px_s128_16spp c4_bmp_shard_128px_s32_16spp c2_ascii95_128px_s32_16spp_x64pages \
 real_dejavuserif_italic_whole_font_256px_16spp c3_cjk21k_256px_s128_gray_debug c5_sdf_shard_512px_s64 real_dejavuserif_italic_whole_font_512px_sdf \
 real_dejavuserif_italic_renderglyph_dims_size64_gray_debug real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp big_s512_2048cells_256px_16spp c3_cjk21k_256px_s128_4spp"
if [ $part = bench ] || [ $part = all ]; then
  python3 bench.py --steps 200 --warmup 100 > $out/c3_cjk21k_256px_s128_16spp_bench.json 2> $out/c3_bench.err || exit 1
  echo "c3 done"
  for w in $ALL; do
    python3 bench.py --workload $w --steps 200 --warmup 100 --cpu-seconds 3 > $out/${w}_bench.json 2>/dev/null
    echo "$w done"
  done
  python3 bench.py --workload c2_ascii95_128px_s32_16spp --steps 2000 --warmup 500 --cpu-seconds 3 > $out/c2_ascii95_128px_s32_16spp_bench.json 2>/dev/null
  python3 bench.py --workload c2_ascii95_real_128px_16spp --steps 2000 --warmup 500 --cpu-seconds 3 > $out/c2_ascii95_real_128px_16spp_bench.json 2>/dev/null
  python3 bench.py --workload c5_sdf_shard_512px_s64 --steps 100 --warmup 50 --cpu-seconds 6 > $out/c5_sdf_shard_512px_s64_bench.json 2>/dev/null
  python3 bench.py --workload real_dejavuserif_italic_whole_font_512px_sdf --steps 30 --warmup 10 --cpu-seconds 6 > $out/real_dejavuserif_italic_whole_font_512px_sdf_bench.json 2>/dev/null
  python3 tools/c1_latency.py $GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so > $out/c1_latency.txt 2>/dev/null
  echo "bench lines done"
fi
cd /tmp && export TMPDIR=/tmp
if [ $part = kt ] || [ $part = all ]; then
  for w in $KEY; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
    cp $out/kt_$w/kt_kernel_stats.csv $out/${w}_kernel_stats.csv 2>/dev/null
    rm -rf $out/kt_$w
    echo "kt $w done"
  done
fi
if [ $part = pmc ] || [ $part = all ]; then
  for w in c3_cjk21k_256px_s128_16spp c5_sdf_shard_512px_s64 real_dejavuserif_italic_whole_font_512px_sdf c3_cjk21k_256px_s128_gray_debug c4_bmp_shard_128px_s32_16spp; do
    for c in WRITE_SIZE FETCH_SIZE; do
      rocprofv3 --output-format csv --pmc $c -d $out/pmc_${w}_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    done
    echo "pmc $w done"
  done
  for w in c3_cjk21k_256px_s128_16spp c4_bmp_shard_128px_s32_16spp big_s512_2048cells_256px_16spp; do
    rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc_${w}_SQ -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    echo "sq $w done"
  done
  cd $GRAFT_REPO_ROOT
  python3 - <<PY > $out/pmc_summary.txt
import csv, collections, glob, os
print("rocprofv3 --pmc (separate passes), mean per dispatch and kernel")
for d in sorted(glob.glob("$out/pmc_*")):
    f = os.path.join(d, "pmc_counter_collection.csv")
    if not os.path.exists(f): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in acc.items():
        if not any(t in k for t in ("cov4", "win1", "render_kernel", "sdf_kernel", "prepare")): continue
        for c,v in sorted(cs.items()): print(f"{os.path.basename(d):62s} {k:70s} {c:22s} {sum(v)/len(v):.6g} x{len(v)}")
PY
  cat $out/pmc_summary.txt
  rm -rf $out/pmc_*/
fi
cd $GRAFT_REPO_ROOT
python3 tools/show_bench.py $out/*_bench.json
