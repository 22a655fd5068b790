#!/bin/bash
# Run on the GPU box from the repo root (gpurun): round-2 evidence — one bench JSON per workload, rocprofv3
# kernel-trace stats for the key ones, WRITE_SIZE / FETCH_SIZE / SQ counter passes for the headline kernel.
# usage: tools/collect_r02.sh [tag]     -> gpurun_out/prof_<tag>/   (copy what is to be judged into profiles/r02/)
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 200 > $out/c3_bench.json 2> $out/c3_bench.err || exit 1
echo "c3 done"
for w in c3_cjk21k_256px_s256_16spp c3_cjk21k_256px_s64_16spp c3_cjk21k_256px_s32_16spp c3_cjk21k_256px_s16_16spp c3_strokes21k_256px_s128_16spp \
         real_dejavuserif_italic_whole_font_256px_16spp real_dejavuserif_italic_whole_font_256px_gray_debug c4_bmp_shard_128px_s32_16spp c3_cjk21k_256px_s128_gray_debug c3_cjk21k_256px_s128_winding_i16; do
  python3 bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > $out/${w}_bench.json 2>/dev/null
  echo "$w done"
done
python3 bench.py --workload c2_ascii95_128px_s32_16spp --steps 2000 --warmup 500 --no-cpu-baseline > $out/c2_ascii95_128px_s32_16spp_bench.json 2>/dev/null
python3 bench.py --workload c2_ascii95_real_128px_16spp --steps 2000 --warmup 500 --cpu-seconds 4 > $out/c2_ascii95_real_128px_16spp_bench.json 2>/dev/null
python3 bench.py --workload c5_sdf_shard_512px_s64 --steps 100 --warmup 50 --cpu-seconds 8 > $out/c5_sdf_shard_512px_s64_bench.json 2>/dev/null
python3 tools/exact_bench.py > $out/exact_lattice.json 2>/dev/null
python3 tools/c1_latency.py $GRAFT_REPO_ROOT/font-renderer_amd/libfr_raster.so > $out/c1_latency.txt 2>/dev/null
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
for w in c3_cjk21k_256px_s128_16spp c3_cjk21k_256px_s256_16spp c3_strokes21k_256px_s128_16spp real_dejavuserif_italic_whole_font_256px_16spp real_dejavuserif_italic_whole_font_256px_gray_debug \
         c4_bmp_shard_128px_s32_16spp c2_ascii95_128px_s32_16spp c5_sdf_shard_512px_s64 c3_cjk21k_256px_s128_gray_debug c3_cjk21k_256px_s128_winding_i16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 200 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
  cp $out/kt_$w/kt_kernel_stats.csv $out/${w}_kernel_stats.csv 2>/dev/null
  echo "kt $w done"
done
for w in c3_cjk21k_256px_s128_16spp c5_sdf_shard_512px_s64 c3_cjk21k_256px_s128_winding_i16 c3_cjk21k_256px_s128_gray_debug; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d $out/pmc_${w}_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  done
done
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc_c3_SQ -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
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
        for c,v in sorted(cs.items()): print(f"{os.path.basename(d):55s} {k:70s} {c:22s} {sum(v)/len(v):.6g}")
PY
cat $out/pmc_summary.txt
for f in $out/*_bench.json; do python3 - "$f" <<PY
import json,sys
d=json.load(open(sys.argv[1]))
print(f"{d['config']['workload']:52s} {d['ms_per_step']:9.4f} ms/step {d['gpixel_per_s']:9.1f} Gpx/s kernel {d['roofline']['kernel_ms']:8.4f} ms frac {d['roofline']['frac']:.4f}  {d['roofline']['kernel']}")
PY
done
