#!/bin/bash
# copies the summaries tools/profile_round.sh left under gpurun_out/r04p into profiles/r04 (tracked): every file with the version tag
# given as $1, and the HBM counter summaries also without a tag (bench.py reads profiles/r04/<workload>_hbm_counters.json for
# roofline.traffic)
set -e
tag=${1:?version tag, e.g. v2}
S=gpurun_out/r04p; D=profiles/r04
mkdir -p $D
for f in $S/bench_*.json $S/*_kernel_stats.csv $S/*_hbm_counters.json $S/*_sq_counters.json; do
  [ -s "$f" ] || continue
  b=$(basename "$f"); cp "$f" "$D/${b%.*}_$tag.${b##*.}"
done
for f in $S/*_hbm_counters.json; do [ -s "$f" ] && cp "$f" $D/; done
[ -s $S/wg_region_cycles.txt ] && grep -v amdgpu.ids $S/wg_region_cycles.txt > $D/wg_region_cycles_$tag.txt
[ -s $S/strip_probe.json ] && cp $S/strip_probe.json $D/strip_probe_$tag.json
ls $D
