#!/bin/bash
# Collects the measurements committed under profiles/r04 (run on the GPU box from the repo root):
#   bench lines per workload, rocprofv3 kernel statistics of the --loop-only command, PMC passes (one counter per pass)
set -o pipefail
O=gpurun_out/r04p
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-configs > $O/bench_benchmark4.json 2> $O/bench_benchmark4.err
python3 bench.py --workload s10k --steps 100 --warmup 10 > $O/bench_s10k.json 2> $O/bench_s10k.err
python3 bench.py --workload s100k --steps 40 --warmup 5 > $O/bench_s100k.json 2> $O/bench_s100k.err
python3 bench.py --workload s6d --steps 10 --warmup 3 > $O/bench_s6d.json 2> $O/bench_s6d.err
python3 bench.py --program wavefront --no-cpu --no-configs > $O/bench_benchmark4_wavefront.json 2> $O/bench_benchmark4_wavefront.err
python3 bench.py --cold-start --no-cpu --no-configs > $O/bench_benchmark4_cold.json 2> $O/bench_benchmark4_cold.err
for wl in benchmark4 s10k s6d s100k; do
  steps=100; [ $wl = s6d ] && steps=10; [ $wl = s100k ] && steps=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --loop-only --workload $wl --steps $steps --warmup 5 > $O/prof_$wl.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${wl}_$c -- python3 bench.py --loop-only --workload $wl --steps 10 --warmup 2 > $O/pmc_${wl}_$c.log 2>&1
  done
  python3 tools/pmc_summary.py $O/${wl}_hbm_counters.json FETCH_SIZE=$O/pmc_${wl}_FETCH_SIZE WRITE_SIZE=$O/pmc_${wl}_WRITE_SIZE > /dev/null
  find $O/prof_$wl -name "*kernel_stats.csv" -exec cp {} $O/${wl}_kernel_stats.csv \;
done
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_benchmark4_$c -- python3 bench.py --loop-only --steps 10 --warmup 2 > $O/pmc_benchmark4_$c.log 2>&1
done
python3 tools/pmc_summary.py $O/benchmark4_sq_counters.json SQ_WAVES=$O/pmc_benchmark4_SQ_WAVES SQ_INSTS_VALU=$O/pmc_benchmark4_SQ_INSTS_VALU SQ_INSTS_LDS=$O/pmc_benchmark4_SQ_INSTS_LDS SQ_BUSY_CYCLES=$O/pmc_benchmark4_SQ_BUSY_CYCLES > /dev/null
[ -f gcs_admm_amd/libgcsadmm_timing.so ] && python3 tools/wg_phase_timing.py benchmark4 > $O/wg_region_cycles.txt 2>&1
python3 tools/strip_probe.py > $O/strip_probe.json 2> $O/strip_probe.err
# keep the merge-back small: raw traces are not needed
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O | head -50
