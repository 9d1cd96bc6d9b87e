#!/bin/bash
# Round profiles (run through gpurun from the repo root):  bash tools/collect_profiles.sh r03 [stats|pmc]
# (--no-roofline: the isolated-kernel timing leg would mix its back-to-back launches into the in-sweep averages)
# stats: rocprofv3 kernel trace + stats of a SHORT bench run (32 lanes, 32 new tokens per image: the per-kernel durations
#        do not depend on the number of tokens; the full default run produces millions of trace records);
# pmc:   three separate --pmc passes (one counter each, never combined with other trace domains) over 3 tokens of 16 lanes (one rider ring).
# Raw output is deleted after the summaries are extracted (gpurun_out/ is capped at 64 MiB).
R=${1:-r02}
WHAT=${2:-stats}
export TMPDIR=/tmp
OUT=gpurun_out
# Build BEFORE any profiled process starts: under rocprofv3 the GPU is initialised by the profiler's preload, and a bench.py that found the tree
# stale would spawn hipcc (an exec from a GPU-initialised process: refused / fatal on this pool).  DD_NO_BUILD makes build.build() raise instead.
python3 -m dropoutdecoding_amd.build > $OUT/${R}_build.log 2>&1 || { echo "build failed"; tail -5 $OUT/${R}_build.log; exit 1; }
export DD_NO_BUILD=1
# the sources the per-kernel figures belong to (bench.py: roofline.profile_matches_tree)
python3 -c "
import json, sys; sys.path.insert(0, '.')
import bench
json.dump({'decode_sources_sha256': bench.decode_sources_hash(), 'files': bench.DECODE_SOURCES}, open('$OUT/${R}_sources.json', 'w'), indent=1)"
if [ "$WHAT" = stats ]; then
  rm -rf /tmp/${R}_stats
  timeout 1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${R}_stats -- python3 bench.py --steps 1 --warmup 0 --n-new 32 --no-cpu-baseline --no-roofline --single-images 1 > $OUT/${R}_bench_under_rocprof.log 2>&1
  echo "rocprof rc=$?"
  f=$(find /tmp/${R}_stats -name "*kernel_stats.csv" | head -1)
  head -70 "$f" > $OUT/${R}_kernel_stats.csv
  rm -rf /tmp/${R}_stats
  tail -1 $OUT/${R}_bench_under_rocprof.log | cut -c1-200
  head -32 $OUT/${R}_kernel_stats.csv | cut -c1-160
elif [ "$WHAT" = stats5 ]; then
  # BASELINE config 5 (LLaVA-NeXT-Mistral-7B shapes, fp8 matrices): one batch of 64 images, 32 new tokens each
  rm -rf /tmp/${R}_stats5
  timeout 1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${R}_stats5 -- python3 bench.py --config 5 --steps 1 --warmup 0 --n-new 32 --no-cpu-baseline --no-roofline --single-images 0 > $OUT/${R}_c5_bench_under_rocprof.log 2>&1
  echo "rocprof rc=$?"
  f=$(find /tmp/${R}_stats5 -name "*kernel_stats.csv" | head -1)
  head -60 "$f" > $OUT/${R}_c5_kernel_stats.csv
  rm -rf /tmp/${R}_stats5
  head -24 $OUT/${R}_c5_kernel_stats.csv | cut -c1-160
elif [ "$WHAT" = pmc5 ]; then
  for c in FETCH_SIZE:fetch WRITE_SIZE:write; do
    rm -rf /tmp/${R}_pmc5_${c##*:}
    timeout 500 rocprofv3 --pmc ${c%%:*} --kernel-trace --output-format csv -d /tmp/${R}_pmc5_${c##*:} -- python3 bench.py --config 5 --steps 1 --warmup 0 --n-new 3 --images-per-gpu 16 --no-cpu-baseline --no-roofline --single-images 0 > $OUT/${R}_c5_pmc_${c##*:}.log 2>&1
    echo "pmc ${c%%:*} rc=$?"
  done
  python3 tools/pmc_summary.py /tmp/${R}_pmc5_fetch /tmp/${R}_pmc5_write > $OUT/${R}_c5_pmc_summary.json
  rm -rf /tmp/${R}_pmc5_fetch /tmp/${R}_pmc5_write
else
  for c in FETCH_SIZE:fetch WRITE_SIZE:write SQ_VALU_MFMA_BUSY_CYCLES:mfma; do
    rm -rf /tmp/${R}_pmc_${c##*:}
    timeout 500 rocprofv3 --pmc ${c%%:*} --kernel-trace --output-format csv -d /tmp/${R}_pmc_${c##*:} -- python3 bench.py --steps 1 --warmup 0 --n-new 3 --images-per-gpu 16 --no-cpu-baseline --no-roofline --single-images 0 > $OUT/${R}_pmc_${c##*:}.log 2>&1
    echo "pmc ${c%%:*} rc=$?"
  done
  python3 tools/pmc_summary.py /tmp/${R}_pmc_fetch /tmp/${R}_pmc_write /tmp/${R}_pmc_mfma > $OUT/${R}_pmc_summary.json
  rm -rf /tmp/${R}_pmc_fetch /tmp/${R}_pmc_write /tmp/${R}_pmc_mfma
  python3 -c "
import json; d=json.load(open('$OUT/${R}_pmc_summary.json'))
for k,v in list(d['kernels'].items())[:14]: print(k[:70], v.get('hbm_read_bytes_per_launch'), v.get('hbm_write_bytes_per_launch'), v.get('mfma_util'))"
fi
