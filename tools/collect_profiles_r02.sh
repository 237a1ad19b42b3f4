#!/bin/bash
# rocprofv3 evidence under profiles/r02 (run on the GPU box from the repo root):  bash tools/collect_profiles_r02.sh gpurun_out/prof2
#  bench passes      as tools/collect_profiles.sh (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters)
#  hbm-scale passes  tools/hbm_scale_probe.py at working sets beyond the 256 MiB Infinity Cache: kernel trace + stats,
#                    FETCH_SIZE and WRITE_SIZE in runs of their own
# Every pass is its own process under its own time limit; counters are never combined with sys/hip/hsa traces.
set -e -o pipefail
OUT=${1:-gpurun_out/prof2}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT/hbm"
[ -n "$SKIP_BENCH" ] || bash tools/collect_profiles.sh "$OUT/bench"
P="python3 tools/hbm_scale_probe.py 4 sis_16x2^20 seir_64x2^18 seir_64x2^18+incremental seir_64x2^16_config4_share seir_64x2^16_config4_share+incremental"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hbm/trace" -- $P > "$OUT/hbm/probe_under_rocprof.json" 2> "$OUT/hbm/trace.err"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/hbm/pmc_fetch" -- $P > /dev/null 2> "$OUT/hbm/pmc_fetch.err"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/hbm/pmc_write" -- $P > /dev/null 2> "$OUT/hbm/pmc_write.err"
python3 tools/summarise_profiles.py "$OUT/hbm" "$OUT/hbm/summary"
echo "done: $OUT/bench/summary and $OUT/hbm/summary"
