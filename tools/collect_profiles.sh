#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run from the repo root):
#   bash tools/collect_profiles.sh gpurun_out/prof
# Passes (each its own process; counters never combined with sys/hip/hsa traces):
#   trace        --kernel-trace --stats                 -> per-kernel durations
#   pmc_fetch    --pmc FETCH_SIZE                        -> HBM read traffic
#   pmc_write    --pmc WRITE_SIZE                        -> HBM write traffic
#   pmc_sq1/2/3  SQ counters of k_propagate (VALU busy, lane utilisation, clock)
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
B="python3 bench.py --no-cpu-baseline --no-smc2"   # (the headline workload only: the SMC^2 leg has kernels of its own)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $B --steps 10 --warmup 20 > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $B --steps 3 --warmup 5 > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $B --steps 3 --warmup 5 > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- $B --steps 3 --warmup 5 > /dev/null 2> "$OUT/pmc_sq1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- $B --steps 3 --warmup 5 > /dev/null 2> "$OUT/pmc_sq2.err"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_sq3" -- $B --steps 3 --warmup 5 > /dev/null 2> "$OUT/pmc_sq3.err"
python3 tools/summarise_profiles.py "$OUT" "$OUT/summary"
echo "profiles summarised under $OUT/summary"
