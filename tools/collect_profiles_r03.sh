#!/bin/bash
# rocprofv3 evidence under profiles/r03 (run on the GPU box from the repo root):  bash tools/collect_profiles_r03.sh gpurun_out/prof3
#  bench passes      as tools/collect_profiles.sh (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters)
#  hbm-scale passes  tools/hbm_scale_probe.py at working sets beyond the 256 MiB Infinity Cache
#  interval passes   tools/interval_trace.py: kernel trace and SQ counters per observation interval of every
#                    benchmark configuration (the propagation kernel against its VALU-issue ceiling, interval by interval)
# Every pass is its own process under its own time limit; counters are never combined with sys/hip/hsa traces.
set -e -o pipefail
OUT=${1:-gpurun_out/prof3}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT/hbm" "$OUT/interval"
[ -n "$SKIP_BENCH" ] || bash tools/collect_profiles.sh "$OUT/bench"
if [ -z "$SKIP_HBM" ]; then
# one set of passes per configuration: two configurations can launch a kernel with the same grid (64 filters x 2^18
# particles at 4 tiles per workgroup and 64 x 2^16 at one are both 64 x 64 workgroups of k_offsets)
for C in "sis_16x2^20" "seir_64x2^18" "seir_64x2^18+incremental" "seir_64x2^16_config4_share" "seir_64x2^16_config4_share+incremental"; do
D="$OUT/hbm/$(echo "$C" | tr '^+' 'p_')"
mkdir -p "$D"
P="python3 tools/hbm_scale_probe.py 6 $C"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- $P > "$D/probe_under_rocprof.json" 2> "$D/trace.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D/pmc_fetch" -- $P > /dev/null 2> "$D/pmc_fetch.err"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$D/pmc_write" -- $P > /dev/null 2> "$D/pmc_write.err"
python3 tools/summarise_profiles.py "$D" "$OUT/hbm/summary/$(basename "$D")"
cp "$D/probe_under_rocprof.json" "$OUT/hbm/summary/$(basename "$D")/"
done
fi
if [ -n "$SKIP_INTERVAL" ]; then echo "done (bench/hbm only)"; exit 0; fi
I="python3 tools/interval_trace.py run"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/interval/trace" -- $I "$OUT/interval/plan.json" > "$OUT/interval/trace.log" 2>&1
python3 tools/interval_trace.py parse "$OUT/interval/trace" "$OUT/interval/plan.json" > "$OUT/interval/interval_trace.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/interval/pmc/sq1" -- $I "$OUT/interval/plan_sq.json" > "$OUT/interval/sq1.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/interval/pmc/sq2" -- $I "$OUT/interval/plan_sq.json" > "$OUT/interval/sq2.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/interval/pmc/sq3" -- $I "$OUT/interval/plan_sq.json" > "$OUT/interval/sq3.log" 2>&1
python3 tools/interval_trace.py parse_pmc "$OUT/interval/pmc" "$OUT/interval/plan_sq.json" > "$OUT/interval/interval_sq_counters.json"
echo "done: $OUT/bench/summary, $OUT/hbm/summary, $OUT/interval"
