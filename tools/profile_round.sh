#!/usr/bin/env bash
# The round's rocprofv3 evidence, run ON THE GPU BOX (gpurun): kernel trace of the default bench run (eager launches, side stream on)
# and four PMC passes of a short run with the leaves inline (one kernel at a time).  Summaries are written by tools/rocpd_stats.py,
# tools/pmc_traffic.py and tools/pmc_lds.py into gpurun_out/profiles_rNN/ -- copy what is to be judged into profiles/.
#   usage: bash tools/profile_round.sh r04
set -uo pipefail
R="${1:-r04}"
OUT="gpurun_out/profiles_$R"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_SHORT="python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary"
# 1. kernel trace of the bench's own timed run (20 steps, eager, side stream on)
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o p -- python bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary > "$OUT/trace.log" 2>&1
DB=$(ls "$OUT"/trace/*/p_results.db "$OUT"/trace/p_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" "$OUT/${R}_kernel_stats_final.csv" > "$OUT/stats.log" 2>&1; fi
# 2-5. PMC passes (separate runs; --kernel-trace only, as the guide prescribes)
for C in FETCH_SIZE WRITE_SIZE; do
  CRUSE_OVERLAP=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- $BENCH_SHORT > "$OUT/pmc_$C.log" 2>&1
done
CRUSE_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o p -- $BENCH_SHORT > "$OUT/pmc_mfma.log" 2>&1
CRUSE_OVERLAP=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o p -- $BENCH_SHORT > "$OUT/pmc_lds.log" 2>&1
f() { ls "$OUT/$1"/*/p_counter_collection.csv "$OUT/$1"/p_counter_collection.csv 2>/dev/null | head -1; }
python tools/pmc_traffic.py "$(f pmc_FETCH_SIZE)" "$(f pmc_WRITE_SIZE)" "$OUT/${R}_pmc_hbm_traffic.csv" > "$OUT/traffic.log" 2>&1
python tools/pmc_traffic.py --mfma "$(f pmc_mfma)" "$OUT/${R}_pmc_mfma_util.csv" > "$OUT/mfma.log" 2>&1
python tools/pmc_lds.py "$(f pmc_lds)" "$OUT/${R}_pmc_lds_conflicts.csv" > "$OUT/lds.log" 2>&1
# keep the merge-back small: drop the raw traces, keep summaries and logs
rm -rf "$OUT"/trace "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_mfma "$OUT"/pmc_lds
tail -25 "$OUT/traffic.log"; tail -12 "$OUT/mfma.log"; head -12 "$OUT/${R}_kernel_stats_final.csv"; head -8 "$OUT/${R}_pmc_lds_conflicts.csv"
