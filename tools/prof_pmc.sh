#!/bin/bash
# PMC passes over one bench step (run on the GPU box through gpurun).  One rocprofv3 run per counter
# set: the hardware holds 8 SQ / 4 TCC counters per pass and rocprofv3 does not split them itself.
#   usage: tools/prof_pmc.sh <tag> [extra bench.py args]
#          PROF_PROG="python3 tools/bench_configs.py cfg4 --reps 1" tools/prof_pmc.sh <tag>     (any other program of this repo)
set -e
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu $*"
TRACE="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu $*"
if [ -n "$PROF_PROG" ]; then BENCH="${PROF_PROG/tools\//$ROOT/tools/}"; TRACE="$BENCH"; fi
run() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
    echo "pass $name done"
}
# the timing pass runs warm (1 warm-up + 3 timed steps), like bench.py's own kernel_ms; the counter passes one cold step
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $TRACE > "$OUT/trace.log" 2>&1
echo "trace done"
if [ -z "$PROF_TRACE_ONLY" ]; then   # PROF_TRACE_ONLY=1: kernel times only (cfg3's 483,328-workgroup launches take minutes per counter pass)
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
fi
python3 $ROOT/tools/pmc_summary.py "$OUT" "$ROOT/gpurun_out/${TAG}_pmc.json" > "$ROOT/gpurun_out/${TAG}_summary.txt"
# rocprofv3's own --stats table, this library's kernels only (the rest is torch generating the synthetic PCM)
for f in "$OUT"/trace/*/*kernel_stats.csv; do { head -1 "$f"; grep -E '^"(void )?k_' "$f"; } > "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"; done 2>/dev/null || true
grep "^{" "$OUT/trace.log" | tail -1 > "$ROOT/gpurun_out/${TAG}_bench_line.json" 2>/dev/null || true
head -3 "$OUT"/sq1/*/*counter_collection.csv > "$ROOT/gpurun_out/${TAG}_csv_head.txt" 2>/dev/null || true
rm -rf "$OUT"
