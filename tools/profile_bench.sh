#!/bin/bash
# rocprofv3 passes over the headline bench command (run on the GPU box through gpurun): kernel trace + stats, then the PMC counters,
# one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in their own).  Summaries land in gpurun_out/$1/summary_*.
# usage: tools/profile_bench.sh <outdir-under-gpurun_out> [bench args]
set -uo pipefail
OUT=gpurun_out/${1:-prof}; shift || true
ARGS="--no-cpu --no-legs --no-e2e --steps 3 --warmup 1 $*"
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT \
    -d $OUT/sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
python3 bench.py --no-cpu --no-legs --no-e2e > $OUT/bench_plain.json 2> $OUT/plain.err
# the other BASELINE configurations (bench.py's legs): kernel trace only
rocprofv3 --kernel-trace --stats -d $OUT/legs -- python3 bench.py --no-cpu --no-e2e --legs-only > $OUT/bench_legs_trace.json 2> $OUT/legs.err
LG=$(find $OUT/legs -name "*.db" | head -1)
python3 - "$LG" $OUT/summary_legs_kernel_stats.csv <<'PY'
import csv, sys
sys.path.insert(0, "tools")
import prof_summary as ps
w = csv.writer(open(sys.argv[2], "w", newline="")); w.writerow(["kernel", "calls_with_work", "total_us", "avg_us", "min_us", "max_us", "empty_calls_left_out"])
for r in ps.kernel_stats(sys.argv[1]): w.writerow([r[0], r[1]] + ["%.3f" % x for x in r[2:6]] + [r[6]])
PY
T=$(find $OUT/trace -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1); S=$(find $OUT/sq -name "*.db" | head -1)
echo "dbs: $T $F $W $S"
python3 tools/prof_summary.py "$T" "$F" "$W" $OUT/summary "$OUT/bench_trace.json" "$S"
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq $OUT/legs   # the databases are large; the summaries are what is kept
ls -la $OUT
