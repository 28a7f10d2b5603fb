#!/bin/bash
# One profile set for profiles/ (run on the GPU box from the repo root):
#     bash profiles/run_profile_set.sh r01f
# 1) rocprofv3 --kernel-trace --stats of bench.py (C3, 1 GPU), reduced by summarize_rocprof.py +
#    timeline_gaps.py; 2) and 3) PMC passes FETCH_SIZE / WRITE_SIZE on their own (kernel trace
#    only, as the pool requires); 4) the default un-profiled `python3 bench.py` line.
# Raw rocprofv3 output stays in /tmp (hundreds of MB); summaries go to gpurun_out/sum/<prefix>_*.
set -o pipefail
PFX=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sum
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- \
    python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/${PFX}_bench_under_rocprof.json" 2>/dev/null || exit 1
python3 "$ROOT/profiles/summarize_rocprof.py" /tmp/rp_stats "$OUT/${PFX}_stats" || exit 1
python3 "$ROOT/profiles/timeline_gaps.py" /tmp/rp_stats > "$OUT/${PFX}_timeline.txt" 2>&1
rm -rf /tmp/rp_stats
echo "[profile] stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/rp_$(echo $ctr | tr A-Z a-z | cut -d_ -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
    python3 "$ROOT/profiles/summarize_rocprof.py" $d "$OUT/${PFX}_$(echo $ctr | tr A-Z a-z | cut -d_ -f1)" || exit 1
    rm -rf $d
    echo "[profile] $ctr done"
done
cd "$ROOT"
timeout -k 10 400 python3 bench.py > "$OUT/${PFX}_bench_c3.json" 2> "$OUT/${PFX}_bench_c3.err" || exit 1
tail -c 2500 "$OUT/${PFX}_bench_c3.json"
cat "$OUT/${PFX}_timeline.txt"
