#!/bin/bash
# Per-rank timeline of a K-way sharding, emulated on one GPU (rank 0's shard alone):
#     bash profiles/run_shard_timeline.sh r02a 8
# 1) un-profiled bench.py --emulate-shard K, without and with a forced one-rank RCCL group (every
#    decision pays a real all-reduce launch); 2) rocprofv3 --kernel-trace of the same, reduced by
#    timeline_gaps.py.  Summaries go to gpurun_out/sum/<prefix>_shardK_*.
set -o pipefail
PFX=${1:-prof}
K=${2:-8}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sum
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --emulate-shard $K --steps 40 --warmup 5 --no-cpu-baseline \
    > "$OUT/${PFX}_shard${K}_bench.json" 2> "$OUT/${PFX}_shard${K}_bench.err" || exit 1
VILMA_BENCH_FORCE_RCCL=1 timeout -k 10 300 python3 bench.py --emulate-shard $K --steps 40 --warmup 5 \
    --no-cpu-baseline > "$OUT/${PFX}_shard${K}_bench_rccl.json" 2> "$OUT/${PFX}_shard${K}_bench_rccl.err" || exit 1
cd /tmp
rm -rf /tmp/rp_shard
VILMA_BENCH_FORCE_RCCL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_shard -- \
    python3 "$ROOT/bench.py" --emulate-shard $K --steps 20 --warmup 3 --no-cpu-baseline \
    > "$OUT/${PFX}_shard${K}_bench_under_rocprof.json" 2>/dev/null || exit 1
python3 "$ROOT/profiles/timeline_gaps.py" /tmp/rp_shard > "$OUT/${PFX}_shard${K}_timeline.txt" 2>&1
python3 "$ROOT/profiles/sweep_sequence.py" /tmp/rp_shard 70 > "$OUT/${PFX}_shard${K}_sequence.txt" 2>&1
rm -rf /tmp/rp_shard
cd "$ROOT"
python3 - "$OUT/${PFX}_shard${K}_bench.json" "$OUT/${PFX}_shard${K}_bench_rccl.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], 'ms_per_step %.4f' % d['ms_per_step'], 'ld avg ms %.4f' % d['roofline']['avg_launch_ms'])
PY
cat "$OUT/${PFX}_shard${K}_timeline.txt"
cat "$OUT/${PFX}_shard${K}_sequence.txt"
