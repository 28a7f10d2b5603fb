#!/bin/bash
# One profile set for profiles/ (run on the GPU box from the repo root):
#     bash profiles/run_profile_set.sh r01f
# 1) rocprofv3 --kernel-trace --stats of bench.py (C3, 1 GPU), reduced by summarize_rocprof.py +
#    timeline_gaps.py; 2) and 3) PMC passes FETCH_SIZE / WRITE_SIZE on their own (kernel trace
#    only, as the pool requires); 4) the default un-profiled `python3 bench.py` line.
# Raw rocprofv3 output stays in /tmp (hundreds of MB); summaries go to gpurun_out/sum/<prefix>_*.
set -o pipefail
PFX=${1:-prof}
WL=${2:-C3}
FORM=${3:-auto}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sum
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- \
    python3 "$ROOT/bench.py" --workload $WL --ld-form $FORM --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/${PFX}_bench_under_rocprof.json" 2>/dev/null || exit 1
python3 "$ROOT/profiles/summarize_rocprof.py" /tmp/rp_stats "$OUT/${PFX}_stats" || exit 1
python3 "$ROOT/profiles/timeline_gaps.py" /tmp/rp_stats > "$OUT/${PFX}_timeline.txt" 2>&1
rm -rf /tmp/rp_stats
echo "[profile] stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/rp_$(echo $ctr | tr A-Z a-z | cut -d_ -f1)
    # counters only for the library's kernels: the synthetic setup issues ~100 k torch dispatches
    # that a counter pass would serialise one by one
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "ld_tile|ld_sym|ld_eig|ld_colsum|ld_rowsum|snp_pass|delta_kernel|finalize|sweep_decide|tile_sums" --output-format csv -d $d -- \
        python3 "$ROOT/bench.py" --workload $WL --ld-form $FORM --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
    python3 "$ROOT/profiles/summarize_rocprof.py" $d "$OUT/${PFX}_$(echo $ctr | tr A-Z a-z | cut -d_ -f1)" || exit 1
    rm -rf $d
    echo "[profile] $ctr done"
done
cd "$ROOT"
# HBM traffic per launch of the LD-streaming kernels from the two PMC passes, as the guide
# prescribes for gfx950: (2 x FETCH_SIZE + WRITE_SIZE) KiB; stamped with the kernel source hash
# so bench.py only replays it for the build it was taken on
python3 - "$OUT/${PFX}_fetch_pmc.json" "$OUT/${PFX}_write_pmc.json" "$OUT/${PFX}_traffic.json" $WL $FORM <<'PY'
import hashlib, json, sys
f, w = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
out = {'source': 'profiles/%s + %s (rocprofv3 --pmc, separate passes)' % tuple(a.split('/')[-1] for a in sys.argv[1:3]),
       'formula': '(2*FETCH_SIZE + WRITE_SIZE)*1024 over the full-product launches',
       'workload': sys.argv[4], 'ld_form': sys.argv[5],
       'kernels_hip_sha16': hashlib.sha256(open('vilma_amd/csrc/kernels.hip', 'rb').read()).hexdigest()[:16]}
for k in ('ld_tile_kernel', 'ld_sym_kernel', 'ld_eig_fused_kernel'):
    if k in f and k in w:
        out[k + '_bytes_per_launch'] = (2 * f[k]['FETCH_SIZE']['mean_over_large_dispatches']
                                        + w[k]['WRITE_SIZE']['mean_over_large_dispatches']) * 1024
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))
PY
timeout -k 10 400 python3 bench.py --workload $WL --ld-form $FORM > "$OUT/${PFX}_bench.json" 2> "$OUT/${PFX}_bench.err" || exit 1
tail -c 2500 "$OUT/${PFX}_bench.json"
cat "$OUT/${PFX}_timeline.txt"
