"""Streaming LD loader (vilma_amd/ld_device.py) on the on-disk schema of cli_scale_check.py: per cohort the
wall time of stream_cohort, how long the main thread waited for the host pool, how long it spent on
uploads + device work, how many blocks rocSOLVER took, against the host pool alone (materialize).
    python profiles/cli_scale_check.py --blocks 400 --only-write && python profiles/load_timing.py /tmp/vilma_cli_check"""
import os, sys, time, logging
sys.path.insert(0, os.getcwd())
import numpy as np
logging.basicConfig(level=logging.INFO)
from vilma_amd import load
from vilma_amd.engine import HipEngine
from vilma_amd import ld_device
out = sys.argv[1]
variants = load.load_variant_list(os.path.join(out, 'extract.tsv'))
logging.getLogger().setLevel(logging.WARNING)
for p in range(2):
    t0 = time.perf_counter()
    ld, _ = load.load_ld_from_schema(os.path.join(out, 'c%d.schema' % p), variants, [], 0.9, lazy=True)
    t1 = time.perf_counter()
    N = ld.shape[0]
    n_ld = int(ld.starts[-1])
    eng = HipEngine(1, N, 2, 1)
    z = np.random.default_rng(0).normal(size=n_ld)
    import os
    for workers in [int(v) for v in os.environ.get('LOAD_TIMING_WORKERS', '16,8').split(',')]:
        ld2, _ = load.load_ld_from_schema(os.path.join(out, 'c%d.schema' % p), variants, [], 0.9, lazy=True)
        t2 = time.perf_counter()
        o = ld_device.stream_cohort(eng, 0, ld2, 'auto', z, workers=workers)
        t3 = time.perf_counter()
        print('cohort %d workers %d: schema %.2f s, stream %.2f s (GPU pass %.2f s for %d of %d blocks in stacks; then waited for the host pool %.2f, uploads + device work %.2f), max n %d'
              % (p, workers, t1 - t0, t3 - t2, o.get('gpu_eigh_s', 0.0), o['gpu_eigh'], len(ld2.matrices), o['wait_s'], o['device_s'], max(m.shape[0] for m in ld2.matrices)))
    # pure host eigh pool time for reference
    ld3, _ = load.load_ld_from_schema(os.path.join(out, 'c%d.schema' % p), variants, [], 0.9, lazy=True)
    t4 = time.perf_counter(); ld3.materialize(workers=16); t5 = time.perf_counter()
    print('cohort %d: host materialize alone (16 workers) %.2f s' % (p, t5 - t4))
    eng.close()
