"""A/B of library builds on ONE box: bench.py runs alternating between builds, one compact line each.

    python profiles/ab_bench.py --libs default,vilma_amd/libvilma_hip_t1.so \\
        --runs "--workload C3" "--workload C3K12 --steps 20" [--repeat 2]

Every run is a fresh process (`VILMA_HIP_LIB=<lib> python bench.py --no-cpu-baseline <args>`), so
each build gets its own allocations; builds alternate inside a workload so that box-level drift hits
all of them alike.  Variant libraries come from vilma_amd.build.build_library(extra_flags=..., out=...)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--libs', required=True, help='comma-separated: "default" or a path to a variant .so')
    ap.add_argument('--runs', nargs='+', required=True, help='bench.py argument strings')
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--envs', nargs='*', default=[''],
                    help='environment variants alternated like the libraries: "NAME=VALUE[ NAME2=VALUE2]" ("" = none)')
    a = ap.parse_args()
    libs = a.libs.split(',')
    for run in a.runs:
        for _ in range(a.repeat):
            for lib, ev in [(l, e) for l in libs for e in a.envs]:
                env = dict(os.environ)
                env.pop('VILMA_HIP_LIB', None)
                for kv in ev.split():
                    env[kv.split('=', 1)[0]] = kv.split('=', 1)[1]
                if lib != 'default':
                    env['VILMA_HIP_LIB'] = os.path.join(ROOT, lib)
                cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline'] + run.split()
                r = subprocess.run(cmd, env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith('{')]
                if r.returncode or not line:
                    print('%-34s %s | FAILED rc %d: %s' % (lib + ' ' + ev, run, r.returncode, r.stderr[-300:]), flush=True)
                    continue
                j = json.loads(line[-1])
                rf = j.get('roofline', {})
                ks = {k['name']: round(k['avg_ms'], 4) for k in rf.get('kernels', [])}
                print('%-34s %s | %.1f sweeps/s %.4f ms | %s %.4f ms frac %.3f | %s'
                      % (lib + ' ' + ev, run, j['value'], j['ms_per_step'], rf.get('kernel'), rf.get('avg_launch_ms', 0.0),
                         rf.get('frac', 0.0), ks), flush=True)


if __name__ == '__main__':
    main()
