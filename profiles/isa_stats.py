"""Per-kernel register / spill / memory-instruction counts from a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --save-temps -c vilma_amd/csrc/kernels.hip
    python profiles/isa_stats.py kernels-hip-amdgcn-amd-amdhsa-gfx950.s [name filter ...]

Prints, per kernel whose demangled name contains every filter: VGPRs (+AGPRs), SGPRs, spilled
VGPRs, scratch bytes, LDS bytes, occupancy, and how many global loads / stores of each width the
body holds (static counts: a loop body counts once)."""
import re
import subprocess
import sys
from collections import Counter


def main():
    path, filt = sys.argv[1], sys.argv[2:]
    text = open(path).read()
    # kernel bodies: from "<name>:" up to ".end_amdhsa_kernel"
    out = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?; Occupancy: \d+)', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        if not all(f in dem for f in filt):
            continue
        def meta(key):
            mm = re.search(r'; %s: (\d+)' % re.escape(key), body)
            return int(mm.group(1)) if mm else -1
        ops = Counter(re.findall(r'^\s+((?:global|buffer|scratch|flat)_(?:load|store)_\w+)', body, re.M))
        valu = len(re.findall(r'^\s+v_\w+', body, re.M))
        sp = re.search(r'\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)' % re.escape(name), text)
        out.append((dem, meta('NumVgprs'), meta('NumAgprs'), meta('TotalNumSgprs'),
                    int(sp.group(1)) if sp else -1, meta('ScratchSize'),
                    meta('LDSByteSize'), meta('Occupancy'), valu, dict(ops)))
    for dem, v, ag, sg, sp, scr, lds, occ, valu, ops in out:
        short = re.sub(r'\(.*\)$', '', dem)
        print('%s\n    vgpr %d agpr %d sgpr %d spill %d scratch %d lds %d occupancy %d valu_static %d\n    %s'
              % (short, v, ag, sg, sp, scr, lds, occ, valu,
                 ' '.join('%s:%d' % kv for kv in sorted(ops.items()))))


if __name__ == '__main__':
    main()
