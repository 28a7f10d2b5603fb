#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (run on the GPU box) to the small summaries kept in profiles/.

    python profiles/summarize_rocprof.py <rocprof_output_dir> <out_prefix>

Writes <out_prefix>_kernel_stats.csv (rocprofv3 --stats table, our kernels + top others),
<out_prefix>_vilma_kernels.json (per-kernel count / avg / min / max duration from the kernel
trace) and, when a counter collection is present, <out_prefix>_pmc.json (per-kernel mean counter
value per dispatch)."""
import glob
import json
import os
import sys

import pandas as pd

OURS = ("ld_tile_combine_kernel", "ld_tile_kernel", "ld_sym_combine_kernel", "ld_sym_kernel", "ld_eig_fused_kernel", "ld_eig_wave_kernel",
        "ld_rowsum_combine_kernel", "sweep_decide_kernel", "tile_sums_kernel",
        "ld_rowsum_kernel", "ld_colsum_kernel", 'snp_pass_kernel', 'delta_kernel', 'reduce_cols_kernel',
        'finalize_kernel', 'mean_diff', 'gather_x_kernel', 'scatter_y_kernel', 'mstep_kernel',
        'init_state_kernel', 'snp_given_delta_kernel')


def short(name):
    for k in OURS:
        if k in name:
            return k
    return None


def per_class(grp, col):
    """Launches of ld_eig_fused_kernel<R, NR> by class R: mean of `col` over each class's full
    launches (predicated-off launches last microseconds), and the sum over classes = one product."""
    if not len(grp):
        return None
    grp = grp.copy()
    grp['R'] = grp['Kernel_Name'].str.extract(r'ld_eig_fused_kernel<(\d+),')[0]
    out = {}
    for r, g in grp.groupby('R'):
        big = g[g.us > 0.5 * g.us.max()]
        out[str(r)] = {'launches': int(len(big)), 'mean': float(big[col].mean())}
    return {'per_class': out, 'sum_over_classes': float(sum(v['mean'] for v in out.values()))}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    for f in glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True):
        df = pd.read_csv(f)
        name_col = [c for c in df.columns if 'Name' in c][0]
        keep = df[df[name_col].map(lambda n: short(str(n)) is not None)]
        rest = df[~df.index.isin(keep.index)].head(8)
        pd.concat([keep, rest]).to_csv(prefix + '_kernel_stats.csv', index=False)
    for f in glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True):
        df = pd.read_csv(f, usecols=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        df['k'] = df['Kernel_Name'].map(lambda n: short(str(n)))
        df = df[df['k'].notna()]
        df['us'] = (df['End_Timestamp'] - df['Start_Timestamp']) / 1e3
        out = {}
        for k, grp in df.groupby('k'):
            out[k] = {'calls': int(len(grp)), 'avg_us': float(grp.us.mean()),
                      'min_us': float(grp.us.min()), 'max_us': float(grp.us.max()),
                      'total_ms': float(grp.us.sum() / 1e3)}
        # the dominant kernel's launches split by size: the big launches are the full LD product
        # (launches of a stage queued ahead of a line-search decision that then went the other way
        # exit at once: a few microseconds each; they are not products)
        for kname in ('ld_tile_kernel', 'ld_sym_kernel'):
            ld = df[df.k == kname]
            if len(ld):
                big = ld[ld.us > 0.5 * ld.us.max()]
                out[kname + '_full_product'] = {'calls': int(len(big)), 'avg_us': float(big.us.mean()),
                                                'skipped_launches': int(len(ld) - len(big))}
                for nr in (1, 2):       # by right-hand sides per pass
                    sub = ld[ld.Kernel_Name.str.contains('%s<%d>' % (kname, nr), regex=False)]
                    sub = sub[sub.us > 0.5 * ld.us.max()]
                    if len(sub):
                        out['%s<%d>_full_product' % (kname, nr)] = {'calls': int(len(sub)),
                                                                    'avg_us': float(sub.us.mean())}
        # an eigen-form product = one launch per block-height class (template argument R): the
        # product's kernel time is the sum over the classes of the mean full launch
        eig = per_class(df[df.k == 'ld_eig_fused_kernel'], 'us')
        if eig:
            out['ld_eig_fused_kernel_full_product'] = eig
        json.dump(out, open(prefix + '_vilma_kernels.json', 'w'), indent=1)
    for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        df = pd.read_csv(f, usecols=['Kernel_Name', 'Counter_Name', 'Counter_Value',
                                     'Start_Timestamp', 'End_Timestamp'])
        df['k'] = df['Kernel_Name'].map(lambda n: short(str(n)))
        df = df[df['k'].notna()]
        df['us'] = (df['End_Timestamp'] - df['Start_Timestamp']) / 1e3
        out = {}
        for (k, c), grp in df.groupby(['k', 'Counter_Name']):
            big = grp[grp.us > 0.5 * grp.us.max()]
            out.setdefault(k, {})[c] = {'dispatches': int(len(grp)),
                                        'mean_per_dispatch': float(grp.Counter_Value.mean()),
                                        'mean_over_large_dispatches': float(big.Counter_Value.mean()),
                                        'large_dispatches': int(len(big))}
            if k == 'ld_eig_fused_kernel':
                # per PRODUCT: the sum over the block-height classes of the mean full launch
                pc = per_class(grp, 'Counter_Value')
                out[k][c]['per_class'] = pc['per_class']
                out[k][c]['mean_over_large_dispatches'] = pc['sum_over_classes']
        json.dump(out, open(prefix + '_pmc.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
