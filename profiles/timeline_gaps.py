#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: per-kernel mean duration and the idle gaps between
consecutive kernels in the timed region (last 60% of our kernels)."""
import glob, os, sys
import pandas as pd
src = sys.argv[1]
f = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)[0]
df = pd.read_csv(f, usecols=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp']).sort_values('Start_Timestamp')
ours = df[df.Kernel_Name.str.contains('ld_tile|ld_sym|ld_eig|ld_colsum|ld_rowsum|snp_pass|snp_given|init_state|delta_kernel|reduce_cols|finalize|mean_diff|mstep|decide|nccl|rccl|AllReduce')]
ours = ours.iloc[int(len(ours) * 0.4):].copy()
ours['short'] = ours.Kernel_Name.str.extract(r'(ld_tile_combine|ld_tile|ld_sym_combine|ld_sym|ld_eig_fused|ld_rowsum_combine|ld_rowsum|ld_colsum|snp_pass_kernel<\d, \w+|delta_kernel|reduce_cols|finalize|mean_diff_final|mean_diff|mstep|decide|init_state|nccl\w*|rccl\w*|\w*AllReduce\w*)')[0]
ours['dur'] = (ours.End_Timestamp - ours.Start_Timestamp) / 1e3
ours['gap_before'] = (ours.Start_Timestamp - ours.End_Timestamp.shift(1)) / 1e3
print(ours.groupby('short').agg(calls=('dur', 'size'), dur_us=('dur', 'mean'), gap_before_us=('gap_before', 'mean')).round(2))
span = (ours.End_Timestamp.iloc[-1] - ours.Start_Timestamp.iloc[0]) / 1e3
print('span %.1f us, busy %.1f us (%.1f%%), n_finalize %d -> %.1f us per evaluation'
      % (span, ours.dur.sum(), 100 * ours.dur.sum() / span, (ours.short == 'finalize').sum(),
         span / max(1, (ours.short == 'finalize').sum())))
