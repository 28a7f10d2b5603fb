#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: the raw dispatch sequence (every kernel, ours or not) of a
window in the timed region -- name, queue, start offset, duration, gap to the previous end on any
queue -- to see what sits between two of our kernels (RCCL kernels, copies' helper kernels...).
    sweep_sequence.py <trace dir> [n_rows=90] [skip_fraction=0.7]"""
import glob, os, re, sys
import pandas as pd
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
f = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
qcol = 'Queue_Id' if 'Queue_Id' in df.columns else None
# a decision = the device-resident sweep's decision kernel (r03) or the M-step kernel behind a
# host decision (earlier rounds)
mark = 'sweep_decide' if df.Kernel_Name.str.contains('sweep_decide').any() else 'mstep'
first = df.index[df.Kernel_Name.str.contains(mark)][0]
last = df.index[df.Kernel_Name.str.contains(mark)][-1]
start = int(first + skip * (last - first))
# align on a decision so the window starts at a sweep boundary
while start < len(df) and mark not in df.Kernel_Name[start]:
    start += 1
w = df.iloc[start:start + n].copy()
t0 = w.Start_Timestamp.iloc[0]
prev_end = None
for _, r in w.iterrows():
    name = re.sub(r'\(.*', '', r.Kernel_Name)
    name = re.sub(r'^void ', '', name)[:60]
    gap = (r.Start_Timestamp - prev_end) / 1e3 if prev_end is not None else 0.0
    print('%9.1f us  +%7.1f  dur %8.1f  q%-3s %s' % ((r.Start_Timestamp - t0) / 1e3, gap,
          (r.End_Timestamp - r.Start_Timestamp) / 1e3, r[qcol] if qcol else '?', name))
    prev_end = max(prev_end or 0, r.End_Timestamp)
