#!/usr/bin/env python3
"""Print the per-kernel timeline of one step from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('traceKernel<false', 'shadeKernel', 'generateKernel', 'accumulateKernel'))]
gi = [i for i, r in enumerate(sel) if 'generate' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(gi) // 2
i0, i1 = gi[k], gi[k + 1]
t0 = int(sel[i0]['Start_Timestamp']); prev = None
for r in sel[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-28s start %8.1f dur %7.1f us gap %5.1f' % (r['Kernel_Name'][:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
print('step total %.1f us' % ((prev - t0) / 1e3))
