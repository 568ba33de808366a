#!/usr/bin/env python3
"""Kernels of one wavefront pass from a rocprofv3 kernel trace CSV, all queues, in start order, with the queue each ran on:
shows whether the lanes of a pass (device_api.hip renderPass) overlap. usage: tools/lane_timeline.py <kernel_trace.csv> [pass index]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r['Kernel_Name'] for k in ('traceKernel<', 'shadeKernel', 'generateKernel', 'accumulateKernel'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = [i for i, r in enumerate(rows) if 'accumulate' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(acc) // 2
sel = rows[acc[k - 1] + 1: acc[k] + 1]
t0 = int(sel[0]['Start_Timestamp'])
queues = sorted(set(r['Queue_Id'] for r in sel))
busy = 0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('q%d %-24s start %8.1f end %8.1f dur %7.1f us' % (queues.index(r['Queue_Id']), r['Kernel_Name'].replace('void ', '').replace('twk::', '')[:24], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print('pass total %.1f us, sum of kernel durations %.1f us' % ((max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3))
