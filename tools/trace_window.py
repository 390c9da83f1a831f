#!/usr/bin/env python
"""Print a window of a rocprofv3 --kernel-trace CSV around the kernels of the model's own stream (rd_set_refine_async): start / end
in microseconds, duration, queue, name.   python tools/trace_window.py <rocprof output dir> [kernel-substring]"""
import csv
import glob
import sys

rows = []
for p in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:48], r.get('Queue_Id', '?')))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else 'rd_refine'
lstm_q = [r[3] for r in rows if 'rd_lstm' in r[2]]
main_q = max(set(lstm_q), key=lstm_q.count)
side = [i for i, r in enumerate(rows) if key in r[2] and r[3] != main_q]
print("main queue", main_q, "; %d '%s' kernels on other queues" % (len(side), key))
if side:
    mid = side[len(side) // 2]
    t0 = rows[max(0, mid - 14)][0]
    for s, e, k, q in rows[max(0, mid - 14):mid + 22]:
        print("%9.1f %9.1f  %7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
