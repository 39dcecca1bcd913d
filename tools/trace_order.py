"""Ordered kernel list of ONE eager generator step from a rocprofv3 --kernel-trace CSV (which layer runs on which
kernel, and for how long).  usage: python tools/trace_order.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2] if len(sys.argv) > 2 else 'bicubic_fwd'
starts = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f  gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
    prev_end = e
print('step span %.1f us, %d launches' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
