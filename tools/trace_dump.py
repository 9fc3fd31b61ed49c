"""Ordered (kernel, grid, duration us) list of one bench step out of a rocprofv3 --kernel-trace CSV: python tools/trace_dump.py <csv> <marker substring> [nth]
prints everything between the nth and (nth+1)th launch of the marker kernel (default: pack_frag, 3rd occurrence)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2] if len(sys.argv) > 2 else 'pack_frag'; nth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
a, b = idx[nth], idx[nth + 1]
last = None; rep = 0
def flush():
    if last is not None: print(("%4dx " % rep) + last)
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = "%-60s grid %8s" % (r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
    if key == last: rep += 1
    else:
        flush(); last = key; rep = 1
flush()
