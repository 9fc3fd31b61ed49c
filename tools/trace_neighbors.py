"""What precedes / follows a given kernel in a rocprofv3 --kernel-trace CSV (to identify anonymous runtime kernels such as
__amd_rocclr_copyBuffer by their position in the step).  usage: python tools/trace_neighbors.py <kernel_trace.csv> <substring> [n]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prev = collections.Counter(); nxt = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(rows):
    if pat in r['Kernel_Name']:
        prev[rows[i - 1]['Kernel_Name'][:70] if i else '-'] += 1
        nxt[rows[i + 1]['Kernel_Name'][:70] if i + 1 < len(rows) else '-'] += 1
        sizes[(r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('Stream_Id', r.get('Queue_Id', '?')))] += 1
print("matches:", sum(prev.values()))
print("preceded by:"); [print("  %4d %s" % (c, k)) for k, c in prev.most_common(n)]
print("followed by:"); [print("  %4d %s" % (c, k)) for k, c in nxt.most_common(n)]
print("(grid, block, stream):", sizes.most_common(8))
