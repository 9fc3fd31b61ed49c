"""Timeline summary of one bench step from a rocprofv3 --kernel-trace CSV: wall, per-queue busy, gaps on the main queue."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [i for i, r in enumerate(rows) if 'pack_weights' in r['Kernel_Name']]
a, b = marks[-3], marks[-2]
step = rows[a:b]
t0 = step[0]['s']; t1 = max(r['e'] for r in step)
print('step wall ms %.2f kernels %d' % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    print(' queue', q, 'kernels', len(rs), 'busy ms %.2f' % (sum(r['e'] - r['s'] for r in rs) / 1e6), 'span ms %.2f' % ((max(r['e'] for r in rs) - min(r['s'] for r in rs)) / 1e6))
ev = sorted([(r['s'], 1) for r in step] + [(r['e'], -1) for r in step])
cur = 0; last = None; union = 0
for t, d in ev:
    if cur > 0: union += t - last
    cur += d; last = t
print(' union busy ms %.2f' % (union / 1e6))
main = max(byq.items(), key=lambda kv: len(kv[1]))[1]
main.sort(key=lambda r: r['s'])
gaps = [(main[i + 1]['s'] - main[i]['e'], main[i]['Kernel_Name'][:48], main[i + 1]['Kernel_Name'][:48]) for i in range(len(main) - 1)]
print(' main queue: sum gaps ms %.2f, gaps > 5us: %d' % (sum(g for g, _, _ in gaps if g > 0) / 1e6, sum(1 for g, _, _ in gaps if g > 5000)))
for g, x, y in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    print('   %8.1f us  after %s  before %s' % (g / 1e3, x, y))
