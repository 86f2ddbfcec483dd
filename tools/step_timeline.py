"""Developer aid: per-launch timeline of the last training step in a rocprofv3 kernel trace.
usage: step_timeline.py <dir with *_kernel_trace.csv> [optimizer launches per step = 1 (skg_adamw; torch's fused AdamW: 12)]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name'] or 'skg_adamw' in r['Kernel_Name']]
seg = rows[idx[-per - 1] + 1: idx[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
    n = r['Kernel_Name'].split('(')[0][-44:]
    print("%8.1f %7.1f  q%-2s s%-2s grid %-8s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d / 1e3, r.get('Queue_Id', '?'), r.get('Stream_Id', '?'),
                                                 r.get('Grid_Size_X', r.get('Grid_Size', '')), n))
print(len(seg), 'launches, busy %.1f us, span %.1f us' % (tot / 1e3, (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
