"""Developer aid: GPU occupancy of the training steps in a rocprofv3 kernel trace.  A step = the launches between two
consecutive skg_adamw_kernel launches.  Prints per step: span, busy time (union of kernel intervals over all queues), the
largest idle gaps and what sits on either side.   usage: step_gaps.py <rocprof dir> [first=10] [count=6] [--list]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
count = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0][-40:]
marks = [i for i, r in enumerate(rows) if 'skg_adamw_kernel' in r['Kernel_Name']]
for s in range(first, min(first + count, len(marks) - 1)):
    seg = rows[marks[s] + 1: marks[s + 1] + 1]
    t0 = int(rows[marks[s]]['End_Timestamp'])
    t1 = int(seg[-1]['End_Timestamp'])
    busy, cur_end, gaps = 0, t0, []
    prev = rows[marks[s]]
    for r in seg:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if a > cur_end:
            gaps.append((a - cur_end, name(prev), name(r), (cur_end - t0) / 1e3))
            busy += b - a; cur_end = b
        elif b > cur_end:
            busy += b - cur_end; cur_end = b
        if b >= cur_end:
            prev = r
    queues = sorted({r.get('Queue_Id', '?') for r in seg})
    print("step %d: %d launches, span %.1f us, busy %.1f us (%.0f %%), queues %s" % (s, len(seg), (t1 - t0) / 1e3, busy / 1e3,
                                                                               100.0 * busy / max(t1 - t0, 1), queues))
    for g, a, b, at in sorted(gaps, reverse=True)[:8]:
        print("      gap %7.1f us at %8.1f  after %-40s before %s" % (g / 1e3, at, a, b))
    if '--list' in sys.argv and s == first:
        for r in seg:
            print("   %9.1f %7.1f q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3,
                                            (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', ''), name(r)))
