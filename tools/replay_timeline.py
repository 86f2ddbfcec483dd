"""Developer aid: per-launch timeline of the LAST hipGraph replay (or forward) in a rocprofv3 kernel trace: launches
between the last two skg_pack_detections kernels.   usage: replay_timeline.py <dir> [marker=skg_pack_detections]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "skg_pack_detections"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
seg = rows[idx[-2]: idx[-1]]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
    n = r['Kernel_Name'].split('(')[0][-44:]
    print("%8.1f %7.1f  wg %-7s grid %-8s q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d / 1e3, r.get('Workgroup_Size_X', ''),
                                                     r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Queue_Id', ''), n))
print(len(seg), 'launches, busy %.1f us, span %.1f us' % (tot / 1e3, (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
