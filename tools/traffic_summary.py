"""Developer aid: condenses tools/profile_round.sh output into the two tables kept under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats) and <tag>_hbm_traffic.csv (FETCH_SIZE / WRITE_SIZE passes, per launch).
usage: traffic_summary.py <prof dir> <tag>"""
import collections, csv, glob, os, sys

d, tag = sys.argv[1], sys.argv[2]


def short(name):
    for key, rep in (("void ", ""), ("(skg_gemm_desc)", ""), ("(skg_gemm_group_args)", "")):
        name = name.replace(key, rep)
    return name.split("(")[0] if name.startswith("skg_") and "<" not in name else name


def pmc(counter):
    agg = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (short(r["Kernel_Name"]), r.get("Grid_Size", ""))
            e = agg.setdefault(k, [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return agg


fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
with open(os.path.join(d, tag + "_hbm_traffic.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   --pmc WRITE_SIZE  (separate passes) of: python3 bench.py "
            "--steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-gemm-timer\n")
    f.write("# FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads "
            "(MI355X_MICROARCH.md, HBM): fetch_MB_corrected = 2*raw/1024\n")
    f.write("kernel,grid_size,launches,fetch_KB_raw_avg,fetch_MB_corrected_avg,write_MB_avg\n")
    for k, (s, n) in fetch.items():
        w = write.get(k, [0.0, 1])
        f.write("%s,%s,%d,%.1f,%.2f,%.2f\n" % (k[0], k[1], n, s / n, 2 * s / n / 1024, w[0] / max(w[1], 1) / 1024))
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(d, tag + "_kernel_stats.csv"), "w") as g:
        csv.writer(g).writerows(rows)
    for r in rows[:12]:
        print(",".join(r)[:200])
print(open(os.path.join(d, tag + "_hbm_traffic.csv")).read())
