"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (largest first) + busy / wall summary."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))))
# the timed steps: from the 3rd blockmax_i8 launch on (2 warm-ups)
idx = [i for i, r in enumerate(rows) if "blockmax_i8" in r[2]]
lo = idx[int(sys.argv[2]) if len(sys.argv) > 2 else 2]
hi = idx[-1]
seg = rows[lo - 8:hi + 14]
gaps = []
for a, b in zip(seg, seg[1:]):
    gaps.append(((b[0] - a[1]) / 1e6, a[2], b[2]))
print("wall %.1f ms, busy %.1f ms" % ((seg[-1][1] - seg[0][0]) / 1e6, sum(r[1] - r[0] for r in seg) / 1e6))
for g in sorted(gaps, reverse=True)[:25]:
    print("%8.3f ms  after %-60s before %s" % g)
