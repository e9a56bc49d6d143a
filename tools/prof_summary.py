"""Print the per-step kernel summary of a rocprofv3 --kernel-trace --stats CSV directory (tools only; not used by the product)."""
import csv, glob, sys
d, nsteps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / nsteps / 1e6, 3), "kernels/step", sum(int(r["Calls"]) for r in rows) / nsteps)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-100s %6d %8.3f %10.1f" % (r["Name"][:100], int(r["Calls"]), float(r["TotalDurationNs"]) / nsteps / 1e6, float(r["AverageNs"])))
