"""Summarise rocprofv3 CSV output (kernel stats + PMC counters per kernel) into one text file."""
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    lines.append(f"# {os.path.relpath(f, out)}")
    lines += [l.rstrip() for l in open(f)][:20]
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in sorted(glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[k] += 1
        lines.append(f"# {d}: per-kernel counter sums / dispatches")
        for k in acc:
            lines.append(f"{k}  dispatches={cnt[k]}  " + "  ".join(f"{c}={v/cnt[k]:.4g}" for c, v in sorted(acc[k].items())))
txt = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
print(txt)
