"""Markdown table of a rocprofv3 --kernel-trace --stats kernel_stats.csv (top kernels by total time)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches\n")
print("| kernel | calls | avg µs | total ms | % |")
print("|---|---|---|---|---|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "\\|")
    if len(name) > 110:
        name = name[:110] + "..."
    print(f"| `{name}` | {int(r['Calls'])} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.1f} |")
