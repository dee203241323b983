"""Per-launch durations of the tile kernels from a rocprofv3 kernel trace, by position in the substep (phase launches
when the phases are not fused): python scripts/probe_phases.py t_kernel_trace.csv launches_per_substep"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ("k_tile_pos", "k_tile_vel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    d = d[len(d) // 4:]   # skip the first ticks
    for k in range(per):
        v = d[k::per]
        print(f"  {name} launch {k} of {per}: avg {sum(v) / max(len(v), 1):7.2f} us over {len(v)}")
# gaps between consecutive kernels of the stream (launch hand-over)
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
g = [(ts[i + 1][0] - ts[i][1]) / 1e3 for i in range(len(ts) // 2, len(ts) - 1)]
print(f"  mean gap between kernels {sum(g) / len(g):.2f} us, kernels busy {sum((e - s) for s, e, _ in ts[len(ts)//2:]) / 1e3 / (len(ts) - len(ts)//2):.2f} us avg")
