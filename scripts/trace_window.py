"""Per-kernel mean duration over the LAST ticks of a rocprofv3 --kernel-trace run of bench.py (the settled pile the
timed region runs on; the --stats summary of the same run averages over the 600 settling ticks as well):
   python scripts/trace_window.py <dir>/bench_kernel_trace.csv [ticks=200] [skip_last=70]
A tick starts at a k_aabb_count dispatch (its first kernel; traces from before the double zero block: k_zero).  Durations are End - Start of each dispatch; gaps are Start(next) - End."""
import collections
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 70     # the second (all-events) leg and the stats calls at the end
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfx::", "") for r in rows]
first = "k_zero" if "k_zero" in names else "k_aabb_count"
zeros = [i for i, n in enumerate(names) if n == first]
sel = zeros[-(ticks + skip) - 1:-skip]
durs, tick_ns, busy, gap = collections.defaultdict(list), [], [], []
for a, b in zip(sel[:-1], sel[1:]):
    tick_ns.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
    tot = 0
    for i in range(a, b):
        d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        durs[names[i]].append(d)
        tot += d
        gap.append(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]))
    busy.append(tot)
n = len(tick_ns)
print(f"{n} ticks: {st.mean(tick_ns) / 1e3:.1f} us per tick under the profiler, kernels busy {st.mean(busy) / 1e3:.1f} us, "
      f"mean gap between dispatches {st.mean(gap) / 1e3:.2f} us")
print(f"{'kernel':32s} {'launches/tick':>13s} {'mean us':>9s} {'us/tick':>9s}")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:32s} {len(v) / n:13.2f} {st.mean(v) / 1e3:9.2f} {sum(v) / n / 1e3:9.1f}")
