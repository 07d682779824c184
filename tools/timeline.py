"""Timeline of the J/K calls in a `rocprofv3 --kernel-trace` CSV: per call (delimited by the shell_block_max kernel that opens
every get_jk) the span from the first kernel start to the last kernel end, the summed kernel time, the time during which at
least one kernel runs (union), the number of kernels in flight on average and the idle gaps -- tells a launch-bound call
(union << span) from a kernel-bound one.    usage: python tools/timeline.py <kernel_trace.csv> [first_kernel_substring]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
opener = sys.argv[2] if len(sys.argv) > 2 else "shell_block_max"
starts = [i for i, r in enumerate(rows) if opener in r[2]]
if not starts:
    sys.exit(f"no kernel matching {opener!r}")
starts.append(len(rows))
print(f"{len(starts) - 1} calls; columns: kernels, span ms, sum of kernel ms, union (>= 1 kernel running) ms, mean in flight, "
      "largest gap ms, gap to the next call ms")
for c in range(len(starts) - 1):
    seg = rows[starts[c]:starts[c + 1]]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    tot = sum(r[1] - r[0] for r in seg)
    union, cur_s, cur_e, gap = 0, seg[0][0], seg[0][1], 0
    for s, e, _, _ in seg[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            gap = max(gap, s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    nxt = rows[starts[c + 1]][0] - t1 if starts[c + 1] < len(rows) else 0
    jk = [r for r in seg if "jk_tile" in r[2]]
    first_jk = (jk[0][0] - t0) / 1e6 if jk else 0
    print(f"  call {c}: {len(seg):4d} {(t1 - t0) / 1e6:8.3f} {tot / 1e6:8.3f} {union / 1e6:8.3f} {tot / max(union, 1):6.2f} "
          f"{gap / 1e6:7.3f} {nxt / 1e6:7.3f}   first jk_tile kernel after {first_jk:.3f} ms, "
          f"{len(set(r[3] for r in jk))} queues")
if len(starts) > 2:
    seg = rows[starts[-3]:starts[-2]]
    t0 = seg[0][0]
    print("kernels of the second-to-last call (start ms, duration ms, queue, name):")
    for s, e, n, q in seg:
        print(f"    {(s - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  q{q}  {n[:60]}")
