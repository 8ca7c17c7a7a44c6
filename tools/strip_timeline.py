"""GPU timeline of ONE steady-state frame of the native strip driver (tools/strip_overhead.py under rocprofv3):
every kernel and copy between two consecutive k_raycast starts, with start offset, duration and the gap to the
previous operation's end, plus frame period / union-busy / idle time averaged over the steady frames.

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/strip_overhead.py --only 1920x1080:8:sparse
  python tools/strip_timeline.py DIR
"""
import csv, glob, os, sys

d = sys.argv[1]
ops = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], "q%s" % r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "dma"))
ops.sort()
starts = [i for i, o in enumerate(ops) if "k_raycast" in o[2]]
if len(starts) < 12:
    sys.exit("too few frames in the trace")
steady = starts[len(starts) // 2:]
per, busy = [], []
for a, b in zip(steady[:-1], steady[1:]):
    t0, t1 = ops[a][0], ops[b][0]
    per.append(t1 - t0)
    u, end = 0, t0
    for s, e, *_ in ops[a:b]:
        s = max(s, end)
        if e > s:
            u += e - s; end = e
    busy.append(u)
n = len(per)
print("frames %d: period %.1f us, busy (union) %.1f us, idle %.1f us, operations per frame %.1f" % (
    n, sum(per) / n / 1e3, sum(busy) / n / 1e3, (sum(per) - sum(busy)) / n / 1e3, (steady[-1] - steady[0]) / n))
a, b = steady[len(steady) // 2], steady[len(steady) // 2 + 1]  # a frame from the middle of the steady part: the last ones belong to the draining pipeline
t0, end = ops[a][0], ops[a][0]
print("%9s %8s %8s  %-5s %s" % ("start us", "dur us", "gap us", "queue", "operation"))
for s, e, name, q in ops[a:b]:
    print("%9.1f %8.1f %8.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, q, name))
    end = max(end, e)
