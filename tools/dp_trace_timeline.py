#!/usr/bin/env python3
"""Timeline of the last few training steps out of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory:
every dispatch / copy with its start offset inside the step, duration, stream (queue) and the gap to the previous
dispatch on the same queue.  A step starts at a k_fwd<0,..> launch that follows a k_dwp / k_apply_update / k_bias_apply."""
import csv, glob, re, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), "K"))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", ""), "copy", "C"))
rows.sort()
def short(n):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:34]
# step boundaries: first k_fwd<0 after an update-ish kernel
starts = []
prev_upd = True
for i, (s, e, n, q, k) in enumerate(rows):
    if n.startswith("void k_fwd<0") and prev_upd:
        starts.append(i)
        prev_upd = False
    if "k_dwp" in n or "k_apply_update" in n or "k_bias_apply" in n:
        prev_upd = True
if len(starts) < 8:
    print("too few steps found:", len(starts)); sys.exit(0)
a, b = starts[-6], starts[-3]   # three steady-state steps near the end (the very last ones may be post-passes)
t0 = rows[a][0]
last_end = {}
step_i = 0
for i in range(a, b):
    s, e, n, q, k = rows[i]
    if i in starts:
        print("---- step (previous step took %.1f us)" % ((s - t0) / 1e3) if i != a else "---- step")
        t0 = s
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("  +%7.1f us  %6.1f us  q=%-6s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, short(n)))
    last_end[q] = e
