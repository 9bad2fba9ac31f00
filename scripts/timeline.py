#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 kernel trace of bench.py: for the last timed step, how much wall-clock has
0 / 1 / 2 / 3 streams busy, and which kernels run while fewer than all three streams are busy.
usage: timeline.py <run_kernel_trace.csv> [step index, default -3: bench.py appends a serial step and the native chain after the timed steps]"""
import collections, csv, re, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.split(r"[<(]", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), name))
rows.sort()
# a step starts with the three IO uploads; the Fq12 chain kernel runs exactly once per step: cut half-way between the
# last launch before it that follows a > 100 us idle period on all queues
marks = [i for i, r in enumerate(rows) if r[3] == "fq12_chain_kernel"]
cuts = []
for m in marks:
    i = m
    while i > 0:
        prev_end = max(e for _, e, _, _ in rows[max(0, i - 40):i])
        if rows[i][0] - prev_end > 100_000:
            break
        i -= 1
    cuts.append(i)
steps = [rows[a:b] for a, b in zip(cuts, cuts[1:] + [len(rows)])]
print("steps found:", len(steps), [len(s) for s in steps])
# bench.py appends a serial step and the native chain after the timed steps: analyse the last step that looks like a timed
# one (the most common launch count), or the index given on the command line
if len(sys.argv) > 2:
    st = steps[int(sys.argv[2])]
else:
    cnt = collections.Counter(len(x) for x in steps[1:]).most_common(1)[0][0]
    st = [x for x in steps if abs(len(x) - cnt) <= 2][-1]
t0, t1 = min(s for s, _, _, _ in st), max(e for _, e, _, _ in st)
print("analysed step: %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(st)))
ev = []
for s, e, q, n in st:
    ev.append((s, 1, q, n))
    ev.append((e, -1, q, n))
ev.sort()
active = collections.Counter()
busy_q = collections.Counter()
hist = collections.Counter()
alone = collections.Counter()
prev = t0
for t, d, q, n in ev:
    dt = t - prev
    if dt > 0:
        nq = sum(1 for v in busy_q.values() if v > 0)
        hist[nq] += dt
        if nq < 3:
            for k, v in active.items():
                if v > 0:
                    alone[k] += dt
    prev = t
    active[n] += d
    busy_q[q] += d
for k in sorted(hist):
    print("  %d queue(s) busy: %7.2f ms" % (k, hist[k] / 1e6))
print("kernels active while < 3 queues busy (ms):")
for k, v in alone.most_common(15):
    print("  %-28s %7.2f" % (k, v / 1e6))
perq = collections.defaultdict(list)
for s, e, q, n in st:
    perq[q].append((s, e, n))
for q, lst in perq.items():
    busy = sum(e - s for s, e, _ in lst)
    print("queue %s: first %.2f last %.2f ms, busy %.2f ms, %d launches" % (q, (lst[0][0] - t0) / 1e6, (max(e for _, e, _ in lst) - t0) / 1e6, busy / 1e6, len(lst)))
    # biggest idle gaps on this queue
    gaps = sorted(((lst[i + 1][0] - lst[i][1], lst[i][2], lst[i + 1][2], (lst[i][1] - t0) / 1e6) for i in range(len(lst) - 1)), reverse=True)[:6]
    for g, a, b, at in gaps:
        print("    gap %.3f ms at %.2f ms between %s -> %s" % (g / 1e6, at, a, b))
print("coarse timeline (2 ms buckets): dominant kernel per queue")
B = 2_000_000
nb = int((t1 - t0) // B) + 1
for q, lst in sorted(perq.items()):
    line = []
    for b in range(nb):
        lo, hi = t0 + b * B, t0 + (b + 1) * B
        acc = collections.Counter()
        for s, e, n in lst:
            ov = min(e, hi) - max(s, lo)
            if ov > 0:
                acc[n] += ov
        line.append("%-10s" % (acc.most_common(1)[0][0].replace("_kernel", "")[:10] if acc else "."))
    print("q%s: %s" % (q, " ".join(line)))
