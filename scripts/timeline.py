"""Idle-gap analysis of a rocprofv3 kernel trace (rocpd database): where is the device NOT running a kernel of the main
chain, and between which kernels?

    python scripts/timeline.py <results.db> [--skip-ms T0] [--top N]

Kernels are grouped per stream/queue; the queue with the largest kernel count is taken as the solver's main stream.  Prints its
busy time, the idle time between consecutive kernels split by (previous kernel -> next kernel) pair, a histogram of gap lengths,
and how much of the main stream's idle time is covered by kernels of the other queues (side-stream block builds).
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(?:ahip::)?([A-Za-z_0-9:]+)", name)
    return (m.group(1) if m else name)[:40]


def main():
    db = sys.argv[1]
    top = 25
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("# kernels columns:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    endcol = "end" if "end" in cols else "stop"
    sel = f"select name, start, {endcol}" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
    rows = [(short(n), s, e, q) for n, s, e, q in c.execute(sel)]
    if not rows:
        print("no kernels")
        return
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    print(f"# queues: " + ", ".join(f"{q}: {len(v)} kernels" for q, v in byq.items()) + f"; main = {main_q}")
    ks = byq[main_q]
    # steady part: from the first sweep_kernel after the design upload to the end
    t_first = ks[0][1]
    span = (ks[-1][2] - t_first) / 1e6
    busy = sum(e - s for _, s, e, _ in ks) / 1e6
    print(f"main stream: span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms, {len(ks)} kernels")
    # split the trace into paths: a path starts where a gap > 5 ms precedes (python between steps)
    pair_gap = defaultdict(float)
    pair_cnt = defaultdict(int)
    hist = defaultdict(lambda: [0, 0.0])
    big = []
    others = sorted((s, e) for q, v in byq.items() if q != main_q for _, s, e, _ in v)
    covered = 0.0
    oi = 0
    for (n0, s0, e0, _), (n1, s1, e1, _) in zip(ks[:-1], ks[1:]):
        g = (s1 - e0) / 1e3  # us
        if g <= 0:
            continue
        if g > 5000:
            big.append((g / 1e3, n0, n1))
            continue
        pair_gap[(n0, n1)] += g
        pair_cnt[(n0, n1)] += 1
        b = 1 if g < 2 else 2 if g < 5 else 5 if g < 10 else 10 if g < 20 else 20 if g < 50 else 50 if g < 100 else 100 if g < 300 else 300
        hist[b][0] += 1
        hist[b][1] += g
        # overlap of this gap with other-queue kernels
        while oi < len(others) and others[oi][1] < e0:
            oi += 1
        k = oi
        while k < len(others) and others[k][0] < s1:
            covered += max(0, min(others[k][1], s1) - max(others[k][0], e0)) / 1e3
            k += 1
    tot_gap = sum(pair_gap.values())
    print(f"idle between kernels (gaps <= 5 ms): {tot_gap / 1e3:.2f} ms; of which other queues were busy {covered / 1e3:.2f} ms")
    print(f"gaps > 5 ms (between steps / python): {[f'{g:.1f} ms {a}->{b}' for g, a, b in big][:12]}")
    print("\n# gap histogram (upper edge us: count, total ms)")
    for b in sorted(hist):
        print(f"  <{b if b != 300 else '...':>4} us: {hist[b][0]:6d}  {hist[b][1] / 1e3:8.2f} ms")
    print(f"\n# idle by (previous -> next) kernel, top {top}")
    for (a, b), g in sorted(pair_gap.items(), key=lambda kv: -kv[1])[:top]:
        print(f"  {a:>40s} -> {b:<40s} {pair_cnt[(a, b)]:6d} gaps  {g / 1e3:8.2f} ms  avg {g / pair_cnt[(a, b)]:7.1f} us")
    print("\n# busy time by kernel on the main stream")
    bk = defaultdict(lambda: [0, 0.0])
    for n, s, e, _ in ks:
        bk[n][0] += 1
        bk[n][1] += (e - s) / 1e3
    for n, (cn, t) in sorted(bk.items(), key=lambda kv: -kv[1][1])[:20]:
        print(f"  {n:>40s} {cn:7d}  {t / 1e3:9.2f} ms  avg {t / cn:8.2f} us")
    for q, v in byq.items():
        if q == main_q:
            continue
        bk = defaultdict(lambda: [0, 0.0])
        for n, s, e, _ in v:
            bk[n][0] += 1
            bk[n][1] += (e - s) / 1e3
        print(f"\n# queue {q}")
        for n, (cn, t) in sorted(bk.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"  {n:>40s} {cn:7d}  {t / 1e3:9.2f} ms  avg {t / cn:8.2f} us")


if __name__ == "__main__":
    main()
