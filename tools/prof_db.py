#!/usr/bin/env python3
"""Summaries of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...`).

  python tools/prof_db.py stats  DB [--md OUT.md] [--csv OUT.csv]   per-kernel calls / total / average / share
  python tools/prof_db.py seq    DB [--from I] [--count N]           the dispatches of that step in order: start offset, duration,
                                                                     idle gap in front, workgroups, name
  python tools/prof_db.py step   DB                                  timeline of ONE steady-state (hipGraph-replayed) step:
                                                                     wall, union-busy, concurrency histogram, and for every
                                                                     kernel family the time it runs ALONE on the chip

A "step" is delimited by consecutive `grad_sqnorm_kernel` dispatches (one per training step); the last complete
interval is a graph replay in bench.py.
"""
import argparse
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    name = m.group(1) if m else name
    name = name.replace("at::native::", "aten::")
    return name[:110]


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, queue_id, stream_id, grid_x*grid_y*grid_z/ (workgroup_x*workgroup_y*workgroup_z), vgpr_count, lds_size from kernels order by start").fetchall()
    return [(short(n), s, e, q, st, wg, v, l) for n, s, e, q, st, wg, v, l in rows]


def cmd_stats(args):
    rows = load(args.db)
    agg = defaultdict(lambda: [0, 0])
    for n, s, e, *_ in rows:
        agg[n][0] += 1
        agg[n][1] += e - s
    tot = sum(v[1] for v in agg.values())
    lines = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("name,calls,total_us,avg_us,percent\n")
            for n, (k, t) in lines:
                f.write(f"\"{n}\",{k},{t/1e3:.3f},{t/1e3/k:.3f},{100*t/tot:.3f}\n")
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for n, (k, t) in lines[: args.top]:
        out.append(f"| `{n}` | {k} | {t/1e6:.3f} | {t/1e3/k:.2f} | {100*t/tot:.2f} |")
    text = "\n".join(out)
    if args.md:
        with open(args.md, "w") as f:
            f.write(f"Per-kernel statistics from `{args.db}` (all dispatches of the profiled command; total kernel time {tot/1e6:.2f} ms)\n\n" + text + "\n")
    print(text)


def cmd_seq(args):
    rows = load(args.db)
    marks = [i for i, r in enumerate(rows) if r[0].startswith("grad_sqnorm_kernel")]
    step = rows[marks[-3] + 1: marks[-2] + 1]
    t0, prev_end, gaps, busy = step[0][1], step[0][1], 0, 0
    for i, (n, s0, e0, q, st, wg, v, l) in enumerate(step):
        gap = s0 - prev_end
        gaps += max(gap, 0)
        busy += e0 - s0
        if args.first <= i < args.first + args.count:
            print(f"{i:5d} +{(s0 - t0) / 1e3:9.1f} us  {(e0 - s0) / 1e3:7.1f} us  gap {gap / 1e3:5.1f}  wg {wg:6d}  {n}")
        prev_end = max(prev_end, e0)
    print(f"step: {len(step)} dispatches, busy {busy / 1e6:.3f} ms, idle gaps {gaps / 1e6:.3f} ms")


def cmd_step(args):
    rows = load(args.db)
    marks = [i for i, r in enumerate(rows) if r[0].startswith("grad_sqnorm_kernel")]
    if len(marks) < 3:
        sys.exit("need >= 3 steps in the trace")
    a, b = marks[-3], marks[-2]          # a complete interval well inside the timed (graph-replayed) region
    step = rows[a:b]
    t0, t1 = step[0][1], max(r[2] for r in step)
    t1 = rows[b][1]
    ev = []
    for i, (n, s, e, *_r) in enumerate(step):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    live = set()
    last = t0
    conc = defaultdict(int)
    alone = defaultdict(int)
    shared = defaultdict(float)
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            conc[len(live)] += dt
            if len(live) == 1:
                alone[step[next(iter(live))][0]] += dt
            for j in live:
                shared[step[j][0]] += dt / len(live)
        last = t
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
    conc[0] += max(0, t1 - last)
    wall = t1 - t0
    print(f"step wall {wall/1e6:.3f} ms, {len(step)} dispatches, queues {sorted(set(r[3] for r in step))}")
    print("concurrency (kernels in flight) -> ms:", {k: round(v / 1e6, 3) for k, v in sorted(conc.items())})
    print(f"\ntime share attributed per kernel (dt / kernels in flight), top {args.top}:")
    for n, v in sorted(shared.items(), key=lambda kv: -kv[1])[:args.top]:
        cnt = sum(1 for r in step if r[0] == n)
        tot = sum(r[2] - r[1] for r in step if r[0] == n)
        print(f"  {v/1e6:7.3f} ms share | alone {alone.get(n,0)/1e6:7.3f} | sum {tot/1e6:7.3f} | n={cnt:4d} | {n}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("stats"); s.add_argument("db"); s.add_argument("--md"); s.add_argument("--csv"); s.add_argument("--top", type=int, default=60)
    s.set_defaults(fn=cmd_stats)
    t = sub.add_parser("step"); t.add_argument("db"); t.add_argument("--top", type=int, default=40); t.set_defaults(fn=cmd_step)
    q = sub.add_parser("seq"); q.add_argument("db"); q.add_argument("--from", dest="first", type=int, default=0)
    q.add_argument("--count", type=int, default=100000); q.set_defaults(fn=cmd_seq)
    a = ap.parse_args()
    a.fn(a)
