"""Timeline analysis of a rocprofv3 --kernel-trace CSV: for the last full training step in the trace (delimited by the
fused AdamW launches) report wall time, the union of busy time, the idle time, the average number of kernels in
flight and the per-queue busy time.  Usage: python tools/trace_timeline.py <kernel_trace.csv> [steps_back]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"],
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    # steps end with the last adamw launch of a step; consecutive adamw launches closer than 1 ms belong to one step
    ends = []
    for i in adam:
        if ends and rows[i][0] - rows[ends[-1]][1] < 1_000_000:
            ends[-1] = i
        else:
            ends.append(i)
    if len(ends) < back + 1:
        print("not enough steps in trace", len(ends)); return
    lo_i, hi_i = ends[-back - 1] + 1, ends[-back]
    step = rows[lo_i:hi_i + 1]
    t0, t1 = step[0][0], max(r[1] for r in step)
    wall = t1 - t0
    ev = []
    for s, e, *_ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0; depth = 0; last = t0; weighted = 0; hist = defaultdict(int)
    for t, d in ev:
        if depth > 0:
            busy += t - last
        hist[min(depth, 6)] += t - last
        weighted += depth * (t - last)
        depth += d; last = t
    print(f"launches in step: {len(step)}   wall {wall/1e6:.2f} ms   busy(union) {busy/1e6:.2f} ms   idle {100*(wall-busy)/wall:.1f}%   "
          f"sum of kernel time {weighted/1e6:.2f} ms   avg in flight {weighted/max(busy,1):.2f}")
    print("time with k kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
    perq = defaultdict(lambda: [0, 0])
    for s, e, n, q, st, wg in step:
        perq[q][0] += e - s; perq[q][1] += 1
    for q, (t, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
        print(f"  queue {q}: {c} launches, {t/1e6:.2f} ms")
    # small-grid share: kernels that cannot fill 256 CUs
    small = sum(e - s for s, e, n, q, st, wg in step if wg < 256)
    print(f"kernel time in launches with < 256 workgroups: {small/1e6:.2f} ms;  < 10 us launches: "
          f"{sum(1 for r in step if r[1]-r[0] < 10000)} ({sum(r[1]-r[0] for r in step if r[1]-r[0] < 10000)/1e6:.2f} ms)")
    # gaps: idle intervals by size
    gaps = []
    depth = 0; last = t0
    for t, d in ev:
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d; last = t
    gaps.sort()
    if gaps:
        print(f"idle gaps: {len(gaps)}  median {gaps[len(gaps)//2]/1e3:.1f} us  p90 {gaps[int(len(gaps)*0.9)]/1e3:.1f} us  max {gaps[-1]/1e3:.1f} us  total {sum(gaps)/1e6:.2f} ms")


if __name__ == "__main__":
    main()
