#!/usr/bin/env python3
"""How busy was the GPU, and how much did the streams overlap?  From a rocprofv3 rocpd SQLite result (--kernel-trace):
over the window of the last `steps` bench steps (delimited by the preprocess kernel) -- the union of kernel intervals
(any kernel running), the sum of kernel durations (= union when nothing overlaps), and the same per stream.
Usage: rocpd_timeline.py results.db [out.md] [marker-kernel-substring]"""
import sqlite3
import sys


def union_length(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    db = sys.argv[1]
    marker = sys.argv[3] if len(sys.argv) > 3 else "preprocess_kernel"
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    marks = [r[1] for r in rows if marker in r[0]]
    if len(marks) < 3:
        raise SystemExit("not enough steps in the trace")
    # steady-state window: from the start of the 3rd-last step to the start of the last one (2 full steps)
    t0, t1 = marks[-3], marks[-1]
    win = [(max(s, t0), min(e, t1), st, n) for n, s, e, st in rows if e > t0 and s < t1]
    wall = t1 - t0
    busy = union_length([(s, e) for s, e, _, _ in win])
    total = sum(e - s for s, e, _, _ in win)
    lines = [f"window: 2 steady-state steps, {wall / 2e6:.3f} ms per step",
             "", "| | ms per step | share of the step |", "|---|---|---|",
             f"| some kernel running (union of intervals) | {busy / 2e6:.3f} | {100 * busy / wall:.2f} % |",
             f"| sum of kernel durations | {total / 2e6:.3f} | {100 * total / wall:.2f} % |",
             f"| time with two or more kernels in flight (sum - union) | {(total - busy) / 2e6:.3f} | {100 * (total - busy) / wall:.2f} % |"]
    streams = sorted({st for _, _, st, _ in win})
    lines += ["", "| stream | kernels per step | busy ms per step |", "|---|---|---|"]
    for st in streams:
        iv = [(s, e) for s, e, x, _ in win if x == st]
        lines.append(f"| {st} | {len(iv) / 2:.0f} | {union_length(iv) / 2e6:.3f} |")
    # the longest stretches with NO kernel running, and what ran before / after them
    merged = []
    for s_, e_, _, n_ in sorted(win):
        if merged and s_ <= merged[-1][1]:
            if e_ > merged[-1][1]:
                merged[-1][1], merged[-1][3] = e_, n_
        else:
            merged.append([s_, e_, n_, n_])
    gaps = [(merged[i + 1][0] - merged[i][1], merged[i][3], merged[i + 1][2]) for i in range(len(merged) - 1)]
    idle = sum(g for g, _, _ in gaps)
    lines += ["", f"idle between kernels: {idle / 2e6:.3f} ms per step in {len(gaps) / 2:.0f} gaps; the longest:", "",
              "| gap us | kernel before | kernel after |", "|---|---|---|"]
    for g, a, b in sorted(gaps, reverse=True)[:16]:
        lines.append(f"| {g / 1e3:.1f} | `{a[:70]}` | `{b[:70]}` |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
