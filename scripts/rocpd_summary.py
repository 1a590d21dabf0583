#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into a per-kernel table:
calls, total ms, average us, min/max us, share.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def sequence(con, pattern, out):
    """Dispatches in start order with a marker on those matching `pattern` (context: 2 before, 1 after)."""
    rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    hit = [i for i, r in enumerate(rows) if pattern in r[0]]
    keep = sorted({j for i in hit for j in (i - 2, i - 1, i, i + 1) if 0 <= j < len(rows)})
    with open(out, "w") as f:
        for j in keep:
            n, s, e, st = rows[j]
            f.write(f"{j:6d} {'*' if pattern in n else ' '} stream {st} {(e - s) / 1e3:8.2f} us  {n[:100]}\n")


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    if len(sys.argv) > 4 and sys.argv[2] == "--sequence":
        return sequence(con, sys.argv[3], sys.argv[4])
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    text = "\n".join(lines) + f"\n\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
