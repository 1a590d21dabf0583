#!/usr/bin/env python3
"""Per-kernel totals of rocprofv3 --pmc counters from a rocpd SQLite result.
Usage: rocpd_pmc_summary.py results.db [out.md]
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3.  gfx950 correction (MI355X_MICROARCH.md,
HBM section): FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads -> the
'fetch x2' column doubles it; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute(
        "select kernel_name, counter_name, count(distinct dispatch_id), sum(value), "
        "sum(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    # duration is per (dispatch,counter-instance) row: recompute per dispatch
    dur = dict(con.execute("select kernel_name, sum(d) from (select kernel_name, dispatch_id, max(duration) d "
                           "from counters_collection group by kernel_name, dispatch_id) group by kernel_name").fetchall())
    by_k = {}
    for k, c, n, v, _ in rows:
        by_k.setdefault(k, {"n": n})[c] = v
    names = sorted({c for _, c, _, _, _ in rows})
    lines = ["| kernel | dispatches | total ms | " + " | ".join(names) + " |", "|---|---|---|" + "---|" * len(names)]
    for k, d in sorted(by_k.items(), key=lambda kv: -dur.get(kv[0], 0)):
        short = k if len(k) < 90 else k[:87] + "..."
        lines.append(f"| `{short}` | {d['n']} | {dur.get(k, 0) / 1e6:.3f} | " +
                     " | ".join(f"{d.get(c, 0):.4g}" for c in names) + " |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
