#!/usr/bin/env python3
"""Per-DISPATCH counter values (launch order) of the kernels whose name contains a pattern, from a rocprofv3 --pmc rocpd
result.  FETCH_SIZE / WRITE_SIZE are in KB; the gfx950 x2 FETCH correction (scripts/pmc_traffic.py) is applied in the
'fetch MB (x2)' column.  Usage: rocpd_pmc_dispatches.py results.db pattern [out.md]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    rows = con.execute("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection "
                       "group by dispatch_id, kernel_name, counter_name order by dispatch_id").fetchall()
    by_d, names = {}, []
    for did, k, c, v, dur in rows:
        if pat not in k:
            continue
        by_d.setdefault(did, {"k": k, "us": dur / 1e3})[c] = v
        if c not in names:
            names.append(c)
    lines = ["| dispatch | kernel | us | " + " | ".join(names) + " | fetch MB (x2) | write MB |", "|---|---|---|" + "---|" * (len(names) + 2)]
    for did, d in by_d.items():
        f = d.get("FETCH_SIZE")
        w = d.get("WRITE_SIZE")
        lines.append(f"| {did} | `{d['k'][:70]}` | {d['us']:.1f} | " + " | ".join(f"{d.get(c, 0):.5g}" for c in names) +
                     f" | {'' if f is None else f'{2 * 1024 * f / 1e6:.1f}'} | {'' if w is None else f'{1024 * w / 1e6:.1f}'} |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
