#!/usr/bin/env python3
"""HBM traffic per launch of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
bytes of wide (16 B/lane) coalesced streaming reads -> x2; WRITE_SIZE is exact for 16 B/lane streaming stores.
Usage: pmc_traffic.py fetch_results.db write_results.db out.json"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {k: (n, v) for k, n, v in rows}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        n, fk = f[k]
        _, wk = w.get(k, (n, 0.0))
        out[k] = {"dispatches": n, "fetch_bytes_per_launch_x2": 2 * 1024 * fk / n, "write_bytes_per_launch": 1024 * wk / n,
                  "hbm_bytes_per_launch": (2 * 1024 * fk + 1024 * wk) / n}
    from bench import csrc_hash      # bench.py only reports `traffic` from a pass taken on the sources it was built from
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 gfx950 correction",
               "source_sha256": csrc_hash(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:6]:
        print(f"{k[:80]:80s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch x {v['dispatches']}")


if __name__ == "__main__":
    main()
