#!/usr/bin/env python3
"""HBM traffic per launch of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
bytes of wide (16 B/lane) coalesced streaming reads -> x2; WRITE_SIZE is exact for 16 B/lane streaming stores;
other access patterns are to be calibrated on a known byte count (FETCH_FACTOR_1 below).
Usage: pmc_traffic.py fetch_results.db write_results.db out.json [workload]"""
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


# FETCH_SIZE on gfx950 reports HALF the bytes fetched (128-byte requests tallied at 64 B): the guide says so for wide
# 16-B-per-lane reads "global_load and buffer_load ... lds alike", and scripts/calibrate/calib.hip confirms it for THIS
# code's access patterns on a 1 GiB buffer read exactly once (no reuse, > Infinity Cache; profiles/r03_calib_pmc_fetch.md):
# wide coalesced global loads, LDS-direct loads of 8 rows x 128 B at row pitches 128 / 512 / 2048 B, and the same
# addresses through register loads ALL report 0.5 GiB for 1 GiB read; WRITE_SIZE reports 1 GiB for 1 GiB written.
# So the factor is 2 for every kernel.  (Round 2 used x1 for the MFMA convs, "calibrated" on the pipelined 1x1 kernel
# under the assumption that it reads its activations once from HBM -- it does not: with x2 that kernel moves about
# twice its algorithmic read bytes, which is an over-fetch to fix, not a counter property.)


def fetch_factor(kernel_name):
    return 2


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        n, fk = f[k]
        _, wk = w.get(k, (n, 0.0))
        c = fetch_factor(k)
        out[k] = {"dispatches": n, "fetch_size_raw_bytes_per_launch": 1024 * fk / n, "fetch_correction": c,
                  "fetch_bytes_per_launch": c * 1024 * fk / n, "write_bytes_per_launch": 1024 * wk / n,
                  "hbm_bytes_per_launch": (c * 1024 * fk + 1024 * wk) / n}
    from bench import csrc_hash      # bench.py only reports `traffic` from a pass taken on the sources it was built from
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); gfx950 FETCH correction x2 for every "
                         "kernel (calibrated on no-reuse 1 GiB copies in this code's access patterns: "
                         "profiles/r03_calib_pmc_fetch.md, scripts/calibrate/calib.hip)",
               "workload": sys.argv[4] if len(sys.argv) > 4 else "resnext50_full_b8_1024",
               "source_sha256": csrc_hash(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:6]:
        print(f"{k[:80]:80s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch x {v['dispatches']}")


if __name__ == "__main__":
    main()
