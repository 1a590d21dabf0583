#!/usr/bin/env python3
"""HBM traffic per launch of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
bytes of wide (16 B/lane) coalesced streaming reads -> x2; WRITE_SIZE is exact for 16 B/lane streaming stores;
other access patterns are to be calibrated on a known byte count (FETCH_FACTOR_1 below).
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


# The x2 FETCH correction is for WIDE coalesced streaming reads (64 lanes x 16 B = 1 KB contiguous).  The MFMA conv
# kernels stage their operands with LDS-direct loads whose wave-instruction covers 8 rows x 128 B (rows K*4 bytes
# apart): calibrated in THIS access pattern as the guide asks (profiles/r02b: conv1x1_pipe_kernel, residual variant --
# known algorithmic reads 402 MB per launch, FETCH_SIZE raw 409 MB; no-residual variant 295 MB vs 280 MB; WRITE_SIZE
# 268.4 MB vs 268.2 MB), FETCH_SIZE is exact there, so those kernels get factor 1.  (Round 1 applied x2 to them and
# reported 1.40x "wasted" traffic that was not there.)
FETCH_FACTOR_1 = ("conv_mfma_kernel", "conv1x1_pipe_kernel")


def fetch_factor(kernel_name):
    return 1 if any(t in kernel_name for t in FETCH_FACTOR_1) else 2


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
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); gfx950 FETCH correction x2 for wide "
                         "coalesced streaming reads, x1 (calibrated) for the 128-byte-row LDS-direct staging of the MFMA convs",
               "source_sha256": csrc_hash(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:6]:
        print(f"{k[:80]:80s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch x {v['dispatches']}")


if __name__ == "__main__":
    main()
