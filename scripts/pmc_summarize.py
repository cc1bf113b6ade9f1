#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter CSVs of scripts/pmc_run.sh over the launches of cvo_align_kernel.

usage: pmc_summarize.py <pmc dir> <out json> [<algorithmic bytes per launch>]
FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE counts half the bytes of wide
(16 B/lane) coalesced reads, so the read side is reported raw and doubled (upper bound)."""
import csv, glob, json, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
per = defaultdict(lambda: defaultdict(float))          # counter -> dispatch -> value summed over its rows (XCDs / instances)
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if "cvo_align_kernel" not in row["Kernel_Name"]:
                continue
            per[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
res = {k: sum(v.values()) / len(v) for k, v in per.items()}
n = {k: len(v) for k, v in per.items()}
summary = {"kernel": "cvo_align_kernel", "launches_averaged": n, "per_launch": res, "FETCH_SIZE_unit": "KB", "WRITE_SIZE_unit": "KB"}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    raw = (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
    cor = (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
    summary.update({"hbm_bytes_per_launch_uncorrected": raw, "hbm_bytes_per_launch_corrected": cor,
                    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); "
                                  "read side doubled = upper bound, WRITE_SIZE taken as is"})
if alg:
    summary["algorithmic_bytes_per_launch"] = alg
if "SQ_WAVE_CYCLES" in res:
    w = res["SQ_WAVE_CYCLES"]
    summary["sq_ratios"] = {k: res[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES") if k in res}
if "SQ_INSTS_VALU" in res and "SQ_INSTS_VALU_FMA_F64" in res and "SQ_INSTS_VALU_CVT" in res:
    # instruction classes by issue cost (MI355X_MICROARCH.md rates, scripts/micro/valu_rate.hip): f64 arithmetic, conversions and 64-bit
    # integer ops issue at half the f32 rate, transcendentals at a quarter.  (All conversions are counted in the half-rate class; the
    # kernel's are f32 <-> f64 with few exceptions.)
    half = sum(res.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT64"))
    trans = sum(res.get(k, 0.0) for k in ("SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_TRANS_F64"))
    summary["valu_classes"] = {"half_rate_share": half / res["SQ_INSTS_VALU"], "transcendental_share": trans / res["SQ_INSTS_VALU"]}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1))
