#!/usr/bin/env python3
"""gpurun_out/prof_r05 (scripts/gpu_profile_r05.sh) -> profiles/r05_*   (development aid)
gpurun merges every call's output into the same local directory, so a rocprofv3 run directory may hold files of earlier calls: only the
newest run (by modification time) of each directory is kept before summarising."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r05"); DST = os.path.join(ROOT, "profiles")
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
for d in glob.glob(os.path.join(SRC, "**", "runc"), recursive=True):
    runs = {}
    for f in glob.glob(os.path.join(d, "*_*.csv")): runs.setdefault(os.path.basename(f).split("_")[0], []).append(f)
    if not runs: continue
    newest = max(runs, key=lambda r: max(os.path.getmtime(f) for f in runs[r]))
    for r, fs in runs.items():
        if r != newest:
            for f in fs: os.remove(f)
def one(pattern):
    g = glob.glob(os.path.join(SRC, pattern)); assert len(g) == 1, (pattern, g); return g[0]
def last_json(path):
    line = open(path).read().strip().splitlines()[-1]; json.loads(line); return line
summ = [sys.executable, os.path.join(ROOT, "scripts", "pmc_summarize.py")]
tpath = os.path.join(DST, "pmc_traffic.json"); t = json.load(open(tpath))
lease = open(os.path.join(SRC, "lease.txt")).read().split() if os.path.exists(os.path.join(SRC, "lease.txt")) else ["?", "?"]
if os.path.exists(os.path.join(SRC, "trace_default.json")):
    shutil.copy(one("trace_default/runc/*_kernel_stats.csv"), os.path.join(DST, "r05_kernel_stats.csv"))
    shutil.copy(one("trace_driver/runc/*_kernel_stats.csv"), os.path.join(DST, "r05_driver_kernel_stats.csv"))
    for src, dst in (("trace_default.json", "r05_bench_under_rocprof.json"), ("trace_driver.json", "r05_driver_bench_under_rocprof.json")):
        open(os.path.join(DST, dst), "w").write(last_json(os.path.join(SRC, src)) + "\n")
    bench = json.loads(open(os.path.join(DST, "r05_bench_under_rocprof.json")).read())
    alg = bench["hbm"]["algorithmic_bytes_per_launch"]
    subprocess.check_call(summ + [os.path.join(SRC, "pmc_alone"), os.path.join(DST, "r05_pmc_summary.json"), str(alg)], stdout=subprocess.DEVNULL)
    subprocess.check_call(summ + [os.path.join(SRC, "pmc_adopt"), os.path.join(DST, "r05_pmc_adoption_on_summary.json")], stdout=subprocess.DEVNULL)
    subprocess.check_call(summ + [os.path.join(SRC, "pmc_load"), os.path.join(DST, "r05_pmc_load_summary.json")], stdout=subprocess.DEVNULL)
    s = json.load(open(os.path.join(DST, "r05_pmc_summary.json")))
    t["tum"] = {"kernel": "cvo_align_kernel", "shape": "tum", "pairs": 64,
                "hbm_bytes_per_launch": s["hbm_bytes_per_launch_corrected"], "hbm_bytes_per_launch_uncorrected": s["hbm_bytes_per_launch_uncorrected"],
                "valu_wave_instructions_per_launch": s["per_launch"]["SQ_INSTS_VALU"],
                "valu_half_rate_share": s["valu_classes"]["half_rate_share"], "valu_transcendental_share": s["valu_classes"]["transcendental_share"],
                "source": f"profiles/r05_pmc_summary.json (rocprofv3 --pmc passes of scripts/pmc_run.sh at commit {commit}, lease {lease[0]} {lease[-1]}: 64 pairs, 3072 points, one workgroup per pair, one step in flight, adoption off: the work of a launch)",
                "commit": commit}
    for row in csv.DictReader(open(os.path.join(DST, "r05_kernel_stats.csv"))):
        if "cvo_align" in row["Name"]: print("default run:", row["Calls"], "launches, average", float(row["AverageNs"]) * 1e-6, "ms; bench under the profiler:", round(bench["value"]), "alignments/s, kernel_ms", round(bench["roofline"]["kernel_ms"], 2))
    print(json.dumps(t["tum"], indent=1)); print(json.dumps(s.get("sq_ratios"), indent=1))
    a = json.load(open(os.path.join(DST, "r05_pmc_adoption_on_summary.json")))
    print("VALU wave-instructions per launch: adoption off", s["per_launch"]["SQ_INSTS_VALU"], "on", a["per_launch"].get("SQ_INSTS_VALU"))
    l = json.load(open(os.path.join(DST, "r05_pmc_load_summary.json")))
    print("under load:", l.get("sq_ratios"), "LDS conflict / active", l["per_launch"].get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, l["per_launch"].get("SQ_LDS_IDX_ACTIVE", 1)))
    print("alone: LDS conflict / active", s["per_launch"].get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s["per_launch"].get("SQ_LDS_IDX_ACTIVE", 1)))
if os.path.exists(os.path.join(SRC, "eth_trace.json")):
    shutil.copy(one("eth_trace/runc/*_kernel_stats.csv"), os.path.join(DST, "r05_eth3d_kernel_stats.csv"))
    open(os.path.join(DST, "r05_eth3d_bench_under_rocprof.json"), "w").write(last_json(os.path.join(SRC, "eth_trace.json")) + "\n")
    eb = json.loads(open(os.path.join(DST, "r05_eth3d_bench_under_rocprof.json")).read())
    subprocess.check_call(summ + [os.path.join(SRC, "eth_pmc"), os.path.join(DST, "r05_eth3d_pmc_summary.json"), str(eb["hbm"]["algorithmic_bytes_per_launch"])], stdout=subprocess.DEVNULL)
    s = json.load(open(os.path.join(DST, "r05_eth3d_pmc_summary.json")))
    t["eth3d"] = {"kernel": "cvo_align_kernel", "shape": "eth3d", "pairs": 64,
                  "hbm_bytes_per_launch": s["hbm_bytes_per_launch_corrected"], "hbm_bytes_per_launch_uncorrected": s["hbm_bytes_per_launch_uncorrected"],
                  "valu_wave_instructions_per_launch": s["per_launch"]["SQ_INSTS_VALU"],
                  "valu_half_rate_share": s["valu_classes"]["half_rate_share"], "valu_transcendental_share": s["valu_classes"]["transcendental_share"],
                  "source": f"profiles/r05_eth3d_pmc_summary.json (rocprofv3 --pmc passes of scripts/pmc_run.sh at commit {commit}, lease {lease[0]} {lease[-1]}: 64 pairs, ~9.3 k points, four workgroups per pair, one step in flight)",
                  "commit": commit}
    print("eth3d:", round(eb["value"]), "alignments/s;", json.dumps(t["eth3d"], indent=1))
if os.path.isdir(os.path.join(SRC, "eth_pmc_masked")) and os.path.exists(os.path.join(DST, "r05_eth3d_bench_under_rocprof.json")):
    eb = json.loads(open(os.path.join(DST, "r05_eth3d_bench_under_rocprof.json")).read())
    subprocess.check_call(summ + [os.path.join(SRC, "eth_pmc_masked"), os.path.join(DST, "r05_eth3d_pmc_masked_entry_loads_summary.json"), str(eb["hbm"]["algorithmic_bytes_per_launch"])], stdout=subprocess.DEVNULL)
    m = json.load(open(os.path.join(DST, "r05_eth3d_pmc_masked_entry_loads_summary.json")))
    print("eth3d, experiment build with masked entry loads: traffic per launch", m["hbm_bytes_per_launch_corrected"], "uncorrected", m["hbm_bytes_per_launch_uncorrected"])
json.dump(t, open(tpath, "w"), indent=1)
