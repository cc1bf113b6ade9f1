#!/usr/bin/env python3
"""gpurun_out/prof_r02_eth (scripts/gpu_profile_eth.sh) -> profiles/r02_eth3d_*   (development aid; see collect_profiles_r02.py)"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02_eth"); DST = os.path.join(ROOT, "profiles")
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
for d in glob.glob(os.path.join(SRC, "**", "runc"), recursive=True):
    runs = {}
    for f in glob.glob(os.path.join(d, "*_*.csv")): runs.setdefault(os.path.basename(f).split("_")[0], []).append(f)
    newest = max(runs, key=lambda r: max(os.path.getmtime(f) for f in runs[r]))
    for r, fs in runs.items():
        if r != newest:
            for f in fs: os.remove(f)
g = glob.glob(os.path.join(SRC, "trace/runc/*_kernel_stats.csv")); assert len(g) == 1
shutil.copy(g[0], os.path.join(DST, "r02_eth3d_kernel_stats.csv"))
line = open(os.path.join(SRC, "trace.json")).read().strip().splitlines()[-1]; bench = json.loads(line)
open(os.path.join(DST, "r02_eth3d_bench_under_rocprof.json"), "w").write(line + "\n")
if "--traces-only" in sys.argv:
    print("traces only:", round(bench["value"]), "alignments/s,", "frac", round(bench["roofline"]["frac"], 3)); sys.exit(0)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_summarize.py"), os.path.join(SRC, "pmc"), os.path.join(DST, "r02_eth3d_pmc_summary.json"),
                       str(bench["hbm"]["algorithmic_bytes_per_launch"])], stdout=subprocess.DEVNULL)
s = json.load(open(os.path.join(DST, "r02_eth3d_pmc_summary.json")))
tpath = os.path.join(DST, "pmc_traffic.json"); t = json.load(open(tpath))
t["eth3d"] = {"kernel": "cvo_align_kernel", "shape": "eth3d", "pairs": 64,
              "hbm_bytes_per_launch": s["hbm_bytes_per_launch_corrected"], "hbm_bytes_per_launch_uncorrected": s["hbm_bytes_per_launch_uncorrected"],
              "valu_wave_instructions_per_launch": s["per_launch"]["SQ_INSTS_VALU"],
              "valu_half_rate_share": s["valu_classes"]["half_rate_share"], "valu_transcendental_share": s["valu_classes"]["transcendental_share"],
              "source": f"profiles/r02_eth3d_pmc_summary.json (rocprofv3 --pmc passes of scripts/pmc_run.sh at commit {commit}: 64 pairs, ~9.3 k points, four workgroups per pair, one step in flight)",
              "commit": commit}
json.dump(t, open(tpath, "w"), indent=1)
for row in csv.DictReader(open(os.path.join(DST, "r02_eth3d_kernel_stats.csv"))):
    if "cvo_align" in row["Name"]: print("eth3d run:", row["Calls"], "launches, average", float(row["AverageNs"]) * 1e-6, "ms; bench under the profiler:", round(bench["value"]), "alignments/s, kernel_ms", round(bench["roofline"]["kernel_ms"], 2))
print(json.dumps(t["eth3d"], indent=1))
