#!/usr/bin/env python3
"""gpurun_out/prof_r02 (scripts/gpu_profile_r02.sh) -> profiles/r02_*   (development aid)
gpurun merges every call's output into the same local directory, so each rocprofv3 run directory may hold files of earlier calls:
only the newest run (by modification time) of each directory is kept before summarising."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02"); DST = os.path.join(ROOT, "profiles")
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
for d in glob.glob(os.path.join(SRC, "**", "runc"), recursive=True):
    files = glob.glob(os.path.join(d, "*_*.csv"))
    runs = {}
    for f in files: runs.setdefault(os.path.basename(f).split("_")[0], []).append(f)
    newest = max(runs, key=lambda r: max(os.path.getmtime(f) for f in runs[r]))
    for r, fs in runs.items():
        if r != newest:
            for f in fs: os.remove(f)
def one(pattern):
    g = glob.glob(os.path.join(SRC, pattern)); assert len(g) == 1, (pattern, g); return g[0]
shutil.copy(one("trace_default/runc/*_kernel_stats.csv"), os.path.join(DST, "r02_kernel_stats.csv"))
shutil.copy(one("trace_driver/runc/*_kernel_stats.csv"), os.path.join(DST, "r02_driver_kernel_stats.csv"))
for src, dst in (("trace_default.json", "r02_bench_under_rocprof.json"), ("trace_driver.json", "r02_driver_bench_under_rocprof.json")):
    line = open(os.path.join(SRC, src)).read().strip().splitlines()[-1]; json.loads(line)
    open(os.path.join(DST, dst), "w").write(line + "\n")
shutil.copy(os.path.join(SRC, "valu_issue_microbench.txt"), os.path.join(DST, "r02_valu_issue_microbench.txt"))
bench = json.loads(open(os.path.join(DST, "r02_bench_under_rocprof.json")).read())
alg = bench["hbm"]["algorithmic_bytes_per_launch"]
if "--traces-only" in sys.argv:      # the PMC passes (and the commit they were taken at) stay as they are
    print("traces only:", round(bench["value"]), "alignments/s,", "frac", round(bench["roofline"]["frac"], 3)); sys.exit(0)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_summarize.py"), os.path.join(SRC, "pmc_alone"), os.path.join(DST, "r02_pmc_summary.json"), str(alg)], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_summarize.py"), os.path.join(SRC, "pmc_load"), os.path.join(DST, "r02_pmc_load_summary.json")], stdout=subprocess.DEVNULL)
s = json.load(open(os.path.join(DST, "r02_pmc_summary.json")))
tpath = os.path.join(DST, "pmc_traffic.json"); t = json.load(open(tpath))
t["tum"] = {"kernel": "cvo_align_kernel", "shape": "tum", "pairs": 64,
            "hbm_bytes_per_launch": s["hbm_bytes_per_launch_corrected"], "hbm_bytes_per_launch_uncorrected": s["hbm_bytes_per_launch_uncorrected"],
            "valu_wave_instructions_per_launch": s["per_launch"]["SQ_INSTS_VALU"],
            "valu_half_rate_share": s["valu_classes"]["half_rate_share"], "valu_transcendental_share": s["valu_classes"]["transcendental_share"],
            "source": f"profiles/r02_pmc_summary.json (rocprofv3 --pmc passes of scripts/pmc_run.sh at commit {commit}: 64 pairs, 3072 points, one workgroup per pair, one step in flight, adoption off: the work of a launch)",
            "commit": commit}
json.dump(t, open(tpath, "w"), indent=1)
for row in csv.DictReader(open(os.path.join(DST, "r02_kernel_stats.csv"))):
    if "cvo_align" in row["Name"]: print("default run:", row["Calls"], "launches, average", float(row["AverageNs"]) * 1e-6, "ms; bench under the profiler:", round(bench["value"]), "alignments/s, kernel_ms", round(bench["roofline"]["kernel_ms"], 2))
print(json.dumps(t["tum"], indent=1)); print(json.dumps(s.get("sq_ratios"), indent=1))
