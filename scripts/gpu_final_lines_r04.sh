#!/bin/bash
# round 4, final state: smoke and the three complete bench lines (the GPU suite ran in gpu_final_bench_r04.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_final; mkdir -p $O; hostname > $O/lease.txt; date -u +%FT%TZ >> $O/lease.txt
timeout -k 10 120 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout -k 10 400 python bench.py --shape eth3d > $O/bench_eth3d.json 2> $O/bench_eth3d.err; echo "eth3d rc=$?"
python - <<'PY'
import json
for f in ("bench_driver", "bench_default", "bench_eth3d"):
    d = json.loads(open(f"gpurun_out/r4_final/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3), "GB/s", round(d["roofline"].get("achieved_hbm_GBps", 0) or 0, 0), "score", round(d["with_score_block"]["value"], 0), round(d["with_score_block"]["value"] / d["value"], 3), "upload", round(d["with_host_upload"]["value"], 0), round(d["with_host_upload"]["value"] / d["value"], 3),
          "distinct", d.get("distinct_pairs") and round(d["distinct_pairs"]["value"] / d["value"], 3), "parity", d["parity"]["max_rot_err_rad"], d["parity"]["max_trans_err_m"], "cpu", round(d["cpu_baseline"]["value"], 1),
          "lane-instr/nz", round(d["work"]["valu_lane_instructions_per_nonzero"], 1), "latency", {k: round(v, 3) for k, v in d["latency"].items() if isinstance(v, float)})
PY
