#!/bin/bash
# kernel durations of the score block alone (development aid): rocprofv3 kernel trace of scripts/gpu_score.py
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_score_$1
rm -rf $out
NOFORK=1 REPS=20 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_score.py > $out.log 2>&1
grep "score block\|host time" $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    print(r['Name'][:48], r['Calls'], round(float(r['AverageNs'])), r['MinNs'], r['MaxNs'])
PY
