#!/bin/bash
# round 4: the driver's command pair by pair (scripts/gpu_timeline.py), adoption on / off, and more batch objects in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_timeline; mkdir -p $O
hostname > $O/lease.txt
timeout -k 10 200 python scripts/gpu_timeline.py --repeat 2 --out $O/on.json > $O/on.txt 2>&1; echo "on rc=$?"; grep -E "^repeat|idle" $O/on.txt
timeout -k 10 200 python scripts/gpu_timeline.py --repeat 2 --no-adoption --out $O/off.json > $O/off.txt 2>&1; echo "off rc=$?"; grep -E "^repeat|idle" $O/off.txt
for s in 12 16 20; do
  GPU_MAX_HW_QUEUES=$s timeout -k 10 200 python scripts/gpu_timeline.py --repeat 2 --streams $s > $O/on_s$s.txt 2>&1; echo "streams $s rc=$?"; grep -E "^repeat|idle" $O/on_s$s.txt
done
