#!/bin/bash
# rehearsal of bench.py's N > 1 path on a one-GPU box: ranks share the GPU, each runs the real HIP path on its block, the padded record blocks of the
# C ABI (cvo_batch_padded_records) travel through gloo (RCCL refuses two ranks on one device).  Even and uneven blocks (7 pairs over 2 ranks, 10 over 3).
cd $GRAFT_REPO_ROOT
run() { # ranks extra-args...
  n=$1; shift
  CVO_BENCH_BACKEND=gloo CVO_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 8 --warmup 2 --streams 4 "$@" > gpurun_out/n2.json 2> gpurun_out/n2.err
  echo "ranks $n $*: rc=$? $(python -c "import json; d=json.loads(open('gpurun_out/n2.json').read().strip().splitlines()[-1]); print(round(d['value']), d['config']['collective'][:60])" 2>&1) | $(grep gathered gpurun_out/n2.err | tail -1)"
}
run 2
run 2 --total-pairs 7
run 3 --total-pairs 10
run 3 --total-pairs 2
