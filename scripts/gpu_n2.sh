#!/bin/bash
# rehearsal of bench.py's N > 1 path on a one-GPU box THROUGH THE BARE INVOCATION (`python bench.py --gpus N`, no launcher: bench.py starts its ranks itself):
# ranks share the GPU, each runs the real HIP path on its block, the padded record blocks of the C ABI (cvo_batch_padded_records) travel through gloo (RCCL
# refuses two ranks on one device).  Even and uneven blocks (7 pairs over 2 ranks, 10 over 3), a rank without pairs (2 over 3).  One line under
# torch.distributed.run too (the driver's form).
cd $GRAFT_REPO_ROOT
O=gpurun_out/n2_r04; mkdir -p $O
run() { # tag ranks extra-args...
  tag=$1; n=$2; shift; shift
  CVO_BENCH_BACKEND=gloo CVO_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus $n --steps 8 --warmup 2 --streams 4 "$@" > $O/$tag.json 2> $O/$tag.err
  echo "bare, ranks $n $*: rc=$? $(python -c "import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print(round(d['value']), 'n_gpus', d['n_gpus'], 'gather', d['gather'], 'ranks_in_communicator', d['ranks_in_communicator'], 'timed_region_s', round(d['timed_region_s'], 4))" 2>&1) | $(grep gathered $O/$tag.err | tail -1)"
}
run even2 2
run uneven7 2 --total-pairs 7
run uneven10 3 --total-pairs 10
run short2 3 --total-pairs 2
CVO_BENCH_BACKEND=gloo CVO_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 2 --streams 4 > $O/torchrun2.json 2> $O/torchrun2.err
echo "torch.distributed.run, ranks 2: rc=$? $(python -c "import json; d=json.loads(open('$O/torchrun2.json').read().strip().splitlines()[-1]); print(round(d['value']), 'n_gpus', d['n_gpus'], 'gather', d['gather'])" 2>&1)"
