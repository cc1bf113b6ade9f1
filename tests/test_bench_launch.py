"""`bench.py --gpus N` invoked bare -- the way the driver invokes the one-GPU bench -- must start its N ranks itself: a line measured by
ONE rank that says n_gpus = N would void a scaling run.  --dry-launch makes every rank print its block of the step's pairs
(cvo_shard_range, BASELINE config 4: contiguous blocks of the batch, keyframe_graph.cpp:622-731 is the reference's batch source) and exit;
no GPU is touched."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("gpus,total,blocks", [(2, 0, [(0, 64), (64, 64)]), (3, 10, [(0, 4), (4, 3), (7, 3)]), (8, 0, [(64 * r, 64) for r in range(8)])])
def test_bare_invocation_starts_its_ranks(hiplib, gpus, total, blocks):
    out = _run(["--gpus", str(gpus), "--dry-launch"] + (["--total-pairs", str(total)] if total else []))
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == list(range(gpus))                 # N ranks ran, each once
    for l in lines:
        assert l["world"] == gpus and l["launched_by"] == "bench.py" and l["local_rank"] == l["rank"]
        assert (l["first_pair"], l["pairs"]) == blocks[l["rank"]]
        assert l["block_records"] == max(b[1] for b in blocks)
        # what the N > 1 line says about the collective: the communicator's size, who gathers, on which streams, and which RCCL file the C ABI bound
        assert l["ranks_in_communicator"] == gpus and l["gather"] == "rccl" and "one per step in flight" in l["gather_streams"]
        assert "librccl" in l["rccl_library"], l["rccl_library"]
    assert sum(l["pairs"] for l in lines) == (total or 64 * gpus)


def test_the_gather_stream_fallback_is_named_in_the_line(hiplib):
    out = _run(["--gpus", "2", "--dry-launch"], {"CVO_BENCH_GATHER_STREAM": "1"})
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all("the communicator's own" in l["gather_streams"] for l in lines)
    out = _run(["--gpus", "2", "--dry-launch"], {"CVO_BENCH_GATHER": "torch"})
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert all(l["gather"] == "torch" and l["gather_streams"] is None for l in lines)


def test_under_an_external_launcher_it_is_one_rank(hiplib):
    out = _run(["--gpus", "2", "--dry-launch"], {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["launched_by"] == "external launcher"
    # a launcher whose world does not match --gpus is refused (round 3 ran one rank and printed n_gpus 1 here)
    bad = _run(["--gpus", "4", "--dry-launch"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_a_failing_rank_fails_the_launcher(hiplib):
    out = _run(["--gpus", "2", "--dry-launch", "--total-pairs", "-5"])
    assert out.returncode != 0
