"""RCCL on the one GPU a test box has: the one-process-per-GPU code path of bench.py --gpus N and
faiss_amd/distributed.py under `python -m torch.distributed.run --nproc-per-node 1` with the "nccl" backend and the collectives
forced on -- init order against the library's HIP runtime, all_reduce / all_gather / broadcast / gather, the device merge.
Mirrors faiss/gpu/test/test_multi_gpu.py:31-48 (sharded results == unsharded results)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, env_extra, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_rccl_group_of_one_runs_the_sharded_and_replicated_paths():
    out = _torchrun([os.path.join("tests", "rccl_world1_worker.py")], {"BACKEND": "nccl"}, 600)
    assert out["backend"] == "nccl" and out["world"] == 1 and out["ranks_seen"] == 1 and out["devices"] == [0]
    assert out["broadcast_ok"] and out["ivfpq_shards_ok"] and out["flat_replicas_ok"] and out["max_reduce_ok"], out


@pytest.mark.gpu
def test_bench_under_torchrun_counts_its_ranks_through_rccl():
    """bench.py launched the way the driver launches it for N > 1 (here N = 1): the process group is RCCL, `ranks_seen` comes
    from an all_reduce and the Flat leg's results go through the gather even with one rank."""
    out = _torchrun(["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-ivf", "--scale-legs", "", "--no-cpu-baseline"],
                    {}, 900)
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out.get("ranks_seen_via") == "rccl all_reduce", out
    assert out["value"] > 1e5 and out["roofline"]["frac"] > 0.05


def test_group_of_one_protocol_on_gloo():
    """the same worker on the host (gloo, no GPU): forced collectives in a group of one rank"""
    out = _torchrun([os.path.join("tests", "rccl_world1_worker.py")], {"BACKEND": "gloo"}, 300)
    assert out["backend"] == "gloo" and out["ranks_seen"] == 1 and out["gather_ok"] and out["broadcast_ok"], out
