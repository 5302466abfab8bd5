"""`python3 bench.py --gpus N` invoked bare (the driver's N = 1 command shape; no torch.distributed.run): the script starts its
own ranks — fresh child processes, never an exec of a process that touched the GPU — or says plainly why it cannot."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bare_multi_gpu_command_without_a_gpu_says_so():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert "needs 2 MI355X devices" in r.stderr and "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""


def test_mismatched_world_size_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="1", RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_bare_multi_gpu_command_rehearsed_on_one_device(gpu):
    """Two ranks on cuda:0 through the bare command (gloo carries the bootstrap; the collectives are the C ABI's one-shot
    kernels): rank 0's line comes back through the parent, n_gpus = 2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--prefill", "128",
                        "--decode", "4"], capture_output=True, text=True, timeout=900, env=_env(LFAMD_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "invalid" not in d
