"""CPU: bench.py's own multi-rank launcher (`--gpus N` without torchrun).  The parent must decide BEFORE touching a GPU,
fail loudly when the box has fewer than N devices, and never print a line that claims N GPUs it did not use."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_on_a_box_with_fewer_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box really has 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_gpus_8_on_a_one_gpu_box_exits_2_before_anything_runs():
    """VERDICT r3 #5d: the driver's 8-GPU command on a box without 8 devices must end with exit code 2 and no result line."""
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box really has 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 8" in r.stderr and '"value"' not in r.stdout


def test_an_nccl_failure_exits_non_zero_and_prints_no_value():
    """RCCL over xGMI is the judged path at N > 1: when it cannot come up (forced here) every rank exits non-zero and NOTHING is printed on
    stdout -- round 3 fell back to gloo and printed a host-staged number under the same keys."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SPMV_BENCH_ONE_DEVICE="1", SPMV_BENCH_FORCE_NCCL_FAIL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "4096"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0, r.stdout[-500:]
    assert "nccl (RCCL) initialisation failed" in r.stderr
    assert '"value"' not in r.stdout and '"metric"' not in r.stdout
