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
