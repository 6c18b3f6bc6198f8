"""The sharded HIP path with a REAL process group: two processes share the one GPU of the box, each drives a HIP engine
for half of the bodies, and the per-step collectives (in-place all_gather_into_tensor of positions, all_to_all_single of
the symmetric algorithm's j-side sums) run over gloo on device tensors — RCCL itself refuses two ranks on one device, so
only the transport differs from the 8-GPU job.  The result must match a single-context run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_one_gpu_end_to_end():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "two_rank_gloo_gpu.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "two ranks on one GPU over gloo: algorithm symmetric, exchange ranks 2" in out.stdout
