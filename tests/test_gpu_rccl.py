"""RCCL sanity on the GPU box: the backend the N > 1 bench uses ("nccl" == RCCL on ROCm) initialises
and runs the one collective of the path (all_gather_into_tensor of uint8 records) -- with world_size 1,
which is all a one-GPU box allows (RCCL refuses two ranks on one device).  The multi-rank logic itself is
covered on CPU with gloo (tests/test_dist_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from halo2_liam_eagen_msm_amd import dist as ldist
rec = np.arange(5 * 128, dtype=np.uint8)
got = ldist.all_gather_fixed(rec, 1, dev)
assert got.shape == (1, 640) and np.array_equal(got[0], rec)
t = torch.from_numpy(rec.copy()).to(dev); out = torch.empty(640, dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(out, t); torch.cuda.synchronize()
assert np.array_equal(out.cpu().numpy(), rec)
dist.barrier(); dist.destroy_process_group()
print("rccl ok")
"""


def test_rccl_single_rank_all_gather():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(port)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr
