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


CHILD_ABI = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from halo2_liam_eagen_msm_amd import Context, Node, comm_unique_id
from oracle import cref
# (1) one process per GPU form: unique id -> comm_init -> collective entries, world 1
ctx = Context(0)
uid = comm_unique_id(); assert len(uid) == 128
ctx.comm_init(uid, 1, 0); assert ctx.comm_info() == (1, 0)
n = 5000
pts = cref.gen_points(0, 71, n); sc = cref.gen_scalars(0, 72, n)
ds, dp = ctx.to_device(sc), ctx.to_device(pts)
exp = cref.jac_to_canonical(0, cref.best_multiexp(0, sc, pts, 8))
assert cref.jac_to_canonical(0, ctx.msm_sharded_device(0, ds.ptr, dp.ptr, n)) == exp
assert cref.jac_to_canonical(0, ctx.msm_device(0, ds.ptr, dp.ptr, n)) == exp
sch = cref.gen_scalars(0, 73, n, half=True); dsh = ctx.to_device(sch)
ec, ecs = cref.lhs_msm(0, sch, cref.aff_to_jac(0, pts), 16)
c, cs = ctx.lhs_msm_sharded_device(0, dsh.ptr, dp.ptr, n, 16)
assert cref.jac_to_canonical(0, c) == cref.jac_to_canonical(0, ec)
assert all(cref.jac_to_canonical(0, cs[i]) == cref.jac_to_canonical(0, ecs[i]) for i in range(cs.shape[0]))
# (1b) failures stay collective-safe: a non-canonical scalar is reported from the gathered status slots, and a rank whose
# own pipeline cannot start (here: rejected input points) still enters the exchange with a stand-in buffer carrying its status
from halo2_liam_eagen_msm_amd import api
bad = sc.copy(); bad[1234] = 0xff
dbad = ctx.to_device(bad)
try:
    ctx.msm_sharded_device(0, dbad.ptr, dp.ptr, n); raise SystemExit("expected ScalarOutOfRange")
except api.ScalarOutOfRange as e:
    assert e.index == 1234, e.index
offcurve = pts.copy(); offcurve[77, 0] ^= 1
doff = ctx.to_device(offcurve)
ctx.set_option("validate_points", 1)
for call in (lambda: ctx.msm_sharded_device(0, ds.ptr, doff.ptr, n), lambda: ctx.lhs_msm_sharded_device(0, dsh.ptr, doff.ptr, n, 16)):
    try:
        call(); raise SystemExit("expected a bad-argument status")
    except api.LemsmError as e:
        assert e.status == 6, e.status          # LEMSM_ERR_BAD_ARG
ctx.set_option("validate_points", 0)
assert ctx.comm_info() == (1, 0)          # the communicator survived: the failing rank took part in the exchange
assert cref.jac_to_canonical(0, ctx.msm_sharded_device(0, ds.ptr, dp.ptr, n)) == exp
ctx.comm_destroy(); assert ctx.comm_info() == (0, 0)
ctx.close()
# (2) one process, N GPUs form (N = 1 here): contexts + communicators + threads inside the library
node = Node(ndev=1); assert node.size == 1
node.set_bases(0, pts)
assert cref.jac_to_canonical(0, node.msm(sc)) == exp
c, cs = node.lhs_msm(sch, 16)
assert cref.jac_to_canonical(0, c) == cref.jac_to_canonical(0, ec)
node.close()
print("rccl abi ok")
"""


def test_rccl_behind_c_abi_world1():
    """lemsm_comm_* / lemsm_*_sharded_device / lemsm_node_*: RCCL bound by dlopen inside liblemsm.so, communicator of
    one rank (all a one-GPU box allows), ncclAllGather of the raw device records, results equal the oracle's"""
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD_ABI, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl abi ok" in r.stdout, r.stdout + r.stderr
