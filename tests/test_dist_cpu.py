"""The N > 1 path on CPU: world_size-2 (and 3) gloo groups run the window-sharding logic of
halo2_liam_eagen_msm_amd.dist -- window ownership, ONE all-gather of the per-window records,
host Horner on every rank -- with oracle-built records standing in for the GPU partials."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import hostref
    from halo2_liam_eagen_msm_amd import dist as ldist
    from oracle import cref, pyref
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        curve = pyref.BN254_G1
        rng = pyref.SplitMix64(4242)
        n = 30
        pts = pyref.gen_points(curve, rng, n)
        if kind == "msm":
            sc = pyref.gen_scalars_full(rng, n, curve.order)
            W, rec, recs = hostref.msm_records(curve, sc, pts)
            w0, w1 = ldist.window_range(W, world, rank)
            local = np.frombuffer(b"".join(recs[w0:w1]), np.uint8)
            allrec = ldist.all_gather_records(local, W, rec, world, rank)
            out = hostref.msm_combine(curve, n, allrec.tobytes())
            ok = cref.jac_to_canonical(curve.cid, out) == curve.canonical(curve.msm_naive(sc, pts))
        else:
            sc = pyref.gen_scalars_half(rng, n, curve.order)
            d, rec, recs = hostref.lhs_records(curve, sc, pts, 16)
            p0, p1 = ldist.window_range(d, world, rank)
            local = np.frombuffer(b"".join(recs[p0:p1]), np.uint8)
            allrec = ldist.all_gather_records(local, d, rec, world, rank)
            carry, carries = hostref.lhs_combine(curve, 16, allrec.tobytes())
            ec, ecs = pyref.lhs_msm(curve, sc, pts, 16)
            ok = cref.jac_to_canonical(curve.cid, carry) == curve.canonical(ec) and all(
                cref.jac_to_canonical(curve.cid, carries[i]) == curve.canonical(ecs[i]) for i in range(d))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind", ["msm", "lhs"])
def test_window_sharded_combine_gloo(world, kind):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok for _, ok in res)


def test_window_range_partition():
    from halo2_liam_eagen_msm_amd.dist import window_range
    for W in (1, 16, 17, 33, 56):
        for G in (1, 2, 3, 4, 8):
            spans = [window_range(W, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == W
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
