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
        elif kind == "empty-range":
            # world > number of windows: the last rank owns nothing and still takes part in the all-gather
            W, rec = 2, 128
            w0, w1 = ldist.window_range(W, world, rank)
            local = np.full((w1 - w0) * rec, 17 + w0, np.uint8)
            allrec = ldist.all_gather_records(local, W, rec, world, rank)
            exp = np.concatenate([np.full(rec, 17 + w, np.uint8) for w in range(W)])
            ok = np.array_equal(allrec, exp)
        elif kind == "points":
            # the alternative partition: pairs split across ranks, one Jacobian partial per rank
            # (oracle-built, arbitrary Z), ONE all-gather, product-library host sum on every rank
            from halo2_liam_eagen_msm_amd import api
            sc = pyref.gen_scalars_full(rng, n, curve.order)
            a, b = ldist.point_range(n, world, rank)
            part = curve.msm_naive(sc[a:b], pts[a:b])
            z = 1 + pyref.SplitMix64(77 + rank).next256() % (curve.fp - 1)
            local = np.frombuffer(curve.affine_to_jacobian_raw(part, z), np.uint8)
            allj = ldist.all_gather_fixed(local, world)
            out = api.jacobian_sum(curve.cid, allj.view(np.uint64).reshape(world, 12))
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


@pytest.mark.parametrize("world,kind", [(2, "msm"), (3, "msm"), (2, "lhs"), (3, "lhs"), (2, "points"), (3, "points"), (3, "empty-range")])
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


def test_point_range_partition():
    from halo2_liam_eagen_msm_amd.dist import point_range
    for n in (0, 1, 7, 1 << 20, (1 << 24) + 5):
        for G in (1, 2, 3, 8):
            spans = [point_range(n, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
