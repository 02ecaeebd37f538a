#!/usr/bin/env python3
"""Benchmark of the MSM witness path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload msm|lhs] [--logn L] [--curve bn254_g1|grumpkin]

One "step" = one full n-point MSM (digits -> bucket sort -> bucket accumulation -> bucket
reduction -> host Horner) over synthetic inputs that are resident in HBM before the timed
region.  Default workload: 2^24-point BN254 G1 MSM with full-width scalars (BASELINE.json
configs[2], the size BASELINE.json's target sentence names); `--workload lhs --logn 20` runs
configs[1] (2^20 points, negabase w=4 i.e. base 16, half-width scalars).  For N > 1 the same
MSM is sharded by Pippenger window over the ranks (strong scaling) with one all-gather of the
per-window partial records over RCCL.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_accum1, the bucket
accumulation) at the ALGORITHMIC 96 B per scalar-point pair against the 8 TB/s HBM peak, from
HIP-event timings taken on the library's own stream; `cpu_baseline` times the oracle's
restatement of halo2 best_multiexp on the host cores over a bounded sample of the same inputs
and checks the GPU result on that sample bit-exactly.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ORDER = {
    "bn254_g1": 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    "grumpkin": 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_ISSUE_CYCLES = 4            # cycles per wave64 VALU instruction per SIMD with >= 2 waves (profiles/r02/valu_rates_microbench.txt)
MAX_CLOCK_GHZ = 2.4              # MI355X peak engine clock (MI355X_MICROARCH.md); the VALU-issue peak is priced at it
BYTES_PER_PAIR = 96              # 32 B scalar + 64 B affine point


def gen_scalars(n: int, modulus: int, seed: int) -> np.ndarray:
    """n x 32 B canonical LE scalars: 256 random bits reduced mod `modulus` (>= 2^253) by
    conditional subtraction, or masked below a small modulus's bit length then rejected-by-subtract."""
    rng = np.random.Generator(np.random.PCG64(seed))
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    bl = modulus.bit_length()
    if bl < 256:
        top = (bl - 1) // 64
        limbs[:, top + 1:] = 0
        limbs[:, top] &= np.uint64((1 << (bl - 64 * top)) - 1)
    m = np.array([(modulus >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    for _ in range(8):
        ge = np.ones(n, bool); decided = np.zeros(n, bool)
        for i in (3, 2, 1, 0):
            gt = limbs[:, i] > m[i]; lt = limbs[:, i] < m[i]
            ge = np.where(~decided & lt, False, ge)
            decided |= gt | lt
        if not ge.any():
            break
        borrow = np.zeros(n, np.uint64)
        for i in range(4):
            a = limbs[:, i]
            t = a - m[i]
            b1 = (a < m[i]).astype(np.uint64)
            t2 = t - borrow
            b2 = (t < borrow).astype(np.uint64)
            limbs[:, i] = np.where(ge, t2, a)
            borrow = b1 | b2
    return limbs.view(np.uint8).reshape(n, 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["msm", "lhs", "lhs_witness"], default="msm",
                    help="msm: best_multiexp (the headline); lhs: the MSM core of compute_lhs_witness; lhs_witness: compute_lhs_witness in full "
                         "(carry + the d divisor witnesses, Grumpkin, SURVEY 8 row f2), one GPU")
    ap.add_argument("--logn", type=int, default=None)
    ap.add_argument("--curve", choices=["bn254_g1", "grumpkin"], default="bn254_g1")
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1,
                    help="msm, one GPU: a step is a BATCH of this many MSMs over the same resident points through lemsm_msm_batch_device (calls pipelined "
                         "over two lanes inside the library); 1 = the headline single call.  A separately labelled workload, never the default")
    ap.add_argument("--cpu-sample-log", type=int, default=None, help="log2 of the cpu_baseline sample size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["rccl-abi", "torch"], default="rccl-abi",
                    help="N > 1 window sharding: the C ABI's own RCCL communicator (lemsm_comm_init / lemsm_*_sharded_device: one "
                         "ncclAllGather of raw device records) or the host-side torch.distributed all-gather of halo2_liam_eagen_msm_amd.dist")
    ap.add_argument("--sharding", choices=["windows", "points"], default="windows",
                    help="N > 1 partition of the MSM: Pippenger windows (north star, default) or pairs (SURVEY 8e alternative)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="lemsm_set_option tuning knob for A/B runs (repeatable); reported in config.options")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    logn = args.logn if args.logn is not None else (24 if args.workload == "msm" else (18 if args.workload == "lhs_witness" else 20))
    n = 1 << logn

    import torch
    import torch.distributed as dist
    # LEMSM_BENCH_DEVICE / LEMSM_BENCH_BACKEND exist only to rehearse the N > 1 code path on a
    # one-GPU box (all ranks on cuda:0, gloo collectives); the driver's runs use LOCAL_RANK + nccl.
    dev_index = int(os.environ.get("LEMSM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("LEMSM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else None

    from halo2_liam_eagen_msm_amd import Context
    from halo2_liam_eagen_msm_amd import dist as ldist
    ctx = Context(dev_index)
    if args.workload == "lhs_witness":
        bench_lhs_witness(args, ctx, n, logn, world, rank, dist, torch, dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    curve = args.curve
    cid = {"bn254_g1": 0, "grumpkin": 1}[curve]
    order = ORDER[curve]

    # ---- synthetic inputs, identical on every rank (points replicated per GPU) ----
    t_in = time.time()
    modulus = order if args.workload == "msm" else math.isqrt(order)
    scalars = gen_scalars(n, modulus, 0x5EED0000 + logn)
    d_scalars = ctx.to_device(scalars)
    # Q = a fixed curve point (generator); P_i = (i+1) Q generated on the device
    q = np.zeros(8, np.uint64)
    fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47 if cid == 0 else ORDER["bn254_g1"]
    gx, gy = (1, 2) if cid == 0 else (1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C)
    q[:4] = np.frombuffer(((gx << 256) % fp).to_bytes(32, "little"), np.uint64)
    q[4:] = np.frombuffer(((gy << 256) % fp).to_bytes(32, "little"), np.uint64)
    d_points = ctx.gen_walk(cid, q, n)   # P_i = (i+1) Q: sum s_i P_i == (sum s_i (i+1)) Q, the closed form verify_timed_result checks
    t_in = time.time() - t_in

    exchange = "none"
    if world > 1 and args.sharding == "windows":
        exchange = init_exchange(ctx, args.exchange, world, rank, dist, torch, coll_dev)
        if exchange == "torch" and args.workload == "msm":
            # 16 windows split evenly over 2/4/8 ranks; the single-GPU default at 2^24 (15 windows of 17 bits) does not
            # (the C ABI's sharded entry pins this itself)
            ctx.set_option("window_bits", 16)

    for kv in args.option:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))

    def step():
        if args.workload == "msm":
            if world == 1 and args.batch > 1:
                return ctx.msm_batch_device(cid, [d_scalars.ptr] * args.batch, d_points.ptr, n)[-1]
            if world == 1:
                return ctx.msm_device(cid, d_scalars.ptr, d_points.ptr, n)
            if args.sharding == "points":
                return ldist.sharded_msm_by_points(ctx, cid, d_scalars.ptr, d_points.ptr, n, world, rank, coll_dev)
            if exchange == "rccl-abi":
                return ctx.msm_sharded_device(cid, d_scalars.ptr, d_points.ptr, n)
            return ldist.sharded_msm(ctx, cid, d_scalars.ptr, d_points.ptr, n, world, rank, coll_dev)
        if world == 1:
            return ctx.lhs_msm_device(cid, d_scalars.ptr, d_points.ptr, n, args.base, True)[0]
        if exchange == "rccl-abi":
            return ctx.lhs_msm_sharded_device(cid, d_scalars.ptr, d_points.ptr, n, args.base, True)[0]
        return ldist.sharded_lhs_msm(ctx, cid, d_scalars.ptr, d_points.ptr, n, args.base, world, rank, coll_dev)[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    acc_ms = 0.0; tot_ms = 0.0; launches = 0; clock_sum = 0.0; clock_n = 0
    for _ in range(args.steps):
        result = step()
        tt, ta, nl = ctx.last_timing()
        acc_ms += ta; tot_ms += tt; launches += nl
        ck = ctx.last_accum_clock_mhz()
        if ck > 0:
            clock_sum += ck; clock_n += 1
    clock_mhz = clock_sum / clock_n if clock_n else 0.0
    merge_counts = ctx.last_merge_counts()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        batch = args.batch if (args.workload == "msm" and world == 1) else 1
        value = n * batch * args.steps / elapsed
        # dominant kernel: k_accum1 (one launch per window group; one group at these sizes)
        launches = max(launches, 1)
        accum_ms = acc_ms / launches
        # pairs one launch processes: all n pairs, for this rank's share of the windows
        # algorithmic bytes of one launch = 96 B x n x (the share of the windows that launch covers:
        # 1/world of them per rank, split over the launches of one step when the windows run in groups)
        launches_per_step = launches / args.steps
        achieved = BYTES_PER_PAIR * n / world / launches_per_step / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None if args.option else pmc_traffic(args.workload, curve, logn, world),   # (an --option run is not the plan the PMC passes measured)
                    "traffic_source": "committed rocprofv3 PMC passes of this command (profiles/traffic_accum1.json), not measured in this run",
                    "kernel": "k_accum1", "kernel_ms": round(accum_ms, 4), "launches_per_step": launches_per_step, "pipeline_device_ms": round(tot_ms / args.steps, 4),
                    "note": "integer-ALU-bound path: 96 algorithmic B/pair vs 8 TB/s HBM; see DESIGN.md for the VALU roofline"}
        # the bound that actually binds: VALU issue.  Peak = 1024 SIMDs x 2.4 GHz (the guide's maximum engine clock) / 4
        # cycles per wave instruction, so frac <= 1 by construction; the clock the kernel actually sustained is measured
        # live (in-kernel s_memtime / s_memrealtime stamps, lemsm_last_accum_clock_mhz) and reported beside it.  The
        # instruction count per mixed addition is model input from the committed PMC pass (profiles/valu_accum1.json).
        if args.workload == "msm":
            units, nonzero = ctx.msm_plan(curve, n)[0], 1.0
        else:
            units, nonzero = ctx.lhs_plan(curve, args.base)[0], (args.base - 1) / args.base
        vm = valu_model()
        madds = n * units * nonzero / world / launches_per_step
        wave_instr = madds * vm["instr_per_madd"] / 64
        valu_peak = 1024 * MAX_CLOCK_GHZ * 1e9 / VALU_ISSUE_CYCLES
        ach = wave_instr / (accum_ms * 1e-3) if accum_ms > 0 else 0.0
        clk = clock_mhz / 1e3
        roofline["valu_issue"] = {"instr_per_madd": vm["instr_per_madd"], "instr_per_madd_source": vm["source"],
                                  "madds_per_launch": int(madds),
                                  "achieved_Gwaveinstr_s": round(ach / 1e9, 1), "peak_Gwaveinstr_s": round(valu_peak / 1e9, 1),
                                  "frac": round(ach / valu_peak, 4), "peak_clock_ghz": MAX_CLOCK_GHZ,
                                  "clock_ghz_measured": round(clk, 3) if clk > 0 else None,
                                  "cycles_per_wave_instr_per_simd_at_measured_clock": round(1024 * clk * 1e9 / ach, 3) if (clk > 0 and ach > 0) else None,
                                  "model": "model-derived: PMC instruction count x launches / HIP-event time"}
        # ---- the timed result itself (last timed step) against the closed form of the synthetic input ----
        checks = verify_timed_result(cid, curve, scalars, q, result, args)
        out = {
            "metric": "BN254 G1 MSM scalar-point-pairs/s" if curve == "bn254_g1" else "Grumpkin MSM scalar-point-pairs/s",
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 limbs (256-bit Montgomery integers)", "data": "synthetic",
            "config": {"workload": ("%s MSM, 2^%d points, full-width scalars" % (curve, logn)) if args.workload == "msm"
                       else ("%s compute_lhs_witness MSM core, 2^%d points, negabase B=%d (w=4), half-width scalars" % (curve, logn, args.base)),
                       "n": n, "curve": curve, "sharding": ("pairs x%d" if (args.sharding == "points" and args.workload == "msm") else "pippenger-window x%d") % world,
                       "baseline_config": baseline_config(args.workload, curve, logn, world, args.base),
                       "exchange": {"none": "single GPU", "rccl-abi": "C ABI: lemsm_comm_init + ncclAllGather of raw device records (librccl by dlopen)",
                                    "torch": "torch.distributed all_gather_into_tensor of host-reduced window sums"}[exchange] if args.sharding == "windows" else "torch.distributed all-gather of one Jacobian partial per rank",
                       "input_gen_s": round(t_in, 2)},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, cid, curve, scalars, d_points, n, logn, args, world)
            checks.append("sample vs oracle")
        # bit_exact is derived from checks that ran in THIS process (each raises on a mismatch); null when none did
        if args.option:
            out["config"]["options"] = args.option
        if batch > 1:
            out["config"]["workload"] += ", BATCH of %d calls over the same resident points per step (lemsm_msm_batch_device: two lanes, call k's tail and host fold under call k+1)" % batch
            out["config"]["batch"] = batch
            out["config"]["baseline_config"] = "none (batched variant of configs[1] / configs[2]; the headline line is the single call)"
            out["roofline"]["note"] += "; batch run: kernel_ms / launches are those of the first lane's calls only"
        out["config"]["merge_queues"] = {"short_3to8": merge_counts[0], "medium_9to32": merge_counts[1], "long_slices": merge_counts[2], "multi_slice_buckets": merge_counts[3]}
        out["config"]["bit_exact"] = True if checks else None
        out["config"]["bit_exact_checks"] = checks
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


BUTTERFLY_PEAK = 112.7e9   # MEASURED register-only rate of the strict-field butterfly on the whole chip, 3 waves per SIMD (tools/ubench/butterfly_rates.hip,
                           # profiles/r03/d_butterfly_rates_strict_vs_lazy.txt); the r02 model (~1200 issue cycles per wave-butterfly at 2.1 GHz) said 115e9


def bench_lhs_witness(args, ctx, n, logn, world, rank, dist, torch, red_dev):
    """compute_lhs_witness in full (src/argument_witness_calc.rs:87-136: the carry AND the d divisor witnesses), Grumpkin,
    scalars / affine points resident in HBM and the coefficients left in HBM (lemsm_lhs_witness_device).  One JSON line of
    the same shape as the MSM workloads; `roofline` prices the transform launches of the merge forest (device time from
    HIP events inside the library, algorithmic bytes = one read + one write of every 32-byte element per pass over HBM).
    N > 1: the d merge trees are independent, so rank r computes the functions of its share of the digit positions
    (dist.window_range) on replicated inputs -- no exchange; every rank runs the MSM core for the carries."""
    from halo2_liam_eagen_msm_amd import DeviceBuffer, num_digits
    from halo2_liam_eagen_msm_amd import dist as ldist
    cid, curve = 1, "grumpkin"
    order = ORDER[curve]; fp = ORDER["bn254_g1"]
    scalars = gen_scalars(n, math.isqrt(order), 0x5EED1000 + logn)
    d_scalars = ctx.to_device(scalars)
    q = np.zeros(8, np.uint64)
    gx, gy = 1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C
    q[:4] = np.frombuffer(((gx << 256) % fp).to_bytes(32, "little"), np.uint64)
    q[4:] = np.frombuffer(((gy << 256) % fp).to_bytes(32, "little"), np.uint64)
    d_points = ctx.gen_walk(cid, q, n)
    for kv in args.option:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))
    d = num_digits(cid, args.base)
    out = DeviceBuffer(ctx, 2 * d * (n + args.base + 3) * 32)
    f_range = ldist.window_range(d, world, rank) if world > 1 else None

    def step():
        return ctx.lhs_witness_device(cid, d_scalars.ptr, d_points.ptr, n, args.base, True, out, f_range)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ntt_ms = 0.0; ntt_bytes = 0; ntt_bf = 0; phases = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        carry, index, _ = step()
        ms, by, bf = ctx.divisor_last_ntt()
        ntt_ms += ms; ntt_bytes += by; ntt_bf += bf
        phases += np.array(ctx.lhs_witness_last_phases())
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    achieved = ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
    coeffs = int(index[:, 1].sum() + index[:, 3].sum())
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": witness_traffic(logn, world, args), "traffic_source": "committed rocprofv3 PMC passes (profiles/traffic_witness.json; only the sizes measured there, else null), not measured in this run",
                "kernel": "k_ntt_tile (forward and inverse transforms of the merge forest, all levels of one call)",
                "reuse_levels": ctx.divisor_last_reuse_levels(),
                "kernel_ms": round(ntt_ms / args.steps, 3), "algorithmic_bytes_per_step": ntt_bytes // args.steps,
                "note": "LDS-tiled transforms are bound by the field multiplication, not by HBM (DESIGN.md section 6): butterflies/s against the VALU ceiling below",
                "valu_butterflies": {"achieved_G_s": round(ntt_bf / (ntt_ms * 1e-3) / 1e9, 1) if ntt_ms > 0 else 0.0, "peak_G_s": BUTTERFLY_PEAK / 1e9,
                                     "frac": round(ntt_bf / (ntt_ms * 1e-3) / BUTTERFLY_PEAK, 4) if ntt_ms > 0 else 0.0,
                                     "model": "measured peak: register-only butterfly loop of the same field on the whole chip (profiles/r03/d_butterfly_rates_strict_vs_lazy.txt)"},
                "phases_ms": {k: round(float(v) / args.steps, 2) for k, v in zip(("msm_core", "point_lists", "merge_forest", "coefficient_copy"), phases)}}
    checks = verify_lhs_witness(ctx, cid, scalars, q, d_points, n, args.base, carry, index, out, f_range)
    res = {"metric": "Grumpkin compute_lhs_witness scalar-point-pairs/s (carry + %d divisor witnesses)" % d, "value": round(n * args.steps / elapsed, 1), "unit": "pairs/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
           "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery integers)", "data": "synthetic",
           "config": {"workload": "grumpkin compute_lhs_witness in full, 2^%d points, negabase B=%d, half-width scalars" % (logn, args.base), "n": n, "curve": curve,
                      "coefficients_per_step": coeffs, "baseline_config": "SURVEY section 8 row f2 (the second return value of compute_lhs_witness); no BASELINE.json config names it",
                      "io": "scalars and affine points resident in HBM, coefficients left in HBM (lemsm_lhs_witness_device)",
                      "sharding": "single GPU" if world == 1 else "digit positions x%d (independent merge trees, no exchange; rank 0's figures in roofline)" % world},
           "roofline": roofline}
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_lhs_witness(ctx, cid, scalars, d_points, args)
        checks.append("sample vs oracle (every coefficient of every function)")
    if args.option:
        res["config"]["options"] = args.option
    res["config"]["bit_exact"] = True if checks else None
    res["config"]["bit_exact_checks"] = checks
    print(json.dumps(res), flush=True)


def verify_lhs_witness(ctx, cid, scalars, q, d_points, n, base, carry, index, out, f_range=None):
    """Checker leg for the timed compute_lhs_witness result: the carry against the closed form of the synthetic input, the
    shape of every function, and the property the reference's own test asserts (randpoints_witness_test :661): a function
    vanishes on the points of its list -- checked for one digit position on a multiple d_j P_j picked from the digits the
    oracle computes for a few scalars, by Horner evaluation in Python integers."""
    from oracle import cref, pyref
    exp = cref.jac_to_canonical(cid, cref.scalar_mul(cid, cref.walk_dot(cid, scalars), q))
    if cref.jac_to_canonical(cid, np.ascontiguousarray(carry, np.uint64)) != exp:
        raise SystemExit("the timed compute_lhs_witness carry differs from the closed form of the synthetic input: parity broken")
    checks = ["timed carry vs walk identity"]
    g = pyref.GRUMPKIN; p = g.fp
    d = index.shape[0]
    mine = range(d) if f_range is None else range(f_range[0], f_range[1])
    for f in mine:
        oa, la, ob, lb = (int(v) for v in index[f])
        if not (la + lb >= 1 and la + lb <= 2 * (n + base + 3)):
            raise SystemExit("function %d has an impossible shape (%d, %d)" % (f, la, lb))
    rinv = pow(1 << 256, -1, p)
    qa = g.raw_to_affine(q.tobytes())
    digs = {j: pyref.negbase_digits_padded(int.from_bytes(scalars[j].tobytes(), "little"), base, d) for j in (0, 1, n // 2, n - 1)}   # LSB first
    done = 0
    for pos in mine:
        js = [j for j, dg in digs.items() if dg[pos]]
        if not js or done >= 2:
            continue
        f = pos                                        # function f belongs to digit iteration i = d - 1 - f, which handles position d - 1 - i = f (LSB-first index)
        oa, la, ob, lb = (int(v) for v in index[f])
        a = _download_elems(ctx, out, oa, la); b = _download_elems(ctx, out, ob, lb)
        for j in js[:2]:
            pt = g.mul(digs[j][pos] * (j + 1) % g.order, qa)
            x, y = pt
            va = 0
            for k in range(la - 1, -1, -1):
                va = (va * x + int.from_bytes(a[k].tobytes(), "little")) % p
            vb = 0
            for k in range(lb - 1, -1, -1):
                vb = (vb * x + int.from_bytes(b[k].tobytes(), "little")) % p
            if (va + y * vb) * rinv % p != 0:
                raise SystemExit("function %d does not vanish on %d * P_%d: parity broken" % (f, digs[j][pos], j))
        done += 1
    if done:
        checks.append("witness vanishes on sampled points of its list (%d digit positions)" % done)
    return checks


def _download_elems(ctx, buf, off_elems, count):
    """count 32-byte elements of a DeviceBuffer starting at element off_elems, as (count, 4) u64"""
    import ctypes
    outa = np.empty((count, 4), np.uint64)
    if count:
        ctx._check(ctx.lib.lemsm_device_download(ctx.h, outa.ctypes.data_as(ctypes.c_void_p), buf.ptr + off_elems * 32, count * 32))
    return outa


def cpu_baseline_lhs_witness(ctx, cid, scalars, d_points, args):
    """Oracle leg: compute_lhs_witness of a bounded sample through the compiled, threaded restatement of the reference
    (oracle/c/witness_oracle.inc: schoolbook products below 32 coefficients, radix-2 FFT with the pinned omega above, the d
    merge trees spread over the host threads this process may use), timed; the GPU result on the same sample compared
    function by function, coefficient by coefficient (after normalising the coefficient of highest pole order, a
    RegularFunction being defined up to a scalar).  tests/test_oracle_golden.py pins that restatement to oracle/divisor.py."""
    import json as _json
    from oracle import pyref, divisor as dv, cref
    g = pyref.GRUMPKIN; p = g.fp
    slog = args.cpu_sample_log if args.cpu_sample_log is not None else 14      # ~10 s on 16 threads
    slog = min(slog, int(math.log2(scalars.shape[0])))
    m = 1 << slog
    head_bytes = bytes.fromhex(_json.load(open(os.path.join(ROOT, "tests", "golden", "fr_mont_chains.json")))["omega_pow"]["head"])
    omega_raw = np.frombuffer(head_bytes, np.uint64).copy()
    O = dv.DivisorOracle(g, None)                   # (normalise / to_affine only)
    rows = d_points.download(np.uint64, m * 64).reshape(-1, 8)
    pts_jac = cref.aff_to_jac(cid, rows)
    sc = np.ascontiguousarray(scalars[:m])
    threads = host_threads()
    t0 = time.perf_counter()
    st, ecarry, efns = cref.lhs_witness(sc, pts_jac, args.base, omega_raw, threads, decode=False)
    dt = time.perf_counter() - t0
    if st != 0:
        raise SystemExit("CPU restatement of compute_lhs_witness failed on the sample (status %d)" % st)
    st, ecarry, efns = cref.lhs_witness(sc, pts_jac, args.base, omega_raw, threads)      # again, decoded (untimed)
    ds = ctx.to_device(sc)
    carry, index, out = ctx.lhs_witness_device(cid, ds.ptr, d_points.ptr, m, args.base, True)
    rinv = pow(1 << 256, -1, p)
    if cref.jac_to_canonical(cid, np.ascontiguousarray(carry, np.uint64)) != cref.jac_to_canonical(cid, ecarry):
        raise SystemExit("GPU carry differs from the CPU oracle on the sample: parity broken")
    for f, exp in enumerate(efns):
        oa, la, ob, lb = (int(v) for v in index[f])
        e = O.normalise(exp)
        ga = [int.from_bytes(r.tobytes(), "little") * rinv % p for r in _download_elems(ctx, out, oa, la)]
        gb = [int.from_bytes(r.tobytes(), "little") * rinv % p for r in _download_elems(ctx, out, ob, lb)]
        if (ga, gb) != e:
            raise SystemExit("GPU divisor witness %d differs from the CPU oracle on the sample: parity broken" % f)
    return {"value": round(m / dt, 1), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "first 2^%d pairs of the same inputs, compute_lhs_witness restatement in C (oracle/c/witness_oracle.inc: the reference's products -- schoolbook below 32 coefficients, radix-2 FFT above -- with the %d merge trees spread over %d threads), %.2f s, GPU result on the sample: carry and all %d functions equal" % (slog, len(efns), threads, dt, len(efns))}


def witness_traffic(logn, world, args):
    """HBM bytes of the transform launches of one compute_lhs_witness call from the committed PMC passes, or None"""
    if world != 1 or args.option or args.base != 16:
        return None
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic_witness.json")))["lhs_witness/grumpkin/2^%d/x1/base16" % logn]["bytes_per_call"]
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(workload, curve, logn, world):
    """HBM bytes per k_accum1 launch from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command, tools/pmc_refresh.sh and
    tools/pmc_traffic.py; the gfx950 FETCH_SIZE doubling of MI355X_MICROARCH.md is NOT applied: it holds for wide
    coalesced streams, this kernel gathers 64-byte rows -- calibration in profiles/r01/fetch_size_calibration.txt).
    N > 1 entries were measured on ONE GPU running the ranks' pipelines one after the other (each entry says so).
    PMC cannot be collected from inside the process, so the figure is looked up for the exact workload and is
    null otherwise."""
    path = os.path.join(ROOT, "profiles", "traffic_accum1.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None
    key = "%s/%s/2^%d/x%d" % (workload, curve, logn, world)
    ent = table.get(key)
    return None if ent is None else ent["bytes_per_launch"]


def host_threads():
    """CPU threads this process may actually use: the smaller of the visible CPUs, the affinity mask and the cgroup quota
    (a one-GPU share of the box is 16 of its 256 hardware threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return max(1, n)


def init_exchange(ctx, want, world, rank, dist, torch, coll_dev):
    """N > 1: bring up the C ABI's RCCL communicator (unique id from rank 0, sent over the launcher's process group).
    Every rank ends up with the same answer: "rccl-abi" if all of them joined, else "torch" (printed to stderr)."""
    if want != "rccl-abi":
        return "torch"
    from halo2_liam_eagen_msm_amd import comm_unique_id
    dev = coll_dev if coll_dev is not None else "cpu"
    payload = np.zeros(129, np.uint8)
    if rank == 0:
        try:
            payload[:128] = np.frombuffer(comm_unique_id(), np.uint8)
            payload[128] = 1
        except Exception as e:      # librccl not loadable through the library: fall back, loudly
            print("bench: lemsm_comm_unique_id failed (%s); falling back to the torch.distributed exchange" % e, file=sys.stderr, flush=True)
    t = torch.from_numpy(payload).to(dev)
    dist.broadcast(t, 0)
    payload = t.cpu().numpy()
    ok = int(payload[128])
    if ok:
        try:
            ctx.comm_init(payload[:128].tobytes(), world, rank)
        except Exception as e:
            print("bench: rank %d lemsm_comm_init failed (%s)" % (rank, e), file=sys.stderr, flush=True)
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32).to(dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return "rccl-abi"
    if ok:
        ctx.comm_destroy()
    return "torch"


def valu_model():
    """VALU wave-instructions per mixed addition of k_accum1, from the committed PMC pass (SQ_INSTS_VALU)."""
    path = os.path.join(ROOT, "profiles", "valu_accum1.json")
    try:
        ent = json.load(open(path))["msm/bn254_g1/2^24/x1"]
        return {"instr_per_madd": ent["instr_per_madd"], "source": "profiles/valu_accum1.json (rocprofv3 --pmc SQ_INSTS_VALU)"}
    except Exception:
        return {"instr_per_madd": 2083, "source": "constant (profiles/valu_accum1.json unreadable)"}


def baseline_config(workload, curve, logn, world, base):
    """which BASELINE.json config this run is, if any"""
    if workload == "lhs" and curve == "bn254_g1" and logn == 20 and base == 16 and world == 1:
        return "configs[1]"
    if workload == "msm" and curve == "bn254_g1" and logn == 24:
        return "configs[2]" if world == 1 else "configs[2] sharded x%d (north_star: 2^24 at 1/2/4/8 GPUs)" % world
    if workload == "msm" and curve == "bn254_g1" and logn == 26 and world == 8:
        return "configs[3]"
    if workload == "msm" and curve == "grumpkin" and logn == 22 and world == 1:
        return "configs[4]"
    return None


def verify_timed_result(cid, curve, scalars, q, result, args):
    """Checker leg (oracle as the checker only): the result of the LAST TIMED step must equal (sum_i s_i (i+1)) Q,
    the closed form of the synthetic input P_i = (i+1) Q -- one host dot product mod the group order and one
    scalar multiplication.  Raises on a mismatch; returns the list of checks that passed."""
    from oracle import cref
    exp = cref.jac_to_canonical(cid, cref.scalar_mul(cid, cref.walk_dot(cid, scalars), q))
    got = cref.jac_to_canonical(cid, np.ascontiguousarray(result, np.uint64))
    if got != exp:
        raise SystemExit("the timed %s result differs from the closed form of the synthetic input: parity broken" % args.workload)
    return ["timed result vs walk identity"]


def cpu_baseline(ctx, cid, curve, scalars, d_points, n, logn, args, world):
    """Oracle leg (checker + timed CPU baseline, rank 0 only): a bounded sample of the same inputs through the
    oracle's restatement of the reference's CPU path for THIS workload, and the GPU result on that sample
    compared bit-exactly (single-GPU entry, whatever N is).
      msm: halo2 best_multiexp (thread-chunked serial Pippenger), all host threads;
      lhs: compute_lhs_witness' MSM core, serial like the reference (src/argument_witness_calc.rs:99-127 has no
           parallel loop), one thread."""
    from oracle import cref
    if args.workload == "msm":
        cores = host_threads()
        slog = args.cpu_sample_log if args.cpu_sample_log is not None else min(logn, 21 if world == 1 else 19)
    else:
        cores = 1
        slog = args.cpu_sample_log if args.cpu_sample_log is not None else min(logn, 18 if world == 1 else 16)
    m = 1 << slog
    pts = d_points.download(np.uint64, m * 64).reshape(-1, 8)
    sc = np.ascontiguousarray(scalars[:m])
    ds = ctx.to_device(sc)
    if args.workload == "msm":
        t0 = time.perf_counter()
        ref = cref.best_multiexp(cid, sc, pts, cores)
        dt = time.perf_counter() - t0
        got = ctx.msm_device(cid, ds.ptr, d_points.ptr, m)
        ok = cref.jac_to_canonical(cid, got) == cref.jac_to_canonical(cid, ref)
        what = "best_multiexp restatement (oracle/c)"
    else:
        jac = cref.aff_to_jac(cid, pts)
        t0 = time.perf_counter()
        ref, refs = cref.lhs_msm(cid, sc, jac, args.base, True)
        dt = time.perf_counter() - t0
        got, gots = ctx.lhs_msm_device(cid, ds.ptr, d_points.ptr, m, args.base, True)
        ok = cref.jac_to_canonical(cid, got) == cref.jac_to_canonical(cid, ref) and all(
            cref.jac_to_canonical(cid, gots[i]) == cref.jac_to_canonical(cid, refs[i]) for i in range(refs.shape[0]))
        what = "compute_lhs_witness MSM core restatement (oracle/c, serial like the reference), all %d per-digit carries compared" % refs.shape[0]
    if not ok:
        raise SystemExit("GPU result differs from the CPU oracle on the sample: parity broken")
    return {"value": round(m / dt, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "first 2^%d pairs of the same inputs, %s, %.2f s, GPU result on the sample bit-exact" % (slog, what, dt)}


if __name__ == "__main__":
    main()
