#!/usr/bin/env python3
"""Generates the committed golden fixtures in this directory.

Two kinds of data:

1. fr_mont_chains.json -- derived from the reference's only byte-level golden data,
   the 192 raw-Montgomery bn256::Fr constants of src/precomputed_fft_data.rs.
   The file itself is NOT copied: we keep the three chain heads (omega_pow[0],
   omega_pow_inv[0], half_pow[1]) and a SHA-256 digest of each 64 x 32-byte table.  A
   Montgomery multiplier that regenerates the chains (x_{i+1} = montmul(x_i, x_i) resp.
   montmul(x_i, half)) and hits the digests reproduces all 192 constants.
   Needs /root/reference (present in the build container only).

2. msm_vectors.json -- small seeded input/output vectors produced by the oracle
   (oracle/pyref.py, the independent big-int implementation): the reference holds no MSM
   known-answer vectors (parity unpinned) and cannot be built here (Rust,
   unpinned git dependencies, no toolchain), so these pin the oracle against itself across
   rounds and give the GPU path fixed answers.
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref  # noqa: E402

REF = "/root/reference/src/precomputed_fft_data.rs"


def parse_reference_tables(path=REF):
    text = open(path).read()
    tables = {}
    for name in ("omega_pow", "omega_pow_inv", "half_pow"):
        m = re.search(r"fn %s\(.*?\{(.*?)_=>panic" % name, text, re.S)
        rows = re.findall(r"(\d+)=>\[([0-9,\s]+)\]", m.group(1))
        tab = [None] * 64
        for idx, body in rows:
            b = bytes(int(v) for v in body.split(","))
            assert len(b) == 32
            tab[int(idx)] = b
        assert all(t is not None for t in tab)
        tables[name] = tab
    return tables


def make_chains():
    t = parse_reference_tables()
    out = {"modulus": hex(pyref.R_BN254), "source": "src/precomputed_fft_data.rs:4-215 (digests, not a copy)"}
    for name, tab in t.items():
        out[name] = {
            "head_index": 1 if name == "half_pow" else 0,
            "head": tab[1 if name == "half_pow" else 0].hex(),
            "sha256": hashlib.sha256(b"".join(tab)).hexdigest(),
        }
    json.dump(out, open(os.path.join(HERE, "fr_mont_chains.json"), "w"), indent=1)


def make_vectors():
    vecs = {"format": "scalars: 32B LE hex; points: affine canonical x||y 64B LE hex (identity zeros); "
                      "expected: canonical affine x||y", "msm": [], "lhs": []}
    for curve in (pyref.BN254_G1, pyref.GRUMPKIN):
        for n, seed in ((1, 101), (2, 102), (17, 103), (64, 104)):
            rng = pyref.SplitMix64(seed)
            pts = pyref.gen_points(curve, rng, n)
            sc = pyref.gen_scalars_full(rng, n, curve.order)
            if n == 17:   # sprinkle edge cases: zero scalar, identity point, repeated and opposite points
                sc[0] = 0
                pts[1] = None
                pts[3] = pts[2]
                pts[5] = curve.neg(pts[4]); sc[5] = sc[4]
                sc[6] = curve.order - 1
            exp = curve.msm_naive(sc, pts)
            vecs["msm"].append({
                "curve": curve.name, "n": n, "seed": seed,
                "scalars": [int(s).to_bytes(32, "little").hex() for s in sc],
                "points": [curve.canonical(p).hex() for p in pts],
                "expected": curve.canonical(exp).hex(),
            })
        for n, seed, base in ((1, 201, 5), (9, 202, 5), (33, 203, 16), (12, 204, 3), (10, 205, 255)):
            rng = pyref.SplitMix64(seed)
            pts = pyref.gen_points(curve, rng, n)
            sc = pyref.gen_scalars_half(rng, n, curve.order)
            if n == 9:
                sc[0] = 0
                pts[2] = pts[1]; sc[2] = sc[1]          # the reference's own test shape: equal scalar, equal point
                sc[3] = pyref.scalar_bound(curve.order) - 1   # largest admissible scalar
            carry, carries = pyref.lhs_msm(curve, sc, pts, base)
            assert curve.canonical(carry) == curve.canonical(curve.msm_naive(sc, pts))
            vecs["lhs"].append({
                "curve": curve.name, "n": n, "seed": seed, "base": base,
                "scalars": [int(s).to_bytes(32, "little").hex() for s in sc],
                "points": [curve.canonical(p).hex() for p in pts],
                "digits_lsb_first": [pyref.negbase_digits_padded(s, base, pyref.num_digits(curve.order, base)) for s in sc],
                "expected_carry": curve.canonical(carry).hex(),
                "expected_carries_msb_first": [curve.canonical(c).hex() for c in carries],
            })
    json.dump(vecs, open(os.path.join(HERE, "msm_vectors.json"), "w"), indent=0)


if __name__ == "__main__":
    if os.path.exists(REF):
        make_chains()
    else:
        print("reference not present: fr_mont_chains.json left untouched")
    make_vectors()
    print("golden fixtures written")
