"""Shared helpers for the parity tests (test infrastructure; may use oracle/)."""
import hashlib
import json
import os

import numpy as np

from oracle import cref, pyref

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
CURVES = [pyref.BN254_G1, pyref.GRUMPKIN]


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def limbs4(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), np.uint64).copy()


def int_of(limbs) -> int:
    return int.from_bytes(np.ascontiguousarray(limbs, np.uint64).tobytes(), "little")


def regenerate_chain_digests(montmul, chains):
    """montmul(a_bytes32, b_bytes32) -> bytes32 in raw Montgomery form for modulus r.
    Rebuilds the three 64-entry tables of the reference's src/precomputed_fft_data.rs from their
    heads and returns {name: sha256}."""
    out = {}
    R = ((1 << 256) % pyref.R_BN254).to_bytes(32, "little")
    for name in ("omega_pow", "omega_pow_inv"):
        x = bytes.fromhex(chains[name]["head"])
        tab = [x]
        for _ in range(63):
            x = montmul(x, x)
            tab.append(x)
        out[name] = hashlib.sha256(b"".join(tab)).hexdigest()
    half = bytes.fromhex(chains["half_pow"]["head"])
    tab = [R, half]
    x = half
    for _ in range(62):
        x = montmul(x, half)
        tab.append(x)
    out["half_pow"] = hashlib.sha256(b"".join(tab)).hexdigest()
    return out


def golden_points_raw(curve, hex_list) -> np.ndarray:
    """canonical affine hex -> (n,8) raw Montgomery limbs"""
    out = np.zeros((len(hex_list), 8), np.uint64)
    for i, h in enumerate(hex_list):
        b = bytes.fromhex(h)
        x = int.from_bytes(b[:32], "little"); y = int.from_bytes(b[32:], "little")
        pt = None if (x == 0 and y == 0) else (x, y)
        out[i] = np.frombuffer(curve.affine_to_raw(pt), np.uint64)
    return out


def golden_scalars(hex_list) -> np.ndarray:
    return np.frombuffer(b"".join(bytes.fromhex(h) for h in hex_list), np.uint8).reshape(-1, 32).copy()


def jacobian_with_random_z(curve, pts_aff_raw: np.ndarray, seed: int) -> np.ndarray:
    """(n,8) affine raw -> (n,12) Jacobian raw with arbitrary non-zero Z (as hash_to_curve output has)."""
    rng = pyref.SplitMix64(seed)
    out = np.zeros((pts_aff_raw.shape[0], 12), np.uint64)
    for i in range(pts_aff_raw.shape[0]):
        pt = curve.raw_to_affine(pts_aff_raw[i].tobytes())
        z = 1 + rng.next256() % (curve.fp - 1)
        out[i] = np.frombuffer(curve.affine_to_jacobian_raw(pt, z), np.uint64)
    return out


def canon(curve, jac) -> bytes:
    return cref.jac_to_canonical(curve.cid, np.ascontiguousarray(jac, np.uint64))
