"""The drop-in boundary exercised the way a foreign-language host would: a plain C99 program (no Python, no torch) creates
a context, runs lemsm_msm / lemsm_lhs_msm on the committed golden vectors and compares the canonical bytes; and the
host-side challenge helpers of SURVEY 8(f).4 re-run under the gpu marker so that the driver's GPU record covers them."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import golden_points_raw, golden_scalars, jacobian_with_random_z
from halo2_liam_eagen_msm_amd import _lib
from oracle import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

C_SRC = r'''
#include "lemsm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* file: u32 kind (0 msm, 1 lhs), u32 curve, u32 n, u32 base, then n x 32 B scalars, n x (kind ? 96 : 64) B points, 64 B expected */
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  unsigned hdr[4]; if (fread(hdr, 4, 4, f) != 4) return 4;
  const unsigned kind = hdr[0], curve = hdr[1], n = hdr[2], base = hdr[3];
  const size_t pb = kind ? 96 : 64;
  uint8_t* sc = malloc((size_t)n * 32 + 1); uint64_t* pts = malloc((size_t)n * pb + 8); uint8_t want[64], got[64];
  if (fread(sc, 32, n, f) != n || fread(pts, pb, n, f) != n || fread(want, 1, 64, f) != 64) return 5;
  fclose(f);
  lemsm_ctx* ctx = NULL;
  int rc = lemsm_create(0, &ctx); if (rc != LEMSM_OK) { fprintf(stderr, "create: %s\n", lemsm_strerror(rc)); return 6; }
  uint64_t out[12];
  if (kind == 0) rc = lemsm_msm(ctx, (int)curve, sc, pts, n, out);
  else rc = lemsm_lhs_msm(ctx, (int)curve, sc, pts, n, (uint8_t)base, out, NULL, NULL);
  if (rc != LEMSM_OK) { fprintf(stderr, "call: %s (%s)\n", lemsm_strerror(rc), lemsm_last_error(ctx)); return 7; }
  if (lemsm_jacobian_to_canonical((int)curve, out, got) != LEMSM_OK) return 8;
  lemsm_destroy(ctx);
  if (memcmp(got, want, 64) != 0) { fprintf(stderr, "mismatch\n"); return 9; }
  printf("OK %u %u %u\n", kind, curve, n);
  free(sc); free(pts);
  return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "c_host.c"
    src.write_text(C_SRC)
    exe = tmp_path / "c_host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-llemsm", "-Wl,-rpath," + libdir])
    return exe


def test_c_program_runs_the_golden_vectors(tmp_path):
    exe = _build(tmp_path)
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "msm_vectors.json")))
    ran = 0
    for v in vec["msm"]:
        c = pyref.CURVES[v["curve"]]
        sc = golden_scalars(v["scalars"]); pts = golden_points_raw(c, v["points"])
        f = tmp_path / ("msm_%d.bin" % ran)
        with open(f, "wb") as fh:
            fh.write(np.array([0, c.cid, sc.shape[0], 0], np.uint32).tobytes()); fh.write(np.ascontiguousarray(sc).tobytes())
            fh.write(np.ascontiguousarray(pts, np.uint64).tobytes()); fh.write(bytes.fromhex(v["expected"]))
        out = subprocess.check_output([str(exe), str(f)], timeout=120).decode()
        assert out.startswith("OK 0"), out
        ran += 1
    for v in vec["lhs"]:
        c = pyref.CURVES[v["curve"]]
        sc = golden_scalars(v["scalars"]); pts = jacobian_with_random_z(c, golden_points_raw(c, v["points"]), v["seed"])
        f = tmp_path / ("lhs_%d.bin" % ran)
        with open(f, "wb") as fh:
            fh.write(np.array([1, c.cid, sc.shape[0], v["base"]], np.uint32).tobytes()); fh.write(np.ascontiguousarray(sc).tobytes())
            fh.write(np.ascontiguousarray(pts, np.uint64).tobytes()); fh.write(bytes.fromhex(v["expected_carry"]))
        out = subprocess.check_output([str(exe), str(f)], timeout=120).decode()
        assert out.startswith("OK 1"), out
        ran += 1
    assert ran >= 4


@pytest.mark.parametrize("curve", pyref.CURVES.values(), ids=lambda c: c.name)
def test_challenge_helpers_under_the_gpu_marker(curve):
    """to_curve_x / y_from_x / slope (src/config.rs:166-187) are host functions of the library; this is
    tests/test_host_logic.py's check run again where the driver records it"""
    from test_host_logic import test_to_curve_x_y_from_x_slope
    test_to_curve_x_y_from_x_slope(curve)
