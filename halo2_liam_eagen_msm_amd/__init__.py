"""MI355X (gfx950) MSM witness path for Liam Eagen's MSM argument.

`api` mirrors the reference's Rust entry points over the C ABI in include/lemsm.h;
`dist` shards one MSM by Pippenger window / negabase digit position over the GPUs of a node.
"""
from . import _lib  # noqa: F401
from .api import (  # noqa: F401
    BN254_G1, GRUMPKIN, Context, DeviceBuffer, LemsmError, LengthMismatch, ScalarOutOfRange, BadBase,
    best_multiexp, compute_lhs_witness, compute_lhs_witness_inputs, negbase_decompose, precompute_multiplicities,
    jacobian_to_canonical, jacobian_sum, logb_ceil, order, num_digits, id_by_digit, digit_by_id,
    Bases, Node, comm_unique_id, prepare_scalar_witness, table_entry_by_id, TooManyDigits, RefIndexOutOfBounds,
    RefArithmeticOverflow, ENTRY_DTYPE, SumNotIdentity, compute_divisor_witness, compute_divisor_witness_partial,
    to_curve_x, y_from_x, slope, WouldNotTerminate,
)
