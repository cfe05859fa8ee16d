import numpy as np


def orders_equivalent(cost_ref, order_ref, order_got, rtol=1e-9):
    """Two stable ascending orders agree up to permutations inside groups of (near-)equal cost.

    Exact ties must keep generation order; near-ties (|dc| <= rtol*|c|) may swap because they are decided
    by the last bits of exp/sin/cos/atan2, which differ between libm/SVML and the device math library.
    """
    order_ref = np.asarray(order_ref)
    order_got = np.asarray(order_got)
    if np.array_equal(order_ref, order_got):
        return True
    if sorted(order_ref.tolist()) != sorted(order_got.tolist()):
        return False
    cr = np.asarray(cost_ref)[order_ref]
    cg = np.asarray(cost_ref)[order_got]
    return bool(np.allclose(cr, cg, rtol=rtol, atol=1e-12))
