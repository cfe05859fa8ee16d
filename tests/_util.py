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


def match_detections(got_box, got_cls, want_box, want_cls, px=1.0):
    """One-to-one matching of two detection sets: a pair matches when the classes agree and every box coordinate
    differs by at most `px`.  Greedy over the wanted boxes, best (smallest max |d|) partner first.
    -> (pairs [(i_want, j_got)], unmatched_want indices, unmatched_got indices, worst |d| over the pairs)."""
    got_box, want_box = np.asarray(got_box, np.float64).reshape(-1, 4), np.asarray(want_box, np.float64).reshape(-1, 4)
    used = np.zeros(len(got_box), bool)
    pairs, miss, worst = [], [], 0.0
    for i in range(len(want_box)):
        if len(got_box) == 0:
            miss.append(i)
            continue
        d = np.abs(got_box - want_box[i]).max(axis=1)
        d[used | (np.asarray(got_cls) != want_cls[i])] = np.inf
        j = int(np.argmin(d))
        if d[j] <= px:
            used[j] = True
            pairs.append((i, j))
            worst = max(worst, float(d[j]))
        else:
            miss.append(i)
    return pairs, miss, [int(j) for j in np.nonzero(~used)[0]], worst


def spread_params(seed=0):
    """The seeded random YOLOv8n-topology parameter vector with the Detect head's final class convolutions rescaled so
    that confidences spread over (0, 1) instead of sitting within 1e-3 of each other (which is what plain random
    initialisation gives: every anchor passes the 0.25 filter and the NMS order is decided by rounding noise).
    Test input only: weights x 30, biases - 8 on the three 80 -> 80 1x1 convolutions (about 200 of the 5040 anchors
    then pass the filter, with a median confidence gap of 5e-4 between neighbours in the sorted list)."""
    from oracle import yolo_ref as R
    p = R.random_params(seed).copy()
    pos = 0
    for cin, cout, k, _, bn in R.conv_specs():
        nw = cout * cin * k * k
        if not bn and cout == R.NC:
            p[pos:pos + nw] *= 30.0
            p[pos + nw:pos + nw + cout] -= 8.0
        pos += nw + (4 * cout if bn else cout)
    assert pos == p.size
    return p
