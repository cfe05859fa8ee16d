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


def _has_partner(box, cls, other_box, other_cls, px):
    """For every (box, cls): is there a same-class box in the other set with every coordinate within px?"""
    box, other_box = np.asarray(box, np.float64).reshape(-1, 4), np.asarray(other_box, np.float64).reshape(-1, 4)
    cls, other_cls = np.asarray(cls), np.asarray(other_cls)
    if len(box) == 0:
        return np.zeros(0, bool)
    if len(other_box) == 0:
        return np.zeros(len(box), bool)
    d = np.abs(box[:, None, :] - other_box[None, :, :]).max(axis=2)
    d[cls[:, None] != other_cls[None, :]] = np.inf
    return d.min(axis=1) <= px


def selection_sets(got_box, got_cls, runs, px=1.0):
    """The set-valued parity statement for an ill-conditioned selection (greedy NMS): `runs` = [(boxes, classes)] are the
    oracle's own detections on its logits, unperturbed (first) and perturbed by noise of the device's error size.
    core  = boxes of run 0 that every run keeps (their fate does not depend on the rounding noise),
    union = boxes that some run keeps.
    -> (core boxes the device lacks, device boxes outside the union, |core|, |union| summed over the runs)."""
    b0, k0 = runs[0]
    in_all = np.ones(len(b0), bool)
    for b, k in runs[1:]:
        in_all &= _has_partner(b0, k0, b, k, px)
    core_b, core_k = np.asarray(b0).reshape(-1, 4)[in_all], np.asarray(k0)[in_all]
    ub = np.concatenate([np.asarray(b, np.float64).reshape(-1, 4) for b, _ in runs])
    uk = np.concatenate([np.asarray(k) for _, k in runs])
    core_missing = int((~_has_partner(core_b, core_k, got_box, got_cls, px)).sum())
    outside = int((~_has_partner(got_box, got_cls, ub, uk, px)).sum())
    return core_missing, outside, int(in_all.sum()), len(ub)


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
