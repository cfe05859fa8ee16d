"""Second, independently coded restatements of the UNPINNED oracles (lane pixel path, Kalman filter).

OpenCV and filterpy are absent and the reference has no tests, so oracle/c/lane_ref.c, oracle/lane_ref.py and
oracle/kf_ref.py cannot be pinned to the libraries they restate.  What can be done is to catch transcription errors:
every stage is computed a second time by different means (torch float64 convolutions, np.median, connected-component
labelling instead of a flood-fill stack, a plain-Python progressive Hough transform, an information-form filter) and
must agree exactly (integers) / to rounding (floats).  Both sides still rest on the same reading of the libraries."""
import numpy as np
import pytest


def _frames():
    from oracle.lane_ref import synthetic_frame
    rng = np.random.RandomState(17)
    return [synthetic_frame(120, 160, 2, 5), synthetic_frame(96, 208, 0, 1), rng.randint(0, 256, size=(60, 88, 3)).astype(np.uint8)]


def _blur_torch(gray):
    import torch
    import torch.nn.functional as F
    k = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0], dtype=torch.float64)
    x = torch.from_numpy(gray.astype(np.float64))[None, None]
    x = F.pad(x, (2, 2, 2, 2), mode="reflect")                           # BORDER_REFLECT_101
    s = F.conv2d(x, torch.outer(k, k)[None, None])
    return torch.floor((s + 128.0) / 256.0)[0, 0].numpy().astype(np.uint8)


def _canny_labelled(blur, lo, hi):
    """Sobel by float64 convolution with replicate padding, NMS vectorised, hysteresis = 8-connected components of the
    candidates that hold at least one strong pixel (scipy.ndimage.label)."""
    import torch
    import torch.nn.functional as F
    from scipy import ndimage
    x = F.pad(torch.from_numpy(blur.astype(np.float64))[None, None], (1, 1, 1, 1), mode="replicate")
    kx = torch.tensor([[-1.0, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float64)
    gx = F.conv2d(x, kx[None, None])[0, 0].numpy().astype(np.int64)
    gy = F.conv2d(x, kx.t()[None, None])[0, 0].numpy().astype(np.int64)
    m = np.abs(gx) + np.abs(gy)
    mp = np.pad(m, 1)                                                    # magnitude 0 outside the image
    h, w = m.shape
    c = mp[1:-1, 1:-1]
    ax, ay = np.abs(gx), np.abs(gy) << 15
    t22 = ax * 13573
    t67 = t22 + (ax << 16)
    horiz, vert = ay < t22, ay > t67
    neg = (gx ^ gy) < 0

    def sh(dy, dx):
        return mp[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    is_h = (c > sh(0, -1)) & (c >= sh(0, 1))
    is_v = (c > sh(-1, 0)) & (c >= sh(1, 0))
    is_d_pos = (c > sh(-1, -1)) & (c > sh(1, 1))                         # gradient signs equal: neighbours (-1,-1), (+1,+1)
    is_d_neg = (c > sh(-1, 1)) & (c > sh(1, -1))
    is_max = np.where(horiz, is_h, np.where(vert, is_v, np.where(neg, is_d_neg, is_d_pos)))
    cand = is_max & (c > lo)
    strong = cand & (c > hi)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), int))
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


def _ppht_python(image, threshold=50, line_length=50, line_gap=150):
    """cv::HoughLinesProbabilistic step by step in plain Python (rho 1, theta 1 degree)."""
    h, w = image.shape
    numangle, numrho = 180, (w + h) * 2 + 1
    theta = np.float32(np.pi / 180.0)
    trig = [(np.float32(np.cos(float(n) * float(theta))), np.float32(np.sin(float(n) * float(theta)))) for n in range(numangle)]
    accum = np.zeros((numangle, numrho), np.int64)
    mask = image != 0
    mask = mask.copy()
    pts = [(x, y) for y in range(h) for x in range(w) if mask[y, x]]
    state = (1 << 64) - 1
    lines = []
    rnd = lambda v: int(np.rint(np.float32(v)))                           # noqa: E731  cvRound of a float
    count = len(pts)
    while count > 0:
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        idx = (state & 0xFFFFFFFF) % count
        j, i = pts[idx]
        pts[idx] = pts[count - 1]
        count -= 1
        if not mask[i, j]:
            continue
        max_val, max_n = threshold - 1, 0
        for n in range(numangle):
            r = rnd(np.float32(j) * trig[n][0] + np.float32(i) * trig[n][1]) + (numrho - 1) // 2
            accum[n, r] += 1
            if max_val < accum[n, r]:
                max_val, max_n = accum[n, r], n
        if max_val < threshold:
            continue
        a, b = -trig[max_n][1], trig[max_n][0]
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag, dx0 = True, (1 if a > 0 else -1)
            dy0 = rnd(b * np.float32(65536.0) / abs(a))
            y0 = (y0 << 16) + (1 << 15)
        else:
            xflag, dy0 = False, (1 if b > 0 else -1)
            dx0 = rnd(a * np.float32(65536.0) / abs(b))
            x0 = (x0 << 16) + (1 << 15)
        ends = []
        for k in range(2):
            gap, x, y = 0, x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            ex = ey = 0
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if j1 < 0 or j1 >= w or i1 < 0 or i1 >= h:
                    break
                if mask[i1, j1]:
                    gap, ex, ey = 0, j1, i1
                else:
                    gap += 1
                    if gap > line_gap:
                        break
                x, y = x + dx, y + dy
            ends.append((ex, ey))
        good = abs(ends[1][0] - ends[0][0]) >= line_length or abs(ends[1][1] - ends[0][1]) >= line_length
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if mask[i1, j1]:
                    if good:
                        for n in range(numangle):
                            r = rnd(np.float32(j1) * trig[n][0] + np.float32(i1) * trig[n][1]) + (numrho - 1) // 2
                            accum[n, r] -= 1
                    mask[i1, j1] = False
                if (i1, j1) == (ends[k][1], ends[k][0]):
                    break
                x, y = x + dx, y + dy
        if good:
            lines.append((ends[0][0], ends[0][1], ends[1][0], ends[1][1]))
    return np.array(lines, np.int32).reshape(-1, 4)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_lane_pixel_stages_against_second_restatement(k):
    from oracle import lane_ref as L
    frame = _frames()[k]
    st = L.LaneRef().stages(frame)
    f64 = frame.astype(np.int64)
    gray = ((1868 * f64[..., 0] + 9617 * f64[..., 1] + 4899 * f64[..., 2] + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(st["gray"], gray)
    assert np.array_equal(st["blur"], _blur_torch(gray))
    med = float(np.median(st["blur"]))
    assert (st["median"], st["lo"], st["hi"]) == (med, int(max(0, 0.7 * med)), int(min(255, 1.3 * med)))
    assert np.array_equal(st["edges"], _canny_labelled(st["blur"], st["lo"], st["hi"]))


def test_probabilistic_hough_against_plain_python():
    """Three small edge maps (a few hundred edge points each, with real lines in them) through the C oracle's PPHT and
    through the step-by-step Python one: same segments in the same order."""
    from oracle import lane_ref as L
    rng = np.random.RandomState(4)
    got_lines = 0
    for k in range(3):
        img = np.zeros((90, 140), np.uint8)
        for _ in range(3 + k):
            x0, y0, x1, y1 = rng.randint(5, 135), rng.randint(5, 85), rng.randint(5, 135), rng.randint(5, 85)
            n = max(abs(x1 - x0), abs(y1 - y0)) + 1
            xs, ys = np.linspace(x0, x1, n).round().astype(int), np.linspace(y0, y1, n).round().astype(int)
            keep = rng.rand(n) < 0.9
            img[ys[keep], xs[keep]] = 255
        img[rng.randint(0, 90, 60), rng.randint(0, 140, 60)] = 255            # clutter
        for thr, ll, gap in ((25, 30, 10), (30, 20, 4)):
            want = _ppht_python(img, thr, ll, gap)
            have = L.houghp(img, thr, ll, gap)
            assert np.array_equal(have, want), (k, thr, have, want)
            got_lines += len(want)
    assert got_lines >= 6


def test_kalman_oracle_against_information_filter():
    """kf_ref restates filterpy's covariance-form predict / Joseph-form update; the same model as an information filter
    (Y = P^-1, y = Y x; measurement update adds H^T R^-1 H and H^T R^-1 z) must give the same states and covariances."""
    from oracle.harness_ref import ego_motion
    from oracle.kf_ref import KalmanRef
    dt = 0.033
    F = np.eye(6)
    F[0, 2] = F[1, 3] = F[2, 4] = F[3, 5] = dt
    F[0, 4] = F[1, 5] = 0.5 * dt * dt
    H = np.zeros((4, 6))
    H[:4, :4] = np.eye(4)
    Q = np.diag([0.1, 0.1, 0.1, 0.1, 1.0, 1.0])
    Rinv = np.eye(4) / 1.0
    x, P = np.zeros(6), np.eye(6) * 10.0
    ref = KalmanRef()
    z = ego_motion(120, seed=3)
    for t in range(120):
        x, P = F @ x, F @ P @ F.T + Q                                    # predict in covariance form
        Y = np.linalg.inv(P)
        y = Y @ x
        Y2, y2 = Y + H.T @ Rinv @ H, y + H.T @ Rinv @ z[t]
        P = np.linalg.inv(Y2)
        x = P @ y2
        st = ref.step(z[t])
        np.testing.assert_allclose(ref.x, x, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ref.P, P, rtol=1e-8, atol=1e-10)
        assert abs(st[5] - np.hypot(x[2], x[3])) < 1e-9                   # derived speed
