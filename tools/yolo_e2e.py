#!/usr/bin/env python3
"""End-to-end YOLO-mode parity report: ObjectDetector(mode="yolo").detect() on the device vs the fp32 PyTorch-CPU
restatement run end to end on ITS OWN logits (decode + NMS + scale_boxes + int()), per frame: matched boxes within
+-1 px, flips (kept on one side only), worst coordinate difference.  Tuning / reporting aid for tests/test_gpu_yolo.py.
(The fp32 restatement is passed in by the caller: this tool does not import the oracle package itself.)"""
import numpy as np


def oracle_detections(R, net, frame, torch):
    with torch.no_grad():
        f = net.features(torch.from_numpy(R.preprocess(frame))[None])
    xyxy, conf, cls = R.decode(f["head"])
    keep = R.nms(xyxy, conf, cls)
    h, w = frame.shape[:2]
    box = R.scale_boxes(xyxy[keep], h, w)
    return box, conf[keep], cls[keep], f


def report(frames, detect, R, net, torch, match, px=1.0):
    rows = []
    for k, fr in enumerate(frames):
        wb, wc, wk, _ = oracle_detections(R, net, fr, torch)
        gb, gc, gk = detect(fr)
        # the reference truncates with int() (detector.py:111): compare the truncated boxes, like its consumers see them
        pairs, miss, extra, worst = match(np.trunc(gb), gk, np.trunc(wb), wk, px)
        dconf = max([abs(float(gc[j]) - float(wc[i])) for i, j in pairs], default=0.0)
        rows.append(dict(frame=k, n_want=len(wb), n_got=len(gb), matched=len(pairs), missing=len(miss), extra=len(extra),
                         flip_rate=(len(miss) + len(extra)) / max(1, len(wb) + len(gb)), worst_px=worst, worst_dconf=dconf))
    return rows


def candidate_stats(model, frame, R, net, torch):
    """Every anchor's decoded box / confidence / class from the DEVICE's head logits vs the fp32 restatement's own
    (the same decode on both sides, so only the network's arithmetic differs).  Boxes in frame pixels."""
    with torch.no_grad():
        f = net.features(torch.from_numpy(R.preprocess(frame))[None])
    model.detect(frame)
    head = []
    for i in range(3):
        hb, hc = model.tensor(100 + 2 * i), model.tensor(101 + 2 * i)
        head.append((torch.from_numpy(hb.transpose(2, 0, 1).copy())[None], torch.from_numpy(hc.transpose(2, 0, 1).copy())[None]))
    gb, gc, gk = R.decode(head)
    wb, wc, wk = R.decode(f["head"])
    h, w = frame.shape[:2]
    d = np.abs(R.scale_boxes(gb, h, w) - R.scale_boxes(wb, h, w)).max(axis=1)
    return dict(box_max_px=float(d.max()), box_p99_px=float(np.percentile(d, 99)), box_mean_px=float(d.mean()),
                frac_over_1px=float((d > 1.0).mean()), conf_max=float(np.abs(gc - wc).max()),
                class_flips=int((gk != wk).sum()), conf_spread=float(wc.max() - wc.min()),
                n_over_thres=int((wc > 0.25).sum()))
