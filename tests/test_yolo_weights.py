"""YOLO-mode weights: a YOLOv8n state_dict with ultralytics' key names -> the library's flat parameter vector
(reference call site: detector.py:77-84 `YOLO(model_path)`, demo.py:41 / app.py:44 pass "yolov8n.pt").  CPU only."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
nn = torch.nn


def _ultralytics_shaped_skeleton():
    """A module tree with ultralytics' attribute names for YOLOv8n (nn/modules: Conv = conv + bn, Bottleneck = cv1 + cv2,
    C2f = cv1, cv2, m[...], SPPF = cv1, cv2, Detect = cv2[level][0..2], cv3[level][0..2], dfl.conv), registered in that
    package's order, so that state_dict() yields its key list -- including the entries a loader has to skip."""
    class Conv(nn.Module):
        def __init__(self, c1, c2, k=1):
            super().__init__()
            self.conv = nn.Conv2d(c1, c2, k, bias=False)
            self.bn = nn.BatchNorm2d(c2, eps=1e-3)

    class Bottleneck(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.cv1, self.cv2 = Conv(c, c, 3), Conv(c, c, 3)

    class C2f(nn.Module):
        def __init__(self, c1, c2, n):
            super().__init__()
            c = c2 // 2
            self.cv1, self.cv2 = Conv(c1, 2 * c, 1), Conv((2 + n) * c, c2, 1)
            self.m = nn.ModuleList(Bottleneck(c) for _ in range(n))

    class SPPF(nn.Module):
        def __init__(self):
            super().__init__()
            self.cv1, self.cv2 = Conv(256, 128, 1), Conv(512, 256, 1)

    class DFL(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(16, 1, 1, bias=False)

    class Detect(nn.Module):
        def __init__(self):
            super().__init__()
            self.cv2 = nn.ModuleList(nn.Sequential(Conv(c, 64, 3), Conv(64, 64, 3), nn.Conv2d(64, 64, 1)) for c in (64, 128, 256))
            self.cv3 = nn.ModuleList(nn.Sequential(Conv(c, 80, 3), Conv(80, 80, 3), nn.Conv2d(80, 80, 1)) for c in (64, 128, 256))
            self.dfl = DFL()

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            up = nn.Upsample(scale_factor=2)
            self.model = nn.Sequential(
                Conv(3, 16, 3), Conv(16, 32, 3), C2f(32, 32, 1), Conv(32, 64, 3), C2f(64, 64, 2), Conv(64, 128, 3), C2f(128, 128, 2),
                Conv(128, 256, 3), C2f(256, 256, 1), SPPF(), up, nn.Identity(), C2f(384, 128, 1), up, nn.Identity(), C2f(192, 64, 1),
                Conv(64, 64, 3), nn.Identity(), C2f(192, 128, 1), Conv(128, 128, 3), nn.Identity(), C2f(384, 256, 1), Detect())
    return Net()


def _fill_from_oracle(net, params):
    """Copies the oracle model's tensors (oracle/yolo_ref.py: its own attribute names l0..l21, box/cls lists) into the
    skeleton, layer by layer, by this table -- independent of the loader's name list."""
    from oracle import yolo_ref as R
    o = R.build_model(params)
    m = net.model

    def conv(dst, src):
        with torch.no_grad():
            dst.conv.weight.copy_(src.w), dst.bn.weight.copy_(src.g), dst.bn.bias.copy_(src.b)
            dst.bn.running_mean.copy_(src.m), dst.bn.running_var.copy_(src.v)

    def c2f(dst, src):
        conv(dst.cv1, src.cv1), conv(dst.cv2, src.cv2)
        for b, (a1, a2) in zip(dst.m, src.m):
            conv(b.cv1, a1), conv(b.cv2, a2)
    for i, name in ((0, "l0"), (1, "l1"), (3, "l3"), (5, "l5"), (7, "l7"), (16, "l16"), (19, "l19")):
        conv(m[i], getattr(o, name))
    for i, name in ((2, "l2"), (4, "l4"), (6, "l6"), (8, "l8"), (12, "l12"), (15, "l15"), (18, "l18"), (21, "l21")):
        c2f(m[i], getattr(o, name))
    conv(m[9].cv1, o.s1), conv(m[9].cv2, o.s2)
    for lvl in range(3):
        for dst, src in ((m[22].cv2[lvl], o.box[lvl]), (m[22].cv3[lvl], o.cls[lvl])):
            conv(dst[0], src[0]), conv(dst[1], src[1])
            with torch.no_grad():
                dst[2].weight.copy_(src[2].w), dst[2].bias.copy_(src[2].bias)
    with torch.no_grad():
        m[22].dfl.conv.weight.copy_(torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1))


def test_state_dict_with_ultralytics_names_round_trips_to_the_parameter_vector(tmp_path):
    from multimodal_autonomous_driving_perception_and_planning_amd.perception import yolo as Y
    from oracle import yolo_ref as R
    params = R.random_params(7)
    net = _ultralytics_shaped_skeleton()
    _fill_from_oracle(net, params)
    sd = net.state_dict()
    keys = list(sd)
    assert keys[0] == "model.0.conv.weight" and "model.0.bn.num_batches_tracked" in sd and "model.22.dfl.conv.weight" in sd
    assert "model.4.m.1.cv2.bn.running_var" in sd and "model.22.cv3.2.2.bias" in sd and "model.9.cv2.conv.weight" in sd
    # every parameter the graph consumes, and nothing else but the skipped entries
    consumed = sum(v.numel() for k, v in sd.items() if "num_batches_tracked" not in k and ".dfl." not in k)
    assert consumed == params.size == R.param_count()
    pt = str(tmp_path / "yolov8n_sd.pt")
    torch.save(sd, pt)
    assert np.array_equal(Y.load_params(pt), params)
    # the same under the keys of the YOLO wrapper ("model.model.N..."), inside a checkpoint-style dict, as safetensors, as npz
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "epoch": 3}, str(tmp_path / "ckpt.pth"))
    assert np.array_equal(Y.load_params(str(tmp_path / "ckpt.pth")), params)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "w.safetensors"))
    assert np.array_equal(Y.load_params(str(tmp_path / "w.safetensors")), params)
    np.savez(str(tmp_path / "w.npz"), **{k: v.numpy() for k, v in sd.items()})
    assert np.array_equal(Y.load_params(str(tmp_path / "w.npz")), params)
    # and back: the vector re-exported under those names is the skeleton's state_dict
    back = Y.state_dict_from_params(params)
    assert set(back) == {k for k in sd if "num_batches_tracked" not in k and ".dfl." not in k}
    assert all(np.array_equal(back[k], sd[k].numpy()) for k in back)


def test_wrong_or_unreadable_checkpoints_fail_loudly(tmp_path):
    from multimodal_autonomous_driving_perception_and_planning_amd.perception import yolo as Y
    net = _ultralytics_shaped_skeleton()
    sd = net.state_dict()
    bad = dict(sd)
    del bad["model.6.m.1.cv1.bn.running_mean"]
    torch.save(bad, str(tmp_path / "missing.pt"))
    with pytest.raises(KeyError, match="model.6.m.1.cv1.bn.running_mean"):
        Y.load_params(str(tmp_path / "missing.pt"))
    wide = dict(sd)
    wide["model.1.conv.weight"] = torch.zeros(64, 32, 3, 3)           # a YOLOv8s-sized layer
    torch.save(wide, str(tmp_path / "wide.pt"))
    with pytest.raises(ValueError, match="1.conv.weight"):
        Y.load_params(str(tmp_path / "wide.pt"))
    # a file that pickles module objects (what an ultralytics training checkpoint is) is refused as "not found / not readable",
    # which ObjectDetector turns into the reference's fallback to simulated mode (detector.py:82-84)
    torch.save({"model": nn.Sequential(nn.Conv2d(3, 16, 3))}, str(tmp_path / "pickled.pt"))
    with pytest.raises(FileNotFoundError, match="state_dict"):
        Y.load_params(str(tmp_path / "pickled.pt"))
    with pytest.raises(FileNotFoundError):
        Y.load_params(str(tmp_path / "yolov8n.pt"))
