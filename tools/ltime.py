import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
from multimodal_autonomous_driving_perception_and_planning_amd.harness import synthetic_frame
S, h, w, MS = 64, 720, 1280, 512
ctx = nat.Context(0); L = nat.lib(); dev = torch.device("cuda", 0)
base = [synthetic_frame(h, w, s, 0) for s in range(8)]
frames = torch.as_tensor(np.stack([base[s % 8] for s in range(S)])).to(dev)
ws = torch.empty(int(L.av_lane_workspace_bytes(S, h, w, MS)), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(); sh = C.c_void_p(st.cuda_stream)
nat.check(L.av_lane_workspace_init(ctx.handle, sh, S, h, w, MS, nat.ptr(ws)))
state = torch.zeros(S, 8, dtype=torch.float64, device=dev); poly = torch.zeros(S, 2, 3, dtype=torch.float64, device=dev)
pts = torch.zeros(S, 2, 50, 2, dtype=torch.int32, device=dev); info = torch.zeros(S, 8, dtype=torch.int32, device=dev)
conf = torch.zeros(S, 2, dtype=torch.float64, device=dev)
cfg = nat.LaneCfg(50, 50, 150, MS, 0.7)
for _ in range(3):
    nat.check(L.av_lane_detect(ctx.handle, sh, C.byref(cfg), S, h, w, nat.ptr(frames), None, nat.ptr(ws), nat.ptr(state), nat.ptr(poly), nat.ptr(pts), nat.ptr(info), nat.ptr(conf), 2))
st.synchronize()
off, nb = C.c_size_t(), C.c_size_t()
L.av_lane_workspace_view(0, S, h, w, MS, C.byref(off), C.byref(nb))
# front_pack<.., TIMED>: one record of four u64 per wave (= workgroup) (AVHOT_LANE_TIMED=1)
C = w // 4; nb = (h + 47) // 48; nfull = C // 62; rem = C - 62 * nfull; G = min(16, 64 // (rem + 2)) if rem else 0
ngr = (S + G - 1) // G if G else 0
nwg = S * nb * nfull + ngr * nb
raw = ws[off.value: off.value + nwg * 32].cpu().numpy().view(np.uint64).reshape(nwg, 4)[None]
cyc, real, t0 = raw[..., 0].astype(float), raw[..., 1].astype(float), raw[..., 2].astype(float)
print("waves", cyc.size, "cycles/wave mean %.0f min %.0f max %.0f" % (cyc.mean(), cyc.min(), cyc.max()))
print("realtime ticks (100MHz) mean %.1f -> %.2f us per wave; clock = %.3f GHz" % (real.mean(), real.mean() / 100.0, cyc.mean() / real.mean() * 0.1))
print("span of start times: %.1f us; last end - first start: %.1f us" % ((t0.max() - t0.min()) / 100.0, ((t0 + real).max() - t0.min()) / 100.0))
rows = 54
print("cycles per row trip per wave: %.0f" % (cyc.mean() / rows))

hw = raw[..., 3]
sect = (hw >> np.uint64(36)).astype(float)
print('Sobel+class section: %.0f cycles per wave = %.1f%% of the wave; per row %.0f cycles' % (sect.mean(), 100 * sect.mean() / cyc.mean(), sect.mean() / 56))
hwid = (hw & 0xFFFFFFFF).astype(np.uint64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
wave_id = (hwid & np.uint64(0xF)).astype(int); simd = ((hwid >> np.uint64(4)) & np.uint64(3)).astype(int)
cu = ((hwid >> np.uint64(8)) & np.uint64(0xF)).astype(int); sh = ((hwid >> np.uint64(12)) & np.uint64(1)).astype(int); se = ((hwid >> np.uint64(13)) & np.uint64(7)).astype(int)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0r = (t0 - t0.min()) / 100.0; t1r = t0r + real / 100.0
import collections
per = collections.defaultdict(list)
for k, a, b, sd in zip(key.ravel(), t0r.ravel(), t1r.ravel(), simd.ravel()):
    per[int(k)].append((a, b, int(sd)))
print("distinct CUs seen:", len(per))
cnt = sorted(len(v) for v in per.values())
print("waves per CU: min %d median %d max %d" % (cnt[0], cnt[len(cnt)//2], cnt[-1]))
ends = sorted(max(b for a, b, _ in v) for v in per.values())
print("CU finish time us: min %.1f median %.1f max %.1f" % (ends[0], ends[len(ends)//2], ends[-1]))
late = sum(1 for v in per.values() for a, b, _ in v if a > 5.0)
print("waves starting later than 5 us:", late)
# concurrency on one CU over time
k0 = sorted(per.keys())[0]
ev = sorted(per[k0])
print("CU", k0, "waves:", ["%.0f-%.0f s%d" % e for e in ev])
hist = np.histogram(t0r.ravel(), bins=[0, 1, 2, 5, 10, 20, 30, 40, 60, 80, 100, 130])
print("start-time histogram (us):", list(zip(hist[1][:-1].tolist(), hist[0].tolist())))
