import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S, _lib
from unet_lane_detection_amd.trainer import UNetTrainer
feats = [16, 32, 64]; n, h, w = 3, 48, 64
sdn = S.seeded_state_dict(feats, seed=6)
frames = S.synthetic_frames(n, h, w, seed=2)
tgt = torch.from_numpy(S.synthetic_targets(n, h, w, seed=2))
# oracle with retained intermediate grads
sd = O.to_torch_state(sdn)
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if O.is_parameter(k)}
full = dict(sd); full.update(params)
taps = {}
logits = O.forward(full, O.normalize_u8_nhwc(frames), training=True, new_stats={}, taps=taps)
for t in taps.values(): t.retain_grad()
loss = O.bce_with_logits(logits, tgt); loss.backward()
lib = _lib.load()
lib.unet_train_debug_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
tr = UNetTrainer(sdn, device=0)
for j, name in [(2, "dec1"), (1, "dec0"), (0, "bottleneck")]:
    ref = taps[name].grad  # grad wrt the upconv input of decoder step j
    buf = torch.zeros(ref.numel(), device='cuda')
    lib.unet_train_debug_snapshot(tr._h, 100 + j, C.c_void_p(buf.data_ptr()), buf.numel())
    tr.forward_backward(torch.from_numpy(frames), tgt)
    got = buf.cpu().view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
    d = (got - ref).abs()
    print(name, "gA rel err", (d.max() / ref.abs().max()).item(), "shape", tuple(ref.shape))
    if d.max() / ref.abs().max() > 1e-4:
        bad = (d > 1e-4 * ref.abs().max()).nonzero()
        print(" bad count", len(bad), "of", ref.numel(), "first", bad[:10].tolist(), "last", bad[-5:].tolist())
        print(" bad n", sorted(set(bad[:, 0].tolist())), "bad y", sorted(set(bad[:, 2].tolist())), "bad x", sorted(set(bad[:, 3].tolist()))[:40], "bad c", len(set(bad[:,1].tolist())))
