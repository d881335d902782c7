"""Development aid: compares per-stage training tensors (unet_train_debug_snapshot) with the oracle taps to
locate the first stage where a gradient diverges.  Test infrastructure (imports oracle/)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S, _lib
from unet_lane_detection_amd.trainer import UNetTrainer
feats = [16, 32, 64]; n, h, w = 3, 48, 64
sdn = S.seeded_state_dict(feats, seed=6)
frames = S.synthetic_frames(n, h, w, seed=2)
tgt = torch.from_numpy(S.synthetic_targets(n, h, w, seed=2))
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
D = 3
units = [(2*D+2+1, "decoder_blocks.1.3"), (2*D+2+0, "decoder_blocks.1.0"), (2*D+2+3, "decoder_blocks.3.3")]
for uid, name in units:
    zt = taps["z/" + name]
    for stage, ref, what in [(400, zt.detach(), "z"), (300, zt.grad, "dZ")]:
        buf = torch.zeros(ref.numel(), device='cuda')
        lib.unet_train_debug_snapshot(tr._h, stage + uid, C.c_void_p(buf.data_ptr()), buf.numel())
        tr.forward_backward(torch.from_numpy(frames), tgt)
        got = buf.cpu().view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
        d = (got - ref).abs()
        print(name, what, "rel err", (d.max() / ref.abs().max()).item())
        if d.max() / ref.abs().max() > 1e-4:
            bad = (d > 1e-4 * ref.abs().max()).nonzero()
            print(" bad count", len(bad), "of", ref.numel(), "n", sorted(set(bad[:, 0].tolist())), "y", sorted(set(bad[:, 2].tolist())), "x", sorted(set(bad[:, 3].tolist())), "nc", len(set(bad[:, 1].tolist())))

print("---- detail for decoder_blocks.1.3")
name = "decoder_blocks.1.3"; uid = 2*D+2+1
zt = taps["z/" + name]; ref = zt.grad
buf = torch.zeros(ref.numel(), device='cuda')
lib.unet_train_debug_snapshot(tr._h, 300 + uid, C.c_void_p(buf.data_ptr()), buf.numel())
tr.forward_backward(torch.from_numpy(frames), tgt)
got = buf.cpu().view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
d = (got - ref).abs().amax(dim=(0, 2, 3))
c = int(d.argmax()); print("bad channel", c, "err", d[c].item(), "others max", d[torch.arange(64) != c].max().item())
zc = zt.detach()[:, c]
print("z mean", zc.mean().item(), "std", zc.std().item(), "min", zc.min().item(), "max", zc.max().item())
print("ref dZ sample", ref[0, c, 0, :6].tolist()); print("got dZ sample", got[0, c, 0, :6].tolist())
print("diff sample", (got - ref)[0, c, 0, :6].tolist(), (got - ref)[2, c, 11, -6:].tolist())
st = torch.zeros(4 * 64, device='cuda')
lib.unet_train_debug_snapshot(tr._h, 500 + uid, C.c_void_p(st.data_ptr()), st.numel())
tr.forward_backward(torch.from_numpy(frames), tgt)
st = st.cpu().view(4, 64)
print("scale, shift, mean, invstd @c:", st[:, c].tolist())
zz = zt.detach(); mu = zz.mean(dim=(0, 2, 3)); var = zz.var(dim=(0, 2, 3), unbiased=False)
print("oracle mean, invstd @c:", mu[c].item(), (1 / (var[c] + 1e-5).sqrt()).item())
gd = tr.grad_dict()
print("dgamma got/ref", gd["decoder_blocks.1.4.weight"][c].item(), params["decoder_blocks.1.4.weight"].grad[c].item())
print("dbeta got/ref", gd["decoder_blocks.1.4.bias"][c].item(), params["decoder_blocks.1.4.bias"].grad[c].item())
