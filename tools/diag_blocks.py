"""Diagnostic: where does an evaluation forward leave the CPU oracle?  Per-block outputs against forward hooks on the oracle's
blocks, then the pieces of one bottleneck (conv1, conv2, squeeze, squeeze-excite output, shortcut).

    python tools/diag_blocks.py [clip_len=64] [resolution=64] [block index=12]

(This is how the pooled-sum overflow of the LDS-window convolution on 2 x 2 maps was found: every block within 0.8 % of the
oracle except s4.b1, and inside it every piece except the squeeze of the second image of each workgroup.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import stes_oracle as O
from spot_amd.e2e import E2EModel

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
om = O.Impl(7, "rny008_gsm", "gru", clip_len=T, predict_location=True)
O.randomize_(om, seed=21)
m = E2EModel(7, "rny008_gsm", "gru", clip_len=T, modality="rgb", device="cuda", predict_location=True, verbose=False)
m._model.load_state_dict(om.state_dict(), strict=True)
x = torch.randn(1, T, 3, R, R, generator=torch.Generator().manual_seed(22))
torch.set_num_threads(16)
om.eval(); m._model.eval()
outs = {}
feats = om._features
hooks = [feats.stem.register_forward_hook(lambda mod, i, o: outs.__setitem__("stem", o))]
names = []
for sname in ("s1", "s2", "s3", "s4"):
    for bname, blk in getattr(feats, sname).named_children():
        nm = f"{sname}.{bname}"
        names.append(nm)
        hooks.append(blk.register_forward_hook(lambda mod, i, o, nm=nm: outs.__setitem__(nm, o)))
with torch.no_grad():
    od = om(x)
    d = m._model(x.cuda())
eng = m._model.engine
N = T
def nchw(t, C, H, W):
    return t.float().cpu().view(N, H, W, C).permute(0, 3, 1, 2)
print("stem", rel(nchw(eng.X0, 32, eng.H0, eng.W0), outs["stem"]))
for blk, nm in zip(eng.blocks, names):
    o = outs[nm]
    print(nm, "HxW %dx%d" % (blk.Ho, blk.Wo), "out rel %.5f" % rel(nchw(blk.out, blk.cout, blk.Ho, blk.Wo), o), flush=True)
print("cnn_feat", rel(d["cnn_feat"].cpu(), od["cnn_feat"]))
# ---- inside one block: the pieces of the bottleneck against the oracle's submodules
bi = int(sys.argv[3]) if len(sys.argv) > 3 else 12
blk = eng.blocks[bi]
ob = None
k = 0
for sname in ("s1", "s2", "s3", "s4"):
    for bname, b_ in getattr(feats, sname).named_children():
        if k == bi:
            ob = b_
        k += 1
sub = {}
hs = [ob.conv1.register_forward_hook(lambda mod, i, o: sub.__setitem__("conv1", o)),
      ob.conv2.register_forward_hook(lambda mod, i, o: sub.__setitem__("conv2", o)),
      ob.se.register_forward_hook(lambda mod, i, o: sub.__setitem__("se", o)),
      ob.conv3.register_forward_hook(lambda mod, i, o: sub.__setitem__("conv3", o))]
if blk.down:
    hs.append(ob.downsample.register_forward_hook(lambda mod, i, o: sub.__setitem__("down", o)))
with torch.no_grad():
    om(x)
def act(raw, bn, C, H, W, relu=True):
    y = raw.float().view(-1, C) * bn.scale + bn.shift
    if relu:
        y = torch.relu(y)
    return y.cpu().view(N, H, W, C).permute(0, 3, 1, 2)
print("block", names[bi], "Mi", blk.Mi, "Mo", blk.Mo)
print(" conv1", rel(act(blk.c1, blk.bn1, blk.cmid, blk.Hi, blk.Wi), sub["conv1"]))
print(" conv2", rel(act(blk.c2, blk.bn2, blk.cmid, blk.Ho, blk.Wo), sub["conv2"]))
se_in = sub["conv2"]
gate_ref = (sub["se"] / se_in.clamp_min(1e-20)).mean((2, 3)) if False else None
se_out = act(blk.c2, blk.bn2, blk.cmid, blk.Ho, blk.Wo) * blk.segate.float().cpu().view(N, blk.cmid, 1, 1)
print(" se   ", rel(se_out, sub["se"]))
if not eng._eval_bn_live and blk.c3 is not None:
    pass
if blk.down:
    print(" down ", rel(act(blk.sc, blk.bnd, blk.cout, blk.Ho, blk.Wo, relu=False), sub["down"]))
pooled_ref = sub["conv2"].mean((2, 3))
print(" pooled", rel(blk.pooled.float().cpu().view(N, blk.cmid), pooled_ref))
pe = [(rel(blk.pooled.float().cpu().view(N, blk.cmid)[i], pooled_ref[i])) for i in range(N)]
print(" pooled per image:", " ".join("%.2f" % p for p in pe))
pc = [(rel(blk.pooled.float().cpu().view(N, blk.cmid)[:, c * 64:(c + 1) * 64], pooled_ref[:, c * 64:(c + 1) * 64])) for c in range(blk.cmid // 64)]
print(" pooled per channel block:", " ".join("%.2f" % p for p in pc))
