"""Per-block comparison of the HIP engine with the CPU oracle (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import stes_oracle as O
from spot_amd.e2e import E2EModel

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))

train = len(sys.argv) > 1 and sys.argv[1] == "train"
om = O.Impl(7, "rny008_gsm", "gru", clip_len=4, predict_location=True); O.randomize_(om, seed=3)
m = E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cuda", predict_location=True, verbose=False)
m._model.load_state_dict(om.state_dict())
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_rny008_gsm_gru.npz"))
x = torch.from_numpy(z["x"])
caps = {}
def hook(name):
    def f(mod, inp, out): caps[name] = out.detach()
    return f
feats = om._features
feats.stem.register_forward_hook(hook("stem"))
names = []
for s in ("s1", "s2", "s3", "s4"):
    for bn, blk in getattr(feats, s).named_children():
        nm = f"{s}.{bn}"; names.append(nm)
        blk.register_forward_hook(hook(nm))
        blk.conv1.register_forward_hook(hook(nm + ".conv1"))
        blk.conv1.gsm.register_forward_hook(hook(nm + ".gsm"))
        blk.conv2.register_forward_hook(hook(nm + ".conv2"))
        blk.se.register_forward_hook(hook(nm + ".se"))
        blk.conv3.register_forward_hook(hook(nm + ".conv3"))
O.set_storage_rounding(True)
om.train(train); m._model.train(train)
om._pred_fine._dropout.p = 0.0
with torch.no_grad():
    od = om(x)
    eng = m._model.engine
    eng.drop_p = 0.0
    logits, loc, feat = eng.forward(x.cuda().reshape(2, 4, 3, 64, 64), train=train)
torch.cuda.synchronize()
def nchw(t, n, h, w): return t.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2)
N = 8
print("stem", rel(nchw(eng.X0, N, eng.H0, eng.W0), caps["stem"]))
for nm, b in zip(names, eng.blocks):
    g = nchw(b.G, N, b.Hi, b.Wi)
    # bn+relu of conv outputs
    def act(raw, bn, h, w, relu=True):
        v = raw.float() * bn.scale + bn.shift
        if relu: v = torch.relu(v)
        return nchw(v, N, h, w)
    print(nm, "gsm", rel(g, caps[nm + ".gsm"]), "conv1", rel(act(b.c1.view(-1, b.cmid), b.bn1, b.Hi, b.Wi), caps[nm + ".conv1"]),
          "conv2", rel(act(b.c2.view(-1, b.cmid), b.bn2, b.Ho, b.Wo), caps[nm + ".conv2"]),
          "conv3", rel(act(b.c3.view(-1, b.cout), b.bn3, b.Ho, b.Wo, False), caps[nm + ".conv3"]),
          "out", rel(nchw(b.out.view(-1, b.cout), N, b.Ho, b.Wo), caps[nm]))
print("feat", rel(feat.float().cpu(), od["cnn_feat"]), "logits", rel(logits.cpu(), od["im_feat"]), "loc", rel(loc.cpu(), od["loc_feat"]))
