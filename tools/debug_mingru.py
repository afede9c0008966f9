import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import stes_oracle as O
from spot_amd.e2e import E2EModel
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))
om = O.Impl(7, "rny008_tsm", "mingru", clip_len=4, predict_location=True); O.randomize_(om, seed=3)
m = E2EModel(7, "rny008_tsm", "mingru", clip_len=4, modality="rgb", device="cuda", predict_location=True, verbose=False)
m._model.load_state_dict(om.state_dict())
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_rny008_tsm_mingru.npz"))
x = torch.from_numpy(z["x"])
om.eval(); m._model.eval()
caps = []
for l in om._pred_fine.rnn_layers:
    l.register_forward_hook(lambda mod, i, o: caps.append((i[0].detach(), o.detach())))
with torch.no_grad():
    od = om(x)
    eng = m._model.engine
    logits, loc, feat = eng.forward(x.cuda(), train=False)
torch.cuda.synchronize()
print("feat", rel(feat.float().cpu(), od["cnn_feat"]))
for i, (inp, out) in enumerate(caps):
    lz = om._pred_fine.rnn_layers[i]
    pre_ref = torch.cat([lz.linear_z(inp), lz.linear_h_tilde(inp)], -1)
    H = out.shape[-1]
    print(i, "pre", rel(eng.min_pre[i].view(8, 2 * H).float().cpu(), pre_ref.reshape(8, 2 * H)),
          "h", rel(eng.min_h[i].view(8, H).float().cpu(), out.reshape(8, H)), "|h|", float(out.abs().mean()), "|pre|", float(pre_ref.abs().mean()))
print("logits", rel(logits.cpu(), od["im_feat"]))
