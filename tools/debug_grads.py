"""Per-parameter gradient comparison of the HIP engine with the CPU oracle (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import stes_oracle as O
from spot_amd.e2e import E2EModel

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))

om = O.Impl(7, "rny008_gsm", "gru", clip_len=4, predict_location=True); O.randomize_(om, seed=3)
m = E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cuda", predict_location=True, verbose=False)
m._model.load_state_dict(om.state_dict())
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_rny008_gsm_gru.npz"))
x, label, xy = torch.from_numpy(z["x"]), torch.from_numpy(z["label"]), torch.from_numpy(z["xy"])
om.train(); om._pred_fine._dropout.p = 0.0
od = om(x)
oloss, _, _ = O.stes_loss(od["im_feat"], od["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
oloss.backward()
m._model._pred_fine._dropout.p = 0.0
eng = m._model.engine; eng.drop_p = 0.0
out = m.train_step(x.cuda(), label.cuda(), xy.cuda(), lr=1e-3, optimizer_step=False)
torch.cuda.synchronize()
print("loss", out[:3].cpu().numpy(), float(oloss))
og = dict((n, p.grad) for n, p in om.named_parameters() if p.grad is not None)
for n, p in m._model.named_parameters():
    if p.grad is None or n not in og: continue
    print(f"{rel(p.grad.cpu(), og[n]):8.4f}  |g|={float(og[n].norm()):.3e}  {n}")
