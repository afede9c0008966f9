"""Quick timing of the fused training step at a given size (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spot_amd.e2e import E2EModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 224
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.manual_seed(0)
m = E2EModel(7, "rny008_gsm", "gru", clip_len=T, modality="rgb", device="cuda", predict_location=True, verbose=False)
with torch.no_grad():
    for n, p in m._model.named_parameters():
        if "conv3D.weight" in n: p.normal_(0, 1.0 / p[0].numel() ** 0.5)
        if n.endswith("conv3.bn.weight"): p.fill_(0.25)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, T, 3, R, R, device="cuda", generator=g)
label = (torch.rand(B, T, device="cuda", generator=g) < 0.03).long() * torch.randint(1, 7, (B, T), device="cuda", generator=g)
xy = torch.rand(B, T, 2, device="cuda", generator=g)
for i in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.time()
    out = m.train_step(x, label, xy, lr=1e-3)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"step {i}: {1e3*(t1-t0):.1f} ms  loss {out[:3].tolist()}  {B*T/(t1-t0):.0f} frames/s  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
