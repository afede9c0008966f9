"""The data-parallel trainer with TWO ranks on the real engine.  Both ranks are fresh child processes that share the one
GPU of the test box; the process group is gloo, so gradient buckets and broadcasts are staged through host memory by a
small adapter with torch.distributed's calling convention (RCCL needs one GPU per rank; the trainer only sees the
adapter).  Checked: the bucketed all-reduce issued from inside backward delivers the SUM of the two ranks' gradients, each
taken with the GLOBAL loss denominators (reference train_e2e_spatial.py:569-572,589-591), and both ranks hold identical
weights after AdamW."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _HostStaged:
    """torch.distributed's all_reduce / broadcast on device tensors through host copies (gloo)."""

    def __init__(self, dist):
        self._d, self.ReduceOp = dist, dist.ReduceOp
        self.local = []                                    # what this rank contributed to every large all-reduce, in call order

    class _Done:
        def wait(self):
            return True

    def all_reduce(self, t, op=None, async_op=False):
        h = t.detach().cpu()
        if h.numel() > 2:
            self.local.append(h.clone())
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(h)
        return self._Done()

    def broadcast(self, t, src=0):
        h = t.detach().cpu()
        self._d.broadcast(h, src=src)
        t.copy_(h)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import stes_oracle as O
    from spot_amd import dp, lib
    from spot_amd.e2e import E2EModel
    # deterministic mode: every cross-workgroup float sum in a fixed order, so that a rank's contribution to the reduction can be
    # compared with a separate run of the same half batch bit for bit (without it two runs differ by 20-30 % in the backbone)
    lib.set_deterministic(True)

    def build(seed):
        om = O.Impl(7, "rny008_gsm", "gru", clip_len=4, predict_location=True)
        O.randomize_(om, seed=seed)
        m = E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cuda", predict_location=True, verbose=False)
        m._model.load_state_dict(om.state_dict(), strict=True)
        m._model._pred_fine._dropout.p = 0.0
        m._model.engine.drop_p = 0.0
        return m

    g = torch.Generator().manual_seed(21)                       # the global batch: four clips, two per rank
    x = torch.randn(4, 4, 3, 64, 64, generator=g)
    label = torch.randint(0, 7, (4, 4), generator=g)
    label[0, 0] = 0; label[3, 1] = 2
    xy = torch.rand(4, 4, 2, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    xr, lr_, xyr = x[sl].cuda(), label[sl].cuda(), xy[sl].cuda()
    den = dp.loss_denominators(label.cuda(), 5.0)               # over ALL ranks' labels
    # this rank's own gradient with the global denominators, no optimizer step
    m1 = build(5)
    m1.train_step(xr, lr_, xyr, lr=1e-3, ext_denoms=den, optimizer_step=False)
    torch.cuda.synchronize()
    own = m1._model.engine.flat_grad.detach().cpu().clone()
    # the data-parallel step (rank 1 starts from other weights: the trainer must broadcast rank 0's)
    m2 = build(5 if rank == 0 else 6)
    staged = _HostStaged(dist)
    trainer = dp.DataParallelTrainer(m2, staged, lr=1e-3)
    staged.local = []                                      # (drop the broadcasts' bookkeeping: only the step's buckets from here)
    own_out = m1._model.engine.loss_buf.detach().cpu().clone()
    out = trainer.step(xr, lr_, xyr).cpu()
    torch.cuda.synchronize()
    eng = m2._model.engine
    torch.save({"own": own, "own_out": own_out, "reduced": eng.flat_grad.detach().cpu(), "weights": eng.flat.detach().cpu(), "out": out,
                "den": den.cpu(), "buckets": dict(eng.buckets), "local": staged.local}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_engine_step_sums_gradients_and_keeps_replicas_identical(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=110)
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(2))
    # both ranks hold the same reduced gradient and the same weights after AdamW, bit for bit
    assert torch.equal(r0["reduced"], r1["reduced"]) and torch.equal(r0["weights"], r1["weights"])
    # every rank divided by the same denominators, and they are the sums of the ranks' own class-weight sums / event counts
    assert torch.equal(r0["den"], r1["den"]) and (r0["out"][3:5] + r1["out"][3:5]).tolist() == r0["den"].tolist()
    # the all-reduce delivered, bucket by bucket in the order backward completes them, exactly the sum of what the two ranks
    # put in (fp32, two addends: no rounding freedom) -- nothing missing, nothing reduced twice, every bucket in its place
    order = ["heads", "s4", "s3", "rest"]
    assert len(r0["local"]) == len(r1["local"]) == 4
    covered = 0
    for name, a, b in zip(order, r0["local"], r1["local"]):
        lo, hi = r0["buckets"][name]
        assert a.numel() == hi - lo
        assert torch.equal(r0["reduced"][lo:hi], a + b), name
        covered += hi - lo
    assert covered == r0["reduced"].numel()
    # what a rank put in is its own gradient under the GLOBAL denominators: bit for bit the gradient of a separate run of the same
    # half batch (both in deterministic mode; per-rank denominators would double it)
    for r in (r0, r1):
        assert torch.equal(r["out"][:3], r["own_out"][:3])
        local = torch.cat([r["local"][order.index(n)] for n in sorted(order, key=lambda n: r["buckets"][n][0])])
        assert torch.equal(local, r["own"])
    # the summed loss partials are the global-batch loss: each rank normalised by the global denominators
    tot = (r0["out"][:3] + r1["out"][:3]).numpy()
    assert np.isfinite(tot).all() and abs(tot[0] - (tot[1] + tot[2])) < 1e-4 * max(1.0, abs(tot[0]))
