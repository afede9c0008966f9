"""Can a Python process that has initialised the GPU start child processes (subprocess / multiprocessing spawn)? (debug aid)"""
import subprocess
import sys

import torch

torch.zeros(4, device="cuda").sum().item()
print("gpu initialised:", torch.cuda.is_initialized(), flush=True)
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True)
print("subprocess rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
import torch.multiprocessing as mp


def w(q):
    q.put("spawn ok")


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=w, args=(q,))
    p.start()
    try:
        print(q.get(timeout=60), flush=True)
    except Exception as e:      # noqa: BLE001
        print("spawn failed", repr(e), flush=True)
    p.join(30)
    print("exitcode", p.exitcode)
