#!/usr/bin/env python3
"""Training step time of BASELINE config 2: Unet2D(1,2) evidential training, batch 16 of 1x256x256 (bf16 / fp32)."""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import TMSU
from devis_amd.optim import FusedAdam
from devis_amd.train import adjust_learning_rate

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
a = ap.parse_args()
args = argparse.Namespace(num_classes=2, dataset='TOAR', model_name='U', input_modality='four', end_epochs=99,
                          lambda_epochs=50, Uncertainty_Loss=False, lr=0.002,
                          compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
torch.manual_seed(0)
m = TMSU(args).train()
opt = FusedAdam(m.parameters(), lr=args.lr, weight_decay=1e-5, module=m)
g = torch.Generator().manual_seed(1234)
x = torch.randn((a.batch, 1, a.size, a.size), generator=g).cuda()
y = (torch.rand((a.batch, 1, a.size, a.size), generator=g) > 0.5).long().cuda()

def step():
    adjust_learning_rate(opt, 25, 99, args.lr)
    _, loss = m(x, y, 25, 'train', 'TOAR'); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    return loss

for _ in range(3):
    l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    l = step()
t_host = time.perf_counter() - t0            # the host is done enqueueing; what is left is the GPU's backlog
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print("host enqueue %.2f ms/step; GPU backlog at the end %.2f ms" % (t_host / a.steps * 1e3, (dt * a.steps - t_host) * 1e3))
px = a.batch * a.size * a.size
print("Unet2D(1,2) B=%d %dx%d %s: %.2f ms/step, %.1f M px/s, %.1f TFLOP/s, loss %.7f"
      % (a.batch, a.size, a.size, a.dtype, dt * 1e3, px / dt / 1e6, 3 * 367296 * px / dt / 1e12, l.item()))
