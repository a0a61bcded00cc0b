import sys, os, time, torch, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
from devis_amd.train import adjust_learning_rate
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
m = TMSU(args); m.train()
from devis_amd.optim import FusedAdam
opt = FusedAdam(m.parameters(), lr=args.lr, weight_decay=1e-5, module=m)
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
def step():
    adjust_learning_rate(opt, 25, 99, args.lr)
    _, loss = m(x, y, 25, 'train', 'BraTS'); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter(); ts = []
    for _ in range(5):
        a = time.perf_counter(); step(); ts.append(time.perf_counter() - a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue per step ms:", ["%.2f" % (t * 1e3) for t in ts], "total enqueue %.2f ms, drain %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# phase split
torch.cuda.synchronize()
a = time.perf_counter(); _, loss = m(x, y, 25, 'train', 'BraTS'); b = time.perf_counter(); loss.backward(); c = time.perf_counter(); opt.step(); d = time.perf_counter()
torch.cuda.synchronize()
print("fwd enqueue %.2f ms, bwd enqueue %.2f ms, adam %.2f ms" % ((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3))
