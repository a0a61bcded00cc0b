#!/usr/bin/env python3
"""Loss per step of the bench step, launch plan on / off (same seed): where do they part?  usage: plan_ab.py [steps] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from devis_amd import TMSU, launch_plan  # noqa: E402
from devis_amd.optim import FusedAdam  # noqa: E402
from devis_amd.train import adjust_learning_rate  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NOSYNC = os.environ.get("NOSYNC") == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(plan):
    launch_plan.ENABLED = plan
    args = bench.brats_args(torch.bfloat16)
    torch.manual_seed(0)
    model = TMSU(args)
    model.train()
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((2, 4, S, S, S), generator=g).to(dev)
    y = torch.randint(0, 4, (2, S, S, S), generator=g).to(dev)
    out = []
    for _ in range(steps):
        adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
        _, loss = model(x, y, 25, 'train', 'BraTS')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        out.append(loss if NOSYNC else loss.item())
    return [v.item() if NOSYNC else v for v in out]


if os.environ.get("PLAN_FIRST") == "1":
    b = run(True)
    a = run(False)
else:
    a = run(False)
    b = run(True)
c = run(True)
for i, (u, v, w) in enumerate(zip(a, b, c)):
    print(i, u, v, w, "" if u == v == w else "  <-- differs")
