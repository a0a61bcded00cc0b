#!/usr/bin/env python3
"""The batched weight packing (devis_pack_weights_batch) against the one-weight packing calls, every U-Net layer."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops, Unet
torch.manual_seed(0)
m = Unet(4, 16, 4).cuda()
ent = [(n + ".weight", mod.weight.detach(), "convT" if isinstance(mod, nn.ConvTranspose3d) else "conv")
       for n, mod in m.named_modules() if isinstance(mod, (nn.Conv3d, nn.ConvTranspose3d))]
for dtype in (torch.float32, torch.bfloat16):
    pk = ops.WeightPacker(ent, dtype, torch.device("cuda", 0))
    views = pk.pack()
    torch.cuda.synchronize()
    print("packed", dtype, flush=True)
    bad = 0
    for n, w, kind in ent:
        ref = ops.pack_conv_weight(w, dtype) if kind == "conv" else ops.pack_convT_weight(w, dtype)
        bad += int(not torch.equal(views[n][0], ref))
        if dtype == torch.bfloat16 and kind == "conv":
            co, ci = w.shape[0], w.shape[1]
            t = w.numel() // (co * ci)
            # against the forward packing re-packed by devis_repack_conv_weight_dgrad: both are in the layer's own
            # layout (canonical [Cin][T][Cout], or fragment order for the 64 / 128-channel layers, csrc/layout.hip)
            wd = torch.empty((ci, t, co), dtype=dtype, device=w.device)
            from devis_amd import _lib
            _lib.check(_lib.load().devis_repack_conv_weight_dgrad(ref.data_ptr(), wd.data_ptr(), co, ci, t, _lib.DEVIS_BF16,
                                                                  torch.cuda.current_stream().cuda_stream), "repack")
            if not torch.equal(views[n][1], wd):
                bad += 1
                print("   dgrad packing differs:", n, tuple(w.shape))
    torch.cuda.synchronize()
    print(dtype, "mismatching packings:", bad, flush=True)
