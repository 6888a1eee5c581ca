"""Prints the measured errors the tolerances in tests/test_train_gpu.py / test_unet_infer_gpu.py are derived from:
per golden fixture the worst per-tensor relative L2 error and the cosine of the bf16 gradients, and the number of mask
pixels that differ from the reference in fp32 / bf16 mode.   python tests/tools/diag_bf16_bounds.py   (needs the GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from conftest import golden_names, load_golden
from dnanet_amd.unet import HipAdam, HipUNet

dev = torch.device("cuda:0")
for name in golden_names():
    g = load_golden(name)
    cfg = (int(g["depth"]), tuple(int(v) for v in g["kernel"]), int(g["filters"]))
    x = torch.from_numpy(g["x"]).to(dev)
    for dtype in ("fp32", "bf16"):
        m = HipUNet(*cfg, dev, dtype=dtype)
        m.load_state_dict(g["sd0"])
        m.eval()
        logits, probs = m.infer(x)
        ref = torch.from_numpy(g["logits_eval"])
        diff = int(((probs.cpu() >= 0.5) != (torch.sigmoid(ref) >= 0.5)).sum())
        err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
        line = f"{name:28s} {dtype}: eval logits rel err {err:.2e}, mask pixels differing {diff} of {ref.numel()}"
        if "real" not in name:
            m.train()
            t = torch.from_numpy(g["t"]).to(dev)
            opt = HipAdam(m, lr=float(g["lr"]), weight_decay=float(g["wd"]))
            opt.zero_grad()
            lg = m(x)
            sums = m.dice_forward(lg, t)
            m.backward()
            keys = [k for k in g["grad0"] if not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"))]
            l2 = {k: float((m.grad(k).cpu() - g["grad0"][k]).norm() / (g["grad0"][k].norm() + 1e-20)) for k in keys}
            a = torch.cat([m.grad(k).cpu().flatten() for k in keys]); b = torch.cat([g["grad0"][k].flatten() for k in keys])
            worst = max(l2, key=l2.get)
            line += f"; train loss err {abs(float(sums[4]) - float(g['step0/loss'])):.2e}, worst grad rel-L2 {l2[worst]:.3f} ({worst}), " \
                    f"median {sorted(l2.values())[len(l2) // 2]:.3f}, cosine {float(torch.dot(a, b) / (a.norm() * b.norm())):.4f}"
        print(line, flush=True)
