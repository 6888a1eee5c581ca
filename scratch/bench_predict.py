import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dnanet_amd.model import DNANet_UNet
from dnanet_amd.data import SyntheticDataset, Panel
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ds = SyntheticDataset(n, 5, 4096, seed=7)
panel = Panel(os.path.join(ROOT, 'tests/golden/panel.xml'))
for im in ds:
    im._scaler = np.linspace(65, 475, 4096); im._panel = panel
for caller in (False, True):
    torch.manual_seed(0)
    m = DNANet_UNet(4, (3, 5), 32, device='cuda', apply_allele_caller=caller, dtype='bf16')
    # make the random-init net fire on ~3% of pixels so the caller has runs to chew on: bias the head
    with torch.no_grad():
        m._model.tensor('outputs.bias').fill_(-0.2)
    m.predict_batch(list(ds)[:16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    preds = m.predict_batch(ds)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); preds = m.predict_batch(ds); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    npos = float(np.mean([(p.image >= 0.5).mean() for p in preds[:8]]))
    ncall = sum(len(mk.alleles) for p in preds[:8] for mk in p.meta.get('called_alleles', [])) / 8 if caller else 0
    print(f'predict_batch N={n} allele_caller={caller}: {dt*1e3:.1f} ms  {n/dt:.0f} profiles/s  (positive px {npos:.3f}, alleles/profile {ncall:.1f})')
    # phases
    t0 = time.perf_counter(); x = m.get_inputs(list(ds)); torch.cuda.synchronize(); t1 = time.perf_counter()
    _, probs = m._model.infer(x, want_logits=False, want_probs=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    host = probs.movedim(1, -1).cpu().numpy(); t3 = time.perf_counter()
    print(f'   get_inputs {1e3*(t1-t0):.1f} ms, infer {1e3*(t2-t1):.1f} ms, D2H+numpy {1e3*(t3-t2):.1f} ms')
