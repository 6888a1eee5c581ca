import os, sys, time, cProfile, pstats; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dnanet_amd.model import DNANet_UNet
from dnanet_amd.data import SyntheticDataset, Panel
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ds = SyntheticDataset(256, 5, 4096, seed=7)
panel = Panel(os.path.join(ROOT, 'tests/golden/panel.xml'))
for im in ds:
    im._scaler = np.linspace(65, 475, 4096); im._panel = panel
for caller in (False, True):
    m = DNANet_UNet(4, (3, 5), 32, device='cuda', apply_allele_caller=caller, dtype='bf16')
    with torch.no_grad():
        m._model.tensor('outputs.bias').fill_(-0.2)
    m.predict_batch(list(ds)[:16]); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    m.predict_batch(ds); torch.cuda.synchronize()
    pr.disable()
    print('==== caller', caller)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
