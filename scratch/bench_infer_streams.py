"""Does the eval forward gain from two sub-batches on two HIP streams (one's HBM-bound shallow layers under the other's
MFMA-bound deep layers)?  python scratch/bench_infer_streams.py [rows] [dtype]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnanet_amd.unet import HipUNet
from dnanet_amd.synthetic import make_batch
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = HipUNet(4, (3, 5), 32, dev, dtype=dtype); m.eval()
x = torch.from_numpy(make_batch(256, rows, 4096, seed=1)[0]).to(dev)
probs = torch.empty_like(x)
m.infer(x, want_logits=False)

def single(reps):
    for _ in range(reps):
        m._forward_eval_chunk(x, None, probs)

def split(reps, parts, streams):
    n = x.shape[0] // parts
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for r in range(reps):
        for p in range(parts):
            with torch.cuda.stream(streams[p % len(streams)]):
                m._forward_eval_chunk(x[p * n:(p + 1) * n], None, probs[p * n:(p + 1) * n])
    for s in streams: cur.wait_stream(s)

def timed(fn, *a):
    fn(2, *a[1:]) if a else fn(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(*a) if a else fn(10)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3

print(f"{dtype} rows {rows}")
print(f"one stream, 256             : {timed(single, 10):.3f} ms")
ref = probs.clone()
for parts, ns in ((2, 1), (2, 2), (4, 2), (4, 4), (8, 2)):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    probs.zero_()
    ms = timed(split, 10, parts, streams)
    print(f"{parts} parts on {ns} stream(s)      : {ms:.3f} ms   equal {bool(torch.equal(ref, probs))}")
