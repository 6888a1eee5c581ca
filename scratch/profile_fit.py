"""rocprofv3 target: the fit() leg of bench.py alone (configs[2]: 348 profiles, batch 16, 5 dyes, HIP-graph replay).
    rocprofv3 --kernel-trace --stats -d OUT -o p --output-format csv -- python3 scratch/profile_fit.py [bf16|fp32]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, torch
import bench
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
print(json.dumps(bench.fit_leg(dt, torch.device("cuda:0"), epochs=6)))
