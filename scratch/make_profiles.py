"""Turn the rocprofv3 output of scratch/profile_bench.sh (under gpurun_out/) into the committed files of profiles/.

    python scratch/make_profiles.py [round tag, default r01]

The dominant kernel is the one with the largest total duration in the inference-only kernel-stats pass; FETCH_SIZE /
WRITE_SIZE are summed per kernel name from the two counter passes (KiB, raw) and combined as 2 x FETCH + WRITE
(MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B)."""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"

def bench_line(log):
    lines = [l for l in open(os.path.join(G, log)) if l.startswith('{"metric"')]
    return json.loads(lines[-1])

shutil.copy(os.path.join(G, "prof_infer", "p_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_bf16_h6_infer_kernel_stats.csv"))
shutil.copy(os.path.join(G, "prof_full", "p_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_bf16_h6_kernel_stats.csv"))
infer_line, full_line = bench_line("prof_infer_bench.log"), bench_line("prof_full_bench.log")
json.dump(infer_line, open(os.path.join(P, f"{tag}_bench_bf16_h6_infer_benchline.json"), "w"))
json.dump(full_line, open(os.path.join(P, f"{tag}_bench_bf16_h6_benchline.json"), "w"))

stats = list(csv.DictReader(open(os.path.join(G, "prof_infer", "p_kernel_stats.csv"))))
dominant = max(stats, key=lambda r: float(r["TotalDurationNs"]))
name = dominant["Name"]

def per_kernel(counter, d):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(os.path.join(G, d, "p_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    rows = sorted(tot, key=lambda k: -tot[k])[:30]
    with open(os.path.join(P, f"{tag}_pmc_{counter.lower()}_per_kernel.csv"), "w") as f:
        f.write(f"kernel,launches,{counter}_total_KiB,{counter}_per_launch_KiB\n")
        for k in rows:
            f.write(f'"{k}",{cnt[k]},{tot[k]:.0f},{tot[k] / cnt[k]:.1f}\n')
    return tot[name] / max(cnt[name], 1) * 1024.0

fetch, write = per_kernel("FETCH_SIZE", "pmc_fetch"), per_kernel("WRITE_SIZE", "pmc_write")
sys.path.insert(0, ROOT)
from bench import kernel_source_digest  # noqa: E402  (the sources the dominant kernel is compiled from)
rec = {"kernel": infer_line["roofline"]["kernel"], "rocprof_kernel_name": name, "kernel_source_sha256": kernel_source_digest(),
       "rocprof_avg_launch_us": float(dominant["AverageNs"]) / 1e3, "bench_avg_launch_us": infer_line["roofline"]["avg_launch_us"],
       "fetch_size_bytes_per_launch_raw": fetch, "fetch_size_bytes_per_launch_corrected": 2 * fetch,
       "write_size_bytes_per_launch": write, "traffic_bytes_per_launch": 2 * fetch + write,
       "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "
               "--no-legs`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); averages over "
               "the launches of the dominant kernel (inference and data-gradient launches of the same instantiation)"}
json.dump(rec, open(os.path.join(P, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
print("inference:", infer_line["value"], "profiles/s; roofline", infer_line["roofline"])
print("train_step:", full_line.get("train_step"))

# matrix-core utilisation per kernel ("conv-layer MFMA utilisation from rocprof")
mf = os.path.join(G, "pmc_mfma", "p_counter_collection.csv")
if os.path.isfile(mf):
    acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(int)
    for r in csv.DictReader(open(mf)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            launches[r["Kernel_Name"]] += 1
    rows = sorted(acc, key=lambda k: -acc[k]["SQ_BUSY_CU_CYCLES"])[:30]
    with open(os.path.join(P, f"{tag}_mfma_utilisation_per_kernel.csv"), "w") as f:
        f.write("kernel,launches,mfma_busy_share_of_cu_busy_x4_simd,mfma_insts_per_launch,valu_insts_per_mfma,lds_insts_per_mfma,wave_wait_share\n")
        for k in rows:
            c = acc[k]
            if c["SQ_BUSY_CU_CYCLES"] <= 0:
                continue
            util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
            m = c["SQ_INSTS_MFMA"]
            per = (lambda v: f"{v / m:.2f}") if m > 0 else (lambda v: "")
            f.write(f'"{k}",{launches[k]},{util:.3f},{m / max(launches[k], 1):.0f},{per(c["SQ_INSTS_VALU"])},'
                    f'{per(c["SQ_INSTS_LDS"])},{c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0):.2f}\n')
    print(open(os.path.join(P, f"{tag}_mfma_utilisation_per_kernel.csv")).read()[:3000])

# achieved HBM bandwidth per kernel (north-star: "achieved HBM GB/s against MI355X peak" for the memory-bound stages): measured
# traffic per launch (2 x FETCH_SIZE + WRITE_SIZE of the counter passes) over the average duration of the single-stream
# kernel-trace pass (DN_TUNE_NO_WGRAD_STREAM=1: with the weight gradients on a second stream, overlapping kernels stretch each other)
def per_launch(counter, d):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(os.path.join(G, d, "p_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0; cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}
fe, wr = per_launch("FETCH_SIZE", "pmc_fetch"), per_launch("WRITE_SIZE", "pmc_write")
one = os.path.join(G, "prof_full_1stream", "p_kernel_stats.csv")       # single-stream pass: durations not stretched by overlap
full = {r["Name"]: r for r in csv.DictReader(open(one if os.path.isfile(one) else os.path.join(G, "prof_full", "p_kernel_stats.csv")))}
with open(os.path.join(P, f"{tag}_hbm_bandwidth_per_kernel.csv"), "w") as f:
    f.write("kernel,launches,avg_launch_us,hbm_read_MB_per_launch,hbm_write_MB_per_launch,achieved_GB_per_s,frac_of_8_TB_per_s,share_of_run_time_pct\n")
    for k, r in sorted(full.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"]))[:30]:
        if k not in fe:
            continue
        us = float(r["AverageNs"]) / 1e3
        rd, w_ = 2 * fe[k], wr.get(k, 0.0)
        gbs = (rd + w_) / us / 1e3
        f.write(f'"{k}",{r["Calls"]},{us:.1f},{rd / 1e6:.1f},{w_ / 1e6:.1f},{gbs:.0f},{gbs / 8000:.3f},{float(r["Percentage"]):.2f}\n')
print(open(os.path.join(P, f"{tag}_hbm_bandwidth_per_kernel.csv")).read())
