"""Turn the rocprofv3 output of scratch/profile_bench.sh (gpurun_out/r03_<dtype>/) into the committed files of profiles/.

    python scratch/make_profiles.py [round tag, default r03]

Per precision (fp32 = the headline, bf16): kernel statistics of the inference-only run, of the run with the training leg
(two streams, and one stream: the time base for per-kernel figures of the training step), the JSON lines those runs
printed, FETCH_SIZE / WRITE_SIZE and the matrix-core counters per kernel.  `<tag>_traffic.json` holds, per bench.py kernel
tag, the HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at
64 B; the counters are in KiB), which bench.py reports as `roofline.traffic` while the kernel's sources are unchanged."""
import csv, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
sys.path.insert(0, ROOT)
from bench import kernel_source_digest, TRAFFIC_SOURCES  # noqa: E402


def bench_line(path):
    lines = [l for l in open(path) if l.startswith('{"metric"')]
    return json.loads(lines[-1])


def stats(path):
    return list(csv.DictReader(open(path)))


def counters(path):
    """{kernel name: {counter: [total, launches]}}"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        c = acc[r["Kernel_Name"]][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
    return acc


def is_wgrad35(name, dt):
    return "conv_wgrad_kernel" in name and ("<float, 3, 5" in name if dt == "fp32" else "ELi3E" in name)


traffic = {}
for dt in ("fp32", "bf16"):
    O = os.path.join(G, f"{tag}_{dt}")
    if not os.path.isdir(O):
        print("no", O); continue
    for src, dst in (("prof_infer", "infer_kernel_stats"), ("prof_full", "kernel_stats"), ("prof_full_1stream", "kernel_stats_1stream")):
        shutil.copy(os.path.join(O, src, "p_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_{dt}_h6_{dst}.csv"))
    infer_line = bench_line(os.path.join(O, "prof_infer_bench.log"))
    full_line = bench_line(os.path.join(O, "prof_full_bench.log"))
    one_line = bench_line(os.path.join(O, "prof_full_1stream_bench.log"))
    json.dump(infer_line, open(os.path.join(P, f"{tag}_bench_{dt}_h6_infer_benchline.json"), "w"))
    json.dump(full_line, open(os.path.join(P, f"{tag}_bench_{dt}_h6_benchline.json"), "w"))
    json.dump(one_line, open(os.path.join(P, f"{tag}_bench_{dt}_h6_benchline_1stream.json"), "w"))

    fe, wr = counters(os.path.join(O, "pmc_fetch", "p_counter_collection.csv")), counters(os.path.join(O, "pmc_write", "p_counter_collection.csv"))
    for counter, acc in (("FETCH_SIZE", fe), ("WRITE_SIZE", wr)):
        rows = sorted(acc, key=lambda k: -acc[k][counter][0])[:30]
        with open(os.path.join(P, f"{tag}_pmc_{counter.lower()}_per_kernel_{dt}.csv"), "w") as f:
            f.write(f"kernel,launches,{counter}_total_KiB,{counter}_per_launch_KiB\n")
            for k in rows:
                t, n = acc[k][counter]
                f.write(f'"{k}",{n},{t:.0f},{t / n:.1f}\n')

    def bytes_of(pred):
        """(2 x FETCH + WRITE bytes, fetch bytes raw, write bytes, launches of the main kernel) summed over kernels matching pred"""
        f_ = sum(fe[k]["FETCH_SIZE"][0] for k in fe if pred(k)) * 1024.0
        w_ = sum(wr[k]["WRITE_SIZE"][0] for k in wr if pred(k)) * 1024.0
        return 2 * f_ + w_, f_, w_

    # ---- the inference leg's dominant kernel
    inf_stats = stats(os.path.join(O, "prof_infer", "p_kernel_stats.csv"))
    dom = max(inf_stats, key=lambda r: float(r["TotalDurationNs"]))
    name = dom["Name"]
    tot, f_raw, w_b = bytes_of(lambda k: k == name)
    n_l = fe[name]["FETCH_SIZE"][1]
    rt = infer_line["roofline"]
    traffic[rt["kernel"]] = {
        "kernel": rt["kernel"], "rocprof_kernel_name": name, "sources": TRAFFIC_SOURCES[rt["kernel"].split("<")[0]],
        "kernel_source_sha256": kernel_source_digest(TRAFFIC_SOURCES[rt["kernel"].split("<")[0]]),
        "rocprof_avg_launch_us": float(dom["AverageNs"]) / 1e3, "bench_avg_launch_us": rt["avg_launch_us"],
        "fetch_size_bytes_per_launch_raw": f_raw / n_l, "write_size_bytes_per_launch": w_b / n_l,
        "traffic_bytes_per_launch": tot / n_l,
        "note": f"separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 bench.py --dtype {dt} --steps 4 --warmup 1 --no-cpu-baseline "
                "--no-legs`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); average over every "
                "launch of this instantiation in that run (inference and data-gradient launches alike); rocprof_avg_launch_us from the "
                "inference-only kernel-trace pass"}
    print(dt, "inference:", infer_line["value"], "profiles/s;", rt["kernel"], "bench", rt["avg_launch_us"], "us, rocprof",
          round(float(dom["AverageNs"]) / 1e3, 1), "us; traffic/launch", round(tot / n_l / 1e6), "MB")

    # ---- the training leg's dominant kernel: the 3x5 weight gradient (all its instantiations + the slab reduction = one
    # dn_conv2d_wgrad call, which is what bench.py's event pairs bracket)
    one = {r["Name"]: r for r in stats(os.path.join(O, "prof_full_1stream", "p_kernel_stats.csv"))}
    main = [k for k in one if is_wgrad35(k, dt)]
    red = [k for k in one if "wgrad_reduce_conv_kernel" in k]
    launches = sum(int(one[k]["Calls"]) for k in main)
    us = sum(float(one[k]["TotalDurationNs"]) for k in main + red) / 1e3 / max(launches, 1)
    tot, f_raw, w_b = bytes_of(lambda k: is_wgrad35(k, dt) or "wgrad_reduce_conv_kernel" in k)
    n_l = sum(fe[k]["FETCH_SIZE"][1] for k in fe if is_wgrad35(k, dt))
    rt = one_line["train_step"]["roofline"]
    traffic[rt["kernel"]] = {
        "kernel": rt["kernel"], "rocprof_kernel_names": main + red, "sources": TRAFFIC_SOURCES["conv_wgrad"],
        "kernel_source_sha256": kernel_source_digest(TRAFFIC_SOURCES["conv_wgrad"]),
        "rocprof_avg_launch_us": us, "bench_avg_launch_us": rt["avg_launch_us"],
        "fetch_size_bytes_per_launch_raw": f_raw / max(n_l, 1), "write_size_bytes_per_launch": w_b / max(n_l, 1),
        "traffic_bytes_per_launch": tot / max(n_l, 1),
        "note": "per dn_conv2d_wgrad call = the split-K GEMM (every 3x5 instantiation) + its slab reduction; durations from the "
                "single-stream kernel-trace pass (DN_TUNE_NO_WGRAD_STREAM=1), counters from the --pmc passes (which serialise kernels)"}
    print(dt, "train step:", one_line["train_step"]["value"], "profiles/s (one stream);", full_line["train_step"]["value"], "(two);",
          rt["kernel"], "bench", rt["avg_launch_us"], "us, rocprof", round(us, 1), "us; traffic/launch", round(tot / max(n_l, 1) / 1e6), "MB")

    # ---- matrix-core utilisation per kernel ("conv-layer MFMA utilisation from rocprof")
    mf = counters(os.path.join(O, "pmc_mfma", "p_counter_collection.csv"))
    rows = sorted(mf, key=lambda k: -mf[k]["SQ_BUSY_CU_CYCLES"][0])[:30]
    with open(os.path.join(P, f"{tag}_mfma_utilisation_per_kernel_{dt}.csv"), "w") as f:
        f.write("kernel,launches,mfma_busy_share_of_cu_busy_x4_simd,mfma_insts_per_launch,valu_insts_per_mfma,lds_insts_per_mfma,wave_wait_share\n")
        for k in rows:
            c = {n: v[0] for n, v in mf[k].items()}
            nl = mf[k]["SQ_BUSY_CU_CYCLES"][1]
            if c.get("SQ_BUSY_CU_CYCLES", 0) <= 0:
                continue
            util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
            m = c["SQ_INSTS_MFMA"]
            per = (lambda v: f"{v / m:.2f}") if m > 0 else (lambda v: "")
            f.write(f'"{k}",{nl},{util:.3f},{m / max(nl, 1):.0f},{per(c["SQ_INSTS_VALU"])},{per(c["SQ_INSTS_LDS"])},'
                    f'{c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0):.2f}\n')

    # ---- achieved HBM bandwidth per kernel: counter traffic per launch over the single-stream average duration
    with open(os.path.join(P, f"{tag}_hbm_bandwidth_per_kernel_{dt}.csv"), "w") as f:
        f.write("kernel,launches,avg_launch_us,hbm_read_MB_per_launch,hbm_write_MB_per_launch,achieved_GB_per_s,frac_of_8_TB_per_s,share_of_run_time_pct\n")
        for k, r in sorted(one.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"]))[:32]:
            if k not in fe:
                continue
            u = float(r["AverageNs"]) / 1e3
            rd = 2 * fe[k]["FETCH_SIZE"][0] * 1024.0 / fe[k]["FETCH_SIZE"][1]
            w_ = wr[k]["WRITE_SIZE"][0] * 1024.0 / wr[k]["WRITE_SIZE"][1] if k in wr else 0.0
            gbs = (rd + w_) / u / 1e3
            f.write(f'"{k}",{r["Calls"]},{u:.1f},{rd / 1e6:.1f},{w_ / 1e6:.1f},{gbs:.0f},{gbs / 8000:.3f},{float(r["Percentage"]):.2f}\n')

json.dump(traffic, open(os.path.join(P, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("note", "rocprof_kernel_names")} for k, v in traffic.items()}, indent=1))
