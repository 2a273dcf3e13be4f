"""Turns the scratch output of one profiling call (gpurun_out/<tag>_*) into the committed summaries under profiles/:
kernel stats, bench lines, the PMC summary with the VALU issue model, the large tier's per-kernel traffic.
usage: python tools/make_profiles.py r02"""
import collections, csv, glob, io, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def first(pattern):  # the newest match: gpurun merges every call's files into the same directories
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


# 1. kernel stats of the headline bench
f = first(os.path.join(G, tag + "_stats", "*", "*kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(P, tag + "_kernel_stats.csv"))
for name in ("bench", "bench_driver", "bench_chains", "bench_chains1", "bench_chains4", "stats_bench", "bench_share2"):
    src = os.path.join(G, "%s_%s.json" % (tag, name))
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(P, "%s_%s.json" % (tag, name)), "w").write(lines[-1] + "\n")
for name in ("other_configs", "config3_full_1gpu", "length_probe", "stamps_256", "stamps_2048", "stamps_4096", "queue_probe",
             "chain_stamps_1", "chain_stamps_2", "resident_ab", "dispatch_probe", "gap_probe", "host_path", "large_decode_probe",
             "large80_decode_kernels", "large_ties", "other_configs_256", "fixed_cost", "fast_left", "c3_kstats16", "c3_kstats256",
             "fuzz_soak"):
    src = os.path.join(G, "%s_%s.txt" % (tag, name))
    if os.path.exists(src):
        txt = "\n".join(l for l in open(src).read().splitlines() if "amdgpu.ids" not in l)
        open(os.path.join(P, "%s_%s.txt" % (tag, name)), "w").write(txt + "\n")

# 1b. per-launch durations of the headline kernel in that run (plain single-stream calls: no two launches overlap)
f = first(os.path.join(G, tag + "_stats", "*", "*kernel_trace.csv"))
if f:
    rows = [r for r in csv.DictReader(open(f)) if "k_compress<1, 5, false, 256" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    with open(os.path.join(P, tag + "_kernel_launches.txt"), "w") as o:
        o.write("k_compress<1,5,false,256>: %d launches of `%s` in start order, duration in us (rocprofv3 --kernel-trace); "
                "launches that overlap their predecessor: %d\n" % (len(d), "bench.py --chains 1 --no-pipeline --steps 20 --warmup 5",
                                                                    sum(1 for g in gaps if g < 0)))
        o.write("mean %.2f  min %.2f  max %.2f  last 20 (the plain-call timed loop the bench line's roofline reads): mean %.2f\n"
                % (sum(d) / len(d), min(d), max(d), sum(d[-20:]) / len(d[-20:])))
        for i in range(0, len(d), 20):
            o.write(" ".join("%.1f" % v for v in d[i:i + 20]) + "\n")

# 2. PMC summary + VALU issue model
agg = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(G, tag + "_pmc", "pass*"))):
    f = first(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "k_compress" in r["Kernel_Name"] and "256" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    launches = len(next(iter(agg.values())))
    f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
    tr64 = c.get("SQ_INSTS_VALU_TRANS_F64", 0)
    cvt = c.get("SQ_INSTS_VALU_CVT", 0)
    valu = c.get("SQ_INSTS_VALU", 0)
    other = valu - f64 - tr64 - cvt
    cycles = other * 2 + f64 * 4 + tr64 * 16 + cvt * 4
    floor_us = cycles / 1024.0 / 2400.0
    frames = c.get("SQ_WAVES", 40960)
    fetch = 2 * 1024 * c.get("FETCH_SIZE", 0)
    write = 1024 * c.get("WRITE_SIZE", 0)
    out = {
        "note": "rocprofv3 --pmc passes (tools/pmc_collect.sh: one counter set per pass, --kernel-trace only) over "
                "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end`; averages per "
                "k_compress<1,5,false,256> launch (%d launches).  FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes "
                "for gfx950 (128-B requests tallied at 64 B); FETCH / WRITE unit KiB; SQ_WAVE_CYCLES, SQ_WAIT_*, "
                "SQ_ACTIVE_INST_* count quad-cycles." % launches,
        "counters_per_launch": c,
        "hbm_bytes_per_launch": {"read": fetch, "write": write, "total": fetch + write},
        "per_frame": {"valu": valu / frames, "valu_f64_add_mul_fma": f64 / frames, "valu_f64_trans": tr64 / frames,
                      "valu_cvt": cvt / frames, "salu": c.get("SQ_INSTS_SALU", 0) / frames, "lds": c.get("SQ_INSTS_LDS", 0) / frames,
                      "smem": c.get("SQ_INSTS_SMEM", 0) / frames,
                      "wave_cycles": 4 * c.get("SQ_WAVE_CYCLES", 0) / frames},
        "wave_time_split": {k: c.get(k, 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1) for k in
                            ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
    }
    json.dump(out, open(os.path.join(P, tag + "_pmc.json"), "w"), indent=1)
    model = {
        "issue_floor_us_per_launch": floor_us,
        "source": "%s_pmc.json: (%.1f M other VALU x 2 + %.1f M f64 add/mul/fma x 4 + %.2f M f64 rcp/rsq x 16 + %.1f M "
                  "conversions x 4) cycles / 1024 SIMDs / 2.4 GHz" % (tag, other / 1e6, f64 / 1e6, tr64 / 1e6, cvt / 1e6),
        "valu_wave_instructions_per_launch": valu,
    }
    json.dump(model, open(os.path.join(P, "valu_model.json"), "w"), indent=1)
    print("VALU floor %.1f us per launch; HBM %.1f MB" % (floor_us, (fetch + write) / 1e6))

# 3. large tier: kernel stats and traffic per kernel
for nf in (80, 256):
    st = os.path.join(G, "%s_large%d" % (tag, nf))
    f = first(os.path.join(st, "*", "*kernel_stats.csv"))
    if not f:
        continue
    shutil.copy(f, os.path.join(P, "%s_large%d_kernel_stats.csv" % (tag, nf)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), st, st + "_fetch", st + "_write"],
                       capture_output=True, text=True)
    head = ("%d frames x 131072 samples (%.1f MB of samples; mixed classes, auto, e = 5 %%): per kernel, average duration, HBM "
            "bytes read (FETCH_SIZE doubled, gfx950) and written per launch, from three rocprofv3 runs of tools/large_trace.py "
            "(--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE)\n" % (nf, nf * 131072 * 8 / 1e6))
    open(os.path.join(P, "%s_large%d_traffic.txt" % (tag, nf)), "w").write(head + r.stdout)
    print(r.stdout)


# 4. decoder traffic + the file bench.py reads for roofline.traffic
def path_bytes(stats_dir, kernels):
    """HBM bytes per call of a path = sum over its kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), stats_dir, stats_dir + "_fetch", stats_dir + "_write"],
                       capture_output=True, text=True)
    tot = 0.0
    for l in r.stdout.splitlines()[1:]:
        w = l.split()
        if len(w) >= 6 and any(k in l for k in kernels):
            tot += (float(w[-3]) + float(w[-2])) * 1e6
    return tot, r.stdout


traffic = {"note": "HBM bytes from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 rule + WRITE_SIZE, "
                   "--kernel-trace only), per launch of the named kernel or per call of the named path; written by tools/make_profiles.py"}
pj = os.path.join(P, tag + "_pmc.json")
if os.path.exists(pj):
    traffic["k_compress_256"] = {"bytes": json.load(open(pj))["hbm_bytes_per_launch"]["total"], "file": tag + "_pmc.json"}
st = os.path.join(G, tag + "_large80")
if first(os.path.join(st, "*", "*kernel_stats.csv")):
    b, _ = path_bytes(st, ("k_large_", "k_pack_", "k_compress_large"))
    traffic["chunker_compress"] = {"bytes": b, "file": tag + "_large80_traffic.txt"}
st = os.path.join(G, tag + "_ldec80")
if first(os.path.join(st, "*", "*kernel_stats.csv")):
    b, txt = path_bytes(st, ("k_large_dparse", "k_large_trip243<true", "k_decompress_large"))
    traffic["chunker_decompress"] = {"bytes": b, "file": tag + "_large80_decode_traffic.txt"}
    open(os.path.join(P, tag + "_large80_decode_traffic.txt"), "w").write(
        "80 frames x 131072 samples decoded (tools/large_decode_trace.py, NF=80 KLASS=mix; the trace also holds the compress call that "
        "made the records): per kernel, average duration, HBM bytes read / written per launch\n" + txt)
st = os.path.join(G, tag + "_dec256")
if first(os.path.join(st, "*", "*kernel_stats.csv")):
    b, txt = path_bytes(st, ("k_decompress<",))
    traffic["decompress_f256"] = {"bytes": b, "file": tag + "_decode256_traffic.txt"}
    open(os.path.join(P, tag + "_decode256_traffic.txt"), "w").write(
        "configs[2] in 256-sample frames decoded (tools/decode_trace.py): per kernel, average duration, HBM bytes read / written per launch\n" + txt)
if len(traffic) > 1:
    json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
