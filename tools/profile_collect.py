"""Turns the rocprofv3 result databases of tools/profile_round.sh (gpurun_out/<tag>/) into the
summaries committed under profiles/: <tag>_bench.json, <tag>_kernel_stats.csv,
<tag>_pmc_fetch_write.json, <tag>_pmc_valu.json and spatial_pmc_latest.json (read by bench.py)."""
import sys, os, json, sqlite3, collections, glob, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def db(path):
    f = glob.glob(os.path.join(src, path, "*.db"))
    return sqlite3.connect(f[0]) if f else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# kernel-trace stats: name, calls, total/avg/min/max duration (ns), share
con = db("stats")
# bench.py also renders a few 3840x2160 frames (its secondary line): launches are keyed by kernel AND grid,
# so the 1920x1080 rows are exactly the launches the bench line's numbers are about
raw = collections.defaultdict(list)
try:  # launches in time order, so that the first ones of every kernel can be set aside (r06)
    it = list(con.execute("select name, duration, grid_x, grid_y from kernels order by start"))
except sqlite3.Error:
    it = list(con.execute("select name, duration, grid_x, grid_y from kernels"))  # insertion order = time order
for name, dur, gx, gy in it:
    raw[(short(name), gx * max(gy, 1))].append(dur)
# r06: a kernel's first launches of a process include code-object load, cold caches and cold page tables
# (k_generate_candidate<..., true>: 2.02 ms against 0.63 steady); the STEADY columns drop the first WARM launches of every (kernel,
# grid) group — bench.py's own warm-up is 5 frames — and are what DESIGN.md quotes and what the utilisation figures below divide by
WARM = 8


def steady(v):
    return v[WARM:] if len(v) >= 2 * WARM else v
grids = collections.defaultdict(set)
for (k, g) in raw:
    grids[k].add(g)
rows = collections.defaultdict(list)
rows_steady = collections.defaultdict(list)
for (k, g), v in raw.items():
    # the smallest grid of a frame kernel is the 1080p launch and keeps the plain name
    key = k if g == min(grids[k]) or not k.startswith("k_") else f"{k} [grid {g}, 3840x2160 frames]"
    rows[key] += v
    rows_steady[key] += steady(v)
total = sum(sum(v) for v in rows.values())
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write(f"Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,SteadyCalls(first {WARM} of each grid dropped),SteadyAverageNs,SteadyMedianNs\n")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        sv = sorted(rows_steady[k])
        f.write(f'"{k}",{len(v)},{sum(v)},{sum(v)/len(v):.1f},{min(v)},{max(v)},{100.0*sum(v)/total:.3f},{len(sv)},{sum(sv)/len(sv):.1f},{sv[len(sv)//2]}\n')
stats_avg_ms = {k: sum(v) / len(v) * 1e-6 for k, v in rows_steady.items()}


def counters(path):
    con = db(path)
    if con is None:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rows_ = list(con.execute("select kernel_name, counter_name, value, grid_size from counters_collection"))
    gmin = {}
    for k, c, v, g in rows_:
        gmin[short(k)] = min(gmin.get(short(k), g), g)
    for k, c, v, g in rows_:
        if g == gmin[short(k)] or not short(k).startswith("k_"):  # frame kernels: the 1920x1080 launches only (see above)
            agg[short(k)][c].append(v)
    # steady launches only, as the durations they are divided by (the counter passes run --steps 16 --warmup 2: 3 dropped)
    return {k: {c: dict(launches=len(v[3:] if len(v) >= 9 else v), mean=sum(v[3:] if len(v) >= 9 else v) / len(v[3:] if len(v) >= 9 else v)) for c, v in cs.items()} for k, cs in agg.items()}


fw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fw[c] = {k: dict(launches=v[c]["launches"], mean_KB=v[c]["mean"]) for k, v in counters("pmc_" + c).items() if c in v}
json.dump(fw, open(os.path.join(dst, f"{tag}_pmc_fetch_write.json"), "w"), indent=1)

# class-weighted vector-issue utilisation (r01-r03 priced every instruction at 4 cycles, which put
# k_raycast above 1): dynamic class counts (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _INT64, _CVT) x the issue
# costs MEASURED on this chip (profiles/r02_valu_rates.json: full rate 2.4 / fma 2.8 / half rate 4.2 / transcendental 8.2
# cycles per wave64 instruction per SIMD); the two mixed buckets (INT32; "other" = total - classes: moves, selects, f32
# compares, min / max) at the mean cost of the kernel's own static instructions of that bucket
# (profiles/isa_class_costs.json, tools/isa_budget.py --class-costs); clock = GRBM_GUI_ACTIVE / kernel time.
CLS = {"ADD_F32": "F", "MUL_F32": "F", "FMA_F32": "FMA", "TRANS_F32": "T", "CVT": "C", "INT64": "C"}
try:
    ISA = json.load(open(os.path.join(dst, "isa_class_costs.json")))
except Exception:
    ISA = {"costs_cycles": {"F": 2.384, "FMA": 2.765, "C": 4.209, "T": 8.152}}


def weighted_valu(kernel, e, ms):
    """adds valu_issue_frac_weighted (and its ingredients) to the counter dict e of one kernel"""
    total = e.get("SQ_INSTS_VALU")
    if not total or not ms or e.get("SQ_INSTS_VALU_INT32") is None:
        return
    cost = ISA["costs_cycles"]
    own = None
    for k, v in ISA.items():
        if isinstance(v, dict) and "OTHER" in v and (k == kernel or k.split("<")[0] == kernel.split("<")[0]):
            own = v
            if k == kernel:
                break
    c_int = own["INT32"]["mean_cycles"] if own and "INT32" in own else 3.3
    c_oth = own["OTHER"]["mean_cycles"] if own else 3.55
    cyc, classes = 0.0, 0.0
    for name, cl in CLS.items():
        n = e.get("SQ_INSTS_VALU_" + name, 0.0) or 0.0
        cyc += n * cost[cl]
        classes += n
    n_int = e.get("SQ_INSTS_VALU_INT32", 0.0) or 0.0
    other = max(total - classes - n_int, 0.0)
    cyc += n_int * c_int + other * c_oth
    ghz = 2.4
    if e.get("GRBM_GUI_ACTIVE"):
        g = e["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e6)  # busy cycles of the launch (the counter sums the 8 XCDs) / its duration
        if 1.5 < g < 3.2:
            ghz = g
    e["valu_cycles_per_inst_weighted"] = cyc / total
    e["clock_GHz_from_GRBM_GUI_ACTIVE"] = ghz
    e["valu_other_bucket_share"] = other / total
    e["valu_issue_frac_weighted"] = cyc / 1024.0 / (ms * 1e6 * ghz)


valu = counters("pmc_valu")
for k, cs in counters("pmc_class").items():
    valu.setdefault(k, {}).update({c: v for c, v in cs.items() if c not in valu.get(k, {})})
mix = counters("pmc_mix")
stall = counters("pmc_stall")
for k, cs in stall.items():
    mix.setdefault(k, {}).update({c: v for c, v in cs.items() if c not in mix.get(k, {})})
out = {}
for k, cs in valu.items():
    if not k.startswith("k_") or "bvh" in k:
        continue
    e = {c: v["mean"] for c, v in cs.items()}
    e.update({c: v["mean"] for c, v in mix.get(k, {}).items()})
    if k in stats_avg_ms and e.get("SQ_INSTS_VALU"):
        ms = stats_avg_ms[k]
        # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD (16 lanes) for 4 cycles
        e["avg_ms_kernel_trace"] = ms
        e["valu_issue_frac_4cycle_convention_r01_r03"] = e["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e6 / ms
        weighted_valu(k, e, ms)
        if e.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        if e.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if e.get(c) is not None:
                    e[c + "_share_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
    out[k] = e
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_valu.json"), "w"), indent=1)

shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))

# ---- shadowed-target mode (tools/shadowed_frames.py under the same passes): per-kernel trace + counters
scon = db("sh_stats")
if scon is not None:
    srows = collections.defaultdict(list)
    for name, dur in scon.execute("select name, duration from kernels"):
        srows[short(name)].append(dur)
    stot = sum(sum(v) for v in srows.values())
    with open(os.path.join(dst, f"{tag}_shadowed_kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for k, v in sorted(srows.items(), key=lambda kv: -sum(kv[1])):
            f.write(f'"{k}",{len(v)},{sum(v)},{sum(v)/len(v):.1f},{min(v)},{max(v)},{100.0*sum(v)/stot:.3f}\n')
    s_avg = {k: sum(v) / len(v) * 1e-6 for k, v in srows.items()}
    sh = {}
    for path in ("sh_pmc_FETCH_SIZE", "sh_pmc_WRITE_SIZE", "sh_pmc_valu", "sh_pmc_class", "sh_pmc_stall"):
        for k, cs in counters(path).items():
            if k.startswith("k_") and "bvh" not in k:
                sh.setdefault(k, {}).update({c: v["mean"] for c, v in cs.items()})
    for k, e in sh.items():
        if k in s_avg:
            e["avg_ms_kernel_trace"] = s_avg[k]
            if e.get("SQ_INSTS_VALU"):
                e["valu_issue_frac_4cycle_convention_r01_r03"] = e["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e6 / s_avg[k]
                weighted_valu(k, e, s_avg[k])
                if e.get("SQ_WAVES"):
                    e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
            if e.get("SQ_WAVE_CYCLES"):
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if e.get(c) is not None:
                        e[c + "_share_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
    json.dump(sh, open(os.path.join(dst, f"{tag}_shadowed_pmc.json"), "w"), indent=1)

# spatial kernel HBM traffic, corrected as calibrated in profiles/r01_fetch_calibration.json
sp = [k for k in fw["FETCH_SIZE"] if k.startswith(("k_spatial_coop", "k_spatial_gather", "k_spatial_lds", "k_spatial<false"))]
if sp:
    k = sp[0]
    W, H = 1920, 1080
    fetch = fw["FETCH_SIZE"][k]["mean_KB"] * 1024
    write = fw["WRITE_SIZE"][k]["mean_KB"] * 1024
    streamed = W * H * 112  # own G-buffer (32 B) + own record (64 B) + own radiance (16 B), read as 64-B-stride streams
    corrected = fetch + streamed / 2
    sha = None
    shaf = os.path.join(src, "lib.sha256")
    if os.path.exists(shaf):
        sha = open(shaf).read().split()[0]
    bid = None
    bidf = os.path.join(src, "lib.build_id")
    if os.path.exists(bidf):
        bid = open(bidf).read().split()[0]
    vfrac = {kk: round(e["valu_issue_frac_weighted"], 3) for kk, e in out.items()
             if "valu_issue_frac_weighted" in e and kk.split("<")[0] in ("k_raycast", "k_generate_candidate", "k_resolve", "k_spatial", "k_spatial_coop", "k_spatial_gather", "k_spatial_lds")}
    json.dump({
        "kernel": k, "round": tag, "build_id": bid, "lib_sha256": sha, "valu_issue_frac": vfrac,
        "valu_issue_frac_definition": "class-weighted: dynamic class counters x measured issue costs / (1024 SIMDs x kernel time x measured clock); tools/profile_collect.py",
        "workload": "blocks_restir stand-in 1920x1080, bench options",
        "FETCH_SIZE_KB_per_launch": fetch / 1024, "WRITE_SIZE_KB_per_launch": write / 1024,
        "read_bytes_corrected": corrected, "streamed_read_bytes_known": streamed,
        "hbm_bytes_per_launch": corrected + write,
        "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE in KB, separate --pmc passes. Calibrated on this pool with "
                "tools/fetch_calib.hip (profiles/r01_fetch_calibration.json): a 64-B-stride record stream read with 4 x dwordx4 per "
                "lane reports exactly 1/2 of its bytes (the guide's gfx950 x2 correction), random 64-B record gathers report 1.00 of "
                "their bytes. k_spatial streams 112 B/pixel and gathers the rest, so reads = FETCH_SIZE + streamed/2; WRITE_SIZE is exact.",
        "source": f"profiles/{tag}_pmc_fetch_write.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
    }, open(os.path.join(dst, "spatial_pmc_latest.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(json.dumps({k: {c: (round(v, 3) if v < 100 else round(v)) for c, v in e.items()} for k, e in out.items()}, indent=0))
