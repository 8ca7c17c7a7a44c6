#!/usr/bin/env python3
"""ISA-level budget of the frame kernels: compiles csrc/restir_rt.hip with --save-temps,
reads the gfx950 assembly and prints, as Markdown,

  * per kernel: VGPRs, SGPRs, scratch bytes per lane, LDS, waves per SIMD (compiler remarks), static instruction
    count by class, and
  * per LOOP of each kernel (a label that a later branch jumps back to): the vector instructions of one trip by
    class and their issue cost in SIMD cycles, priced with the per-instruction costs MEASURED on the MI355X
    (profiles/r02_valu_rates.json, tools/valu_rates.hip: wall time per wave64 instruction per SIMD at 8 waves/SIMD).

  python tools/isa_budget.py > profiles/r03_isa_budget.md        (CPU only: hipcc cross-compiles)
  python tools/isa_budget.py --class-costs profiles/isa_class_costs.json
      per kernel, the mean issue cost of the static instructions that fall into the two MIXED buckets of the dynamic
      class counters (SQ_INSTS_VALU_INT32, and "other" = SQ_INSTS_VALU minus all class counters: moves, selects, f32
      compares, min / max, div_scale / div_fixup): tools/profile_collect.py prices the dynamic counts with them
      (valu_issue_frac_weighted).

Classes: F = full rate (v_add/sub/mul/fma/fmac_f32, v_mov, integer add/shift/logic: ~2.5 cycles per wave64
instruction), C = half rate (min/max/cmp/cndmask/cvt/bfe/perm/alignbit/add3/mul_lo/mad_u64/div_scale/div_fmas/
div_fixup/packed f32: ~4.2), T = transcendental (rcp/sqrt/rsq/log/exp: ~8.2).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cedec_2024_rt_amd", "csrc", "restir_rt.hip")
# the product's flags (csrc/Makefile HIPFLAGS; r06: -fno-slp-vectorize had been missing here since round 4) + ISA_BUDGET_EXTRA
FLAGS = ["--offload-arch=gfx950:xnack-", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fno-gpu-flush-denormals-to-zero", "-Wno-unused-value"] + os.environ.get("ISA_BUDGET_EXTRA", "").split()

RATES = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_rates.json")))
GHZ = 2.4


def measured(op):
    e = RATES.get(op)
    return e["w8_wall_ns_per_instr_per_simd"] * GHZ if e else None


F_COST = sum(measured(o) for o in ("v_add_f32", "v_mul_f32", "v_sub_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_lshrrev_b32",
                                   "v_xor_b32", "v_and_b32", "v_or_b32")) / 10
FMA_COST = measured("v_fma_f32")
C_COST = sum(measured(o) for o in ("v_max_f32", "v_min_f32", "v_cmp_lt_f32", "v_cvt_f32_ubyte1", "v_bfe_u32", "v_perm_b32",
                                   "v_alignbit_b32", "v_add3_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_div_scale_f32", "v_div_fixup_f32",
                                   "v_cndmask_b32 (sgpr mask)", "v_max3_f32")) / 14
T_COST = sum(measured(o) for o in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_exp_f32")) / 5

F_OPS = re.compile(r"^v_(add|sub|subrev|mul|fmac|mac|mov|lshrrev|lshlrev|ashrrev|xor|and|or|not|add_co|addc_co|sub_co|subb_co)_(f32|u32|i32|b32|co_u32|b64)?")
T_OPS = re.compile(r"^v_(rcp|sqrt|rsq|log|exp|sin|cos)_")


def classify(op):
    if op.startswith("v_fma_f32"):
        return "F", FMA_COST
    if T_OPS.match(op):
        return "T", T_COST
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
                "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32",
                "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
        return "F", F_COST
    return "C", C_COST


def what(op):
    """coarse purpose buckets used in the text"""
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp")):
        return "IEEE divide"
    if op.startswith(("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_alignbit", "v_add3_u32")):
        return "PCG (64-bit LCG step, rotate)"
    if op.startswith("v_sqrt"):
        return "sqrt"
    if op.endswith(("_f64", "_f64_e32", "_f64_e64")) or "_f64_" in op:
        return "binary64 (sin / cos of rounds 1-2)"
    if op.startswith(("v_cvt_f32_ubyte", "v_cvt_f32_u")):
        return "byte -> float (quantised boxes)"
    if op.startswith(("v_max", "v_min", "v_med3")):
        return "min / max"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "compare / select"
    return "other arithmetic"


def bucket(op):
    """the dynamic class counter a static instruction is (presumably) counted by: SQ_INSTS_VALU_<bucket>, or OTHER"""
    b = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if T_OPS.match(b):
        return "TRANS_F32"
    if b.startswith("v_cvt_"):
        return "CVT"
    if b in ("v_add_f32", "v_sub_f32", "v_subrev_f32"):
        return "ADD_F32"
    if b == "v_mul_f32":
        return "MUL_F32"
    if b in ("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_div_fmas_f32"):
        return "FMA_F32"
    if b.startswith(("v_mad_u64", "v_mad_i64")) or b.endswith("_b64") or b.endswith("_u64") or b.endswith("_i64"):
        return "INT64"
    if re.match(r"^v_(add|sub|subrev|mul_lo|mul_hi|mul|mad|and|or|xor|not|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|bfe|bfi|add3|alignbit|perm|min|max|med3|add_co|addc_co|sub_co|subb_co|lshl_add|lshl_or|and_or|or3|xad|add_lshl|bcnt|ffbh|ffbl|mbcnt_lo|mbcnt_hi)\w*_(u32|i32|b32|u24|i24|u32_b32)$", b) or b.startswith(("v_cmp_", "v_cmpx_")) and b.endswith(("_u32", "_i32")):
        return "INT32"
    return "OTHER"


def compile_asm(tmp):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["--save-temps", "-c", SRC, "-o", os.path.join(tmp, "k.o")], cwd=tmp,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(tmp, "restir_rt-hip-amdgcn-amd-amdhsa-gfx950:xnack-.s")).read()


def resources(tmp):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", SRC, "-o", os.path.join(tmp, "r.o"), "-Rpass-analysis=kernel-resource-usage"],
                         cwd=tmp, capture_output=True, text=True).stderr
    cur, d = None, {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            d[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            d[cur][m.group(1).strip()] = int(m.group(2))
    return d


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def functions(asm):
    """{mangled: [lines]} of the kernel bodies"""
    fns, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            fns[cur] = []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            fns[cur].append(line)
    return fns


def loops(lines):
    """innermost-first list of (label, start, end) where a branch at `end` jumps back to `label` at `start`"""
    pos = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            pos[m.group(1)] = i
    out = []
    for i, line in enumerate(lines):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", line) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            out.append((m.group(1), pos[m.group(1)], i))
    return out


def histogram(lines):
    ops = collections.Counter()
    for line in lines:
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if m and not line.strip().startswith((";", ".")):
            ops[m.group(1)] += 1
    return ops


def summarise(ops):
    cls = collections.Counter()
    cyc = collections.Counter()
    buckets = collections.Counter()
    for op, n in ops.items():
        if not op.startswith("v_"):
            continue
        c, cost = classify(op)
        cls[c] += n
        cyc[c] += n * cost
        buckets[what(op)] += n * cost
    return cls, cyc, buckets


# r06: the template argument lists of the product's kernels as they are now (k_generate_candidate: <FUSE, SHADOWED, DEFER, PIPE, WS, RAYCAST>;
# k_spatial_coop: <WAVES, FUSED, TB>); the experiments library's forms (k_spatial_gather / _lds, k_resolve_stream) are found only when the
# tool compiles with -DRT_EXPERIMENTS (ISA_BUDGET_EXTRA="-DRT_EXPERIMENTS")
KERNELS = [("k_raycast", "k_raycast<false>("), ("k_raycast<WS>", "k_raycast<true>("),
           ("stage 0 as one launch: k_generate_candidate<..., WS, RAYCAST>", "k_generate_candidate<true, false, false, false, true, true>("),
           ("k_generate_candidate<true,false,WS>", "k_generate_candidate<true, false, false, false, true, false>("),
           ("k_generate_candidate<true,false>", "k_generate_candidate<true, false, false, false, false, false>("),
           ("k_resolve<WS>", "k_resolve<true>("), ("k_resolve_stream", "k_resolve_stream("),
           ("k_spatial_gather", "k_spatial_gather<6>("), ("k_spatial_coop", "k_spatial_coop<6, false, 256>("), ("k_spatial_coop<fused>", "k_spatial_coop<6, true, 256>("),
           ("k_halo_mark", "k_halo_mark<true>("), ("k_spatial_lds", "k_spatial_lds("), ("k_resolve", "k_resolve<false>("), ("k_spatial<true>", "k_spatial<true, true>("), ("k_spatial<true> per-lane records", "k_spatial<true, false>("),
           ("k_temporal<false>", "k_temporal<false>("), ("k_tone_mapping", "k_tone_mapping("),
           ("k_path_trace<9,false>", "k_path_trace<9, false>(")]
LOOP_KERNELS = {"k_raycast", "k_generate_candidate<true,false,WS>", "stage 0 as one launch: k_generate_candidate<..., WS, RAYCAST>", "k_spatial_gather", "k_spatial_coop", "k_halo_mark", "k_resolve<WS>", "k_spatial<true>"}


def class_costs(path):
    """{kernel: {bucket: {static instructions, mean cycles}}} of the frame kernels' static ISA"""
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_asm(tmp)
    fns = functions(asm)
    names = demangle(list(fns))
    out = {"costs_cycles": {"F": round(F_COST, 3), "FMA": round(FMA_COST, 3), "C": round(C_COST, 3), "T": round(T_COST, 3)},
           "source": "tools/isa_budget.py --class-costs: static gfx950 ISA of each kernel, instruction costs of profiles/r02_valu_rates.json"}
    for label, needle in KERNELS:
        cand = [m for m, d in names.items() if needle in d.replace("void ", "")]
        if not cand:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for op, n in histogram(fns[cand[0]]).items():
            if not op.startswith("v_"):
                continue
            _, cost = classify(op)
            a = agg[bucket(op)]
            a[0] += n
            a[1] += n * cost
        short = needle.rstrip("(")
        out[short] = {b: {"static": n, "mean_cycles": round(c / n, 3)} for b, (n, c) in sorted(agg.items())}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--class-costs":
        class_costs(sys.argv[2])
        return
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_asm(tmp)
        res = resources(tmp)
    fns = functions(asm)
    names = demangle(list(fns))
    print("# ISA budget of the frame kernels (gfx950, hipcc -O3 -ffp-contract=off)\n")
    print("Made by `tools/isa_budget.py`; costs from `profiles/r02_valu_rates.json` (measured on MI355X, `tools/valu_rates.hip`):")
    print(f"F (full rate) = {F_COST:.2f} cycles per wave64 instruction per SIMD (v_fma_f32 {FMA_COST:.2f}), C (half rate) = {C_COST:.2f}, "
          f"T (transcendental) = {T_COST:.2f}.\n")
    print("## Kernel resources and static instruction mix\n")
    print("| kernel | VGPR | AGPR | SGPR | scratch B/lane | LDS B | waves/SIMD | VALU static | F | C | T | SALU | VMEM | LDS ops |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    picked = []
    for label, needle in KERNELS:
        cand = [m for m, d in names.items() if needle in d.replace("void ", "")]
        if not cand:
            continue
        mang = cand[0]
        picked.append((label, mang))
        ops = histogram(fns[mang])
        cls, cyc, _ = summarise(ops)
        r = res.get(mang, {})
        salu = sum(n for o, n in ops.items() if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop")))
        vmem = sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_", "scratch_", "flat_")))
        lds = sum(n for o, n in ops.items() if o.startswith("ds_"))
        print(f"| `{label}` | {r.get('VGPRs', '?')} | {r.get('AGPRs', '?')} | {r.get('TotalSGPRs', r.get('SGPRs', '?'))} | {r.get('ScratchSize', '?')} | "
              f"{r.get('LDS Size', '?')} | {r.get('Occupancy', '?')} | {sum(cls.values())} | {cls['F']} | {cls['C']} | {cls['T']} | {salu} | {vmem} | {lds} |")
    print("\n## Loops (one trip): vector instructions by class, issue cost in SIMD cycles, and what they are for\n")
    for label, mang in picked:
        body = fns[mang]
        ls = loops(body)
        if not ls or label not in LOOP_KERNELS:
            continue
        print(f"### `{label}`\n")
        print("| loop (label, static lines) | VALU | F | C | T | issue cycles / trip | by purpose (cycles) |")
        print("|---|---|---|---|---|---|---|")
        # one row per label (its farthest back-branch); only loops that contain no other listed loop, plus the
        # outermost traversal loop bodies are interesting: keep innermost loops and loops of < 700 lines
        best = {}
        for lab, a, b in ls:
            if lab not in best or b > best[lab][2]:
                best[lab] = (lab, a, b)
        ls2 = sorted(best.values(), key=lambda t: t[1])
        inner = [t for t in ls2 if not any(o is not t and o[1] >= t[1] and o[2] <= t[2] for o in ls2)]
        keep = [t for t in ls2 if t in inner or (t[2] - t[1]) < 700]
        for lab, a, b in keep:
            ops = histogram(body[a:b + 1])
            cls, cyc, buckets = summarise(ops)
            n = sum(cls.values())
            if n < 30:
                continue
            by = ", ".join(f"{k} {v:.0f}" for k, v in buckets.most_common())
            print(f"| {lab} ({b - a + 1}) | {n} | {cls['F']} | {cls['C']} | {cls['T']} | {sum(cyc.values()):.0f} | {by} |")
        print()


if __name__ == "__main__":
    main()
