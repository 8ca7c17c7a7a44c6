#!/usr/bin/env python3
"""Register / spill / occupancy check of the frame kernels (RT_RESOLVE_WAVES 8 and RT_SHSPATIAL_WAVES 7 fit without
spilling only because of -fno-slp-vectorize; another hipcc or EXTRA flag could spill again — correct results, large slow-down).

csrc/Makefile compiles restir_rt.hip with -Rpass-analysis=kernel-resource-usage (remarks only: the code is the same) and keeps
the remarks in build/resource_usage.log; this script turns them into a table (stdout, `--table FILE`) and FAILS (exit 1) when a
kernel named in BUDGET spills more vector registers or holds fewer wavefronts per SIMD than recorded here.

  python tools/check_resources.py cedec_2024_rt_amd/csrc/build/resource_usage.log [--table profiles/r05_kernel_resources.txt]
"""
import re
import subprocess
import sys

# kernel (demangled name up to the argument list, regex) -> (max spilled VGPRs, min wavefronts per SIMD)
# numbers = the build of round 5 (hipcc of ROCm 7.2.0, csrc/Makefile flags); the shadowed pass is the one kernel that is
# ALLOWED to spill (72 VGPRs + spills at 7 wavefronts per SIMD beat 112 VGPRs at 4: docs/MEASUREMENT_LOG_r04.md section 9)
BUDGET = [
    (r"^k_raycast<false>", 0, 8),
    (r"^k_raycast<true>", 0, 7),                                              # strips: work-sharing closest-hit walk
    (r"^k_generate_candidate<true, false, false, false, true, true>", 0, 6),   # the whole frame's stage 0: primary ray + candidates + temporal
    (r"^k_generate_candidate<true, false, false, false, true, false>", 0, 6),  # candidates + temporal (strips, rt_timing)
    (r"^k_generate_candidate<true, false, false, false, false, false>", 0, 5), # ... without the work-sharing walk (rt_tuning 13 = 0)
    (r"^k_generate_candidate<false, false, false, false, false, false>", 0, 6), # rt_generate_candidate
    (r"^k_spatial_coop<6, false, 256>", 0, 6),                                # the roofline kernel
    (r"^k_spatial_coop<6, true, 256>", 0, 5),                                 # strips: halo lists read / written in the pass
    (r"^k_resolve<", 0, 8),
    (r"^k_spatial<true, true>", 80, 7),
    (r"^k_tone_mapping", 0, 8),
    (r"^k_halo_mark<", 0, 8),
]


def parse(path):
    txt = open(path, errors="replace").read()
    rows = []
    names = []
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        names.append(b.split()[0])
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for b, d in zip(re.split(r"remark: Function Name: ", txt)[1:], dem):
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))  # noqa: E731
        # strip the argument list, keep the template arguments
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        name = d[:cut].replace("void ", "").replace("rt::", "")
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("TotalSGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         occ=g(r"Occupancy \[waves/SIMD\]"), spill=g("VGPRs Spill"), sspill=g("SGPRs Spill"), lds=g(r"LDS Size \[bytes/block\]")))
    return rows


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rows = parse(sys.argv[1])
    if not rows:
        raise SystemExit("no kernel-resource-usage remarks in " + sys.argv[1])
    lines = ["%-86s VGPR AGPR SGPR scratch occ spillV spillS   LDS" % "kernel"]
    seen = {}
    for r in rows:
        if r["name"] in seen or "rocprim" in r["name"]:  # the library's sort / scan kernels (BVH build) are not ours to budget
            continue
        seen[r["name"]] = r
        lines.append("%-86s %4d %4d %4d %7d %3d %6d %6d %5d" % (r["name"][:86], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["occ"], r["spill"], r["sspill"], r["lds"]))
    bad = []
    for pat, max_spill, min_occ in BUDGET:
        hit = [r for r in seen.values() if re.search(pat, r["name"])]
        if not hit:
            bad.append("no kernel matches %r (renamed? update tools/check_resources.py)" % pat)
        for r in hit:
            if r["spill"] > max_spill or r["occ"] < min_occ:
                bad.append("%s: %d spilled VGPRs (budget %d), %d wavefronts per SIMD (budget >= %d)" % (r["name"], r["spill"], max_spill, r["occ"], min_occ))
    table = "\n".join(lines) + "\n"
    if "--table" in sys.argv:
        open(sys.argv[sys.argv.index("--table") + 1], "w").write(table)
    elif "--quiet" not in sys.argv:
        sys.stdout.write(table)
    if bad:
        sys.stderr.write("kernel resource budget exceeded (tools/check_resources.py):\n  " + "\n  ".join(bad) + "\n")
        raise SystemExit(1)


if __name__ == "__main__":
    main()
