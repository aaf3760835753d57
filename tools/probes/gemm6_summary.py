"""gpurun_out/<tag>_gemm6_* (tools/probes/gemm6_probe.sh) -> profiles/<round>_gemm6_probe.json: plain timing of the 4-wave probe beside gemm3, and per
kernel the PMC averages per launch: SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, MFMA busy, waits, GRBM_GUI_ACTIVE and the effective clock
(GRBM_GUI_ACTIVE / launch duration, guide: DVFS give-back).      python tools/probes/gemm6_summary.py <tag> <round>"""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TAG, ROUND = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(gemm\d_kernel<[^>]*>)", name)
    return m.group(1) if m else name[:60]


out = {"what": "4 waves x 128 x 128 per wave (tools/probes/src/gemm6_4wave.hip: one wave per SIMD, 256 accumulator AGPRs, reads + LDS-DMA between the wave's own MFMAs, ONE "
               "barrier per K-tile, compiler-scheduled HIP) beside the product's gemm3 kernels (8 waves, two groups, 8 barriers per K-tile) through the C ABI, "
               "same operands, one process per shape (tools/probes/gemm6_probe.sh); SCHED = where tile t + 2's 16 DMA pieces go (0: all in the kk = 3 step; 1: 8 + 8 "
               "over kk = 3 / 0; 2: 4 per step)", "plain_timing": {}, "pmc": {}}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{TAG}_gemm6_*x*.json"))):
    txt = open(f).read().strip()
    if not txt:
        continue
    # (the probe prints one key per measurement; a key printed twice = the two interleaved rounds -> keep both)
    pairs = re.findall(r'"(gemm\d_\w+)": \{"ms": ([\d.]+), "tflops": ([\d.]+)\}', txt)
    d = collections.defaultdict(list)
    for k, ms, tf in pairs:
        d[re.sub(r"_round\d", "", k)].append(float(tf))
    shape = re.search(r'"shape": \[(\d+), (\d+), (\d+)\], "b_oc": (\d)', txt).groups()
    out["plain_timing"]["x".join(shape[:3]) + (" rc,oc" if shape[3] == "1" else " rc,rc")] = {k: v for k, v in d.items()}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{TAG}_gemm6_pmc_*.csv"))):
    shape = re.search(r"pmc_(\d+x\d+x\d+x\d)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        rec = out["pmc"].setdefault(shape, {}).setdefault(k, collections.defaultdict(list))
        rec[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"):
            rec["_dur_ns_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for shape, ks in out["pmc"].items():
    M, N, K, _ = (int(v) for v in shape.split("x"))
    for k, rec in ks.items():
        a = {c: sum(v) / len(v) for c, v in rec.items()}
        res = {c: round(v, 1) for c, v in a.items() if not c.startswith("_")}
        res["launches"] = len(rec.get("SQ_WAVE_CYCLES", []))
        if "GRBM_GUI_ACTIVE" in a:
            res["launch_us_in_GRBM_pass"] = round(a["_dur_ns_GRBM_GUI_ACTIVE"] / 1e3, 1)
            res["effective_clock_GHz"] = round(a["GRBM_GUI_ACTIVE"] / 8 / a["_dur_ns_GRBM_GUI_ACTIVE"], 3)   # (the counter is summed over the 8 XCDs)
            res["tflops_in_GRBM_pass"] = round(2.0 * M * N * K / a["_dur_ns_GRBM_GUI_ACTIVE"] / 1e3, 1)
        if "SQ_WAVE_CYCLES" in a:
            n_mfma = M * N * K / (32 * 32 * 16)
            res["wave_quad_cycles_per_mfma"] = round(a["SQ_WAVE_CYCLES"] / n_mfma, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
                res["mfma_busy_over_busy_x_simds"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_BUSY_CYCLES"] * 4 * 32), 4)   # (per-XCD busy cycles x 32 CUs x 4 SIMDs)
        ks[k] = res
json.dump(out, open(os.path.join(root, "profiles", f"{ROUND}_gemm6_probe.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
