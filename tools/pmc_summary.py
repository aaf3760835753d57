"""gpurun_out/<tag>_pmc_<target>_*.csv (tools/gpu_pmc.sh <tag> ...) -> profiles/<round>_pmc_<target>.json: per-launch averages per kernel, HBM-side bytes with
the guide's gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE is reported in KB and undercounts
16-byte-per-lane streams by 2x on gfx950; WRITE_SIZE in KB as reported), L2 hit rate, MFMA busy.

    python tools/pmc_summary.py <tag> <round, e.g. r05> fc1 [attn_bwd vit_fwd]"""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

ALG = {  # algorithmic bytes per launch
    "fc1": ("gemm3_kernel<false, true, 8>", 16384 * 6 * 2560 * 2 + 64 * 2560 * 3328 * 2 + 16384 * 6 * 3328 * 2 + 16384 * 6 * 1664 * 2,
            "experts.fc1 + SwiGLU on gathered rows, 98 304 routed rows, 64 experts: A rows (one per routed row: the algorithm's count; the "
            "gathered launch re-reads the 84 MB of tokens instead of a 503 MB permuted copy) + W read, h + act written"),
    "attn_bwd": ("attn_bwd", 65536 * 2560 * 2 * 8, "q, k, v, o, do read + dq, dk, dv written once (S = 65 536, 20 x 128)"),
    "vit_fwd": ("attn_fwd2_kernel<72", 16 * 4900 * 1152 * 2 * 4, "q, k, v read + o written once (16 x 4900 x 16 x 72)"),
}


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()[:80]


TAG, ROUND = sys.argv[1], sys.argv[2]
for target in sys.argv[3:]:
    pat, alg, alg_note = ALG[target]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{TAG}_pmc_{target}_*.csv")):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for kname, cs in vals.items():
        a = {c: sum(v) / len(v) for c, v in cs.items()}
        k = {"launches_averaged": len(cs.get("FETCH_SIZE", [])), **{c: round(v, 1) for c, v in a.items()}}
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            k["hbm_bytes_per_launch"] = round((2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1000)
        if "TCC_HIT_sum" in a:
            k["l2_hit_rate"] = round(a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
        kernels[kname] = k
    total = sum(k.get("hbm_bytes_per_launch", 0) for k in kernels.values())
    out = {"target": target, "what": f"rocprofv3 --pmc, separate passes, --kernel-trace only (tools/gpu_pmc.sh -> tools/pmc_targets.py {target})",
           "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_note": alg_note, "hbm_bytes_per_launch_all_kernels": total,
           "traffic_over_algorithmic": round(total / alg, 3) if total else None, "kernels": kernels,
           "notes": ["FETCH_SIZE (KB) doubled per the guide's gfx950 correction; it counts the L2s' fabric-side requests (infinity-cache hits included): an "
                     "upper bound of HBM reads"]}
    if target == "fc1":
        import bench

        out["kernel_tag"] = bench.PMC_KERNEL_TAG
        out["hbm_bytes_per_launch"] = total
    json.dump(out, open(os.path.join(root, "profiles", f"{ROUND}_pmc_{target}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])
