"""profiles/<round>_kernel_resources.txt: compiler-reported registers / spills / scratch / static LDS / occupancy of every kernel in aria_amd/csrc
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3; no GPU needed) + scratch operations inside each gemm3 kernel's MFMA region (ISA).
    python tools/kernel_resources.py > profiles/r05_kernel_resources.txt"""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
print("# per-kernel resource usage of libaria_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3)")
print("# occ = waves/SIMD the register budget allows; LDS = STATIC bytes/block (the tiled GEMM / attention kernels take theirs dynamically)")
print(f"{'file':10s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'scratch':>7s} {'LDS':>6s} {'occ':>3s}  kernel")
for src in sorted(glob.glob("aria_amd/csrc/*.hip")):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Iaria_amd/csrc", "-c", src, "-o",
                          "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.split("\n"):
        m = re.search(r"remark: +(.*?): (.*?) \[-Rpass", line)
        if not m:
            m = re.search(r"remark: Function Name: (.*?) \[-Rpass", line)
            if m:
                cur = {"name": m.group(1)}
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            continue
        cur[key] = val
        if key == "LDS Size [bytes/block]":
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
            print(f"{os.path.basename(src):10s} {cur.get('VGPRs', '?'):>5s} {cur.get('AGPRs', '?'):>5s} {cur.get('TotalSGPRs', '?'):>5s} "
                  f"{cur.get('VGPRs Spill', '?'):>6s} {cur.get('ScratchSize [bytes/lane]', '?'):>7s} {val:>6s} {cur.get('Occupancy [waves/SIMD]', '?'):>3s}  {name}")
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Iaria_amd/csrc", "-S", "--cuda-device-only",
                      "aria_amd/csrc/gemm3.hip", "-o", "-"], capture_output=True, text=True).stdout.split("\n")
starts = [(i, re.match(r"^_ZN12_GLOBAL__N_112gemm3_kernelI(\w+?)EEv10GemmParams:", l).group(1)) for i, l in enumerate(asm)
          if re.match(r"^_ZN12_GLOBAL__N_112gemm3_kernelI\w+EEv10GemmParams:", l)]
print("\n# gemm3 kernels: scratch operations in the whole kernel / between the first and the last MFMA (the K loop lies inside that span)")
for n, (i, ver) in enumerate(starts):
    body = asm[i:starts[n + 1][0] if n + 1 < len(starts) else len(asm)]
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    sc = [j for j, l in enumerate(body) if "scratch_" in l]
    print(f"gemm3_kernel<{ver}>: {len(sc)} / {len([j for j in sc if mf and mf[0] < j < mf[-1]])}")
