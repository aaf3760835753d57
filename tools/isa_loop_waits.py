"""List the `s_waitcnt vmcnt(..)` a compiled .hip file has INSIDE loops, with the instruction that follows (hipcc -S output on stdin or a
path).  A full drain (vmcnt(0)) inside a loop whose next consumer is not the staged data itself is the pattern to look for: a wait that
belongs to a value loaded in front of the loop lands at its first use inside it and then drains the loop's own prefetch every iteration."""
import re, subprocess, sys
src = sys.argv[1]
asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Iaria_amd/csrc", "-S", "--cuda-device-only", src, "-o", "-"],
                     capture_output=True, text=True).stdout.split("\n")
kernel, in_loop_labels, loop_depth = None, set(), {}
out = {}
for i, line in enumerate(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kernel = m.group(1)
        continue
    if kernel is None:
        continue
    if "in Loop:" in line or "Loop Header" in line:
        cur_in_loop = True
    if re.match(r"^\.LBB\d+_\d+:", line):
        cur_in_loop = ("Loop" in line)
    if "s_waitcnt" in line and "vmcnt" in line and globals().get("cur_in_loop"):
        nxt = [a.strip() for a in asm[i + 1:i + 6] if a.strip() and not a.strip().startswith(";")][:2]
        out.setdefault(kernel, []).append((line.strip(), nxt))
for k, v in out.items():
    print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110])
    for w, nxt in v:
        print("    ", w, "->", " | ".join(n[:60] for n in nxt))
