"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_usage.py shg_vqa_amd/csrc/gemm.hip [name filter]"""
import os, re, subprocess, sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if src.endswith("attention.hip") else []
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage"] + extra + ["-c", src, "-o", "/tmp/_usage.o"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
filt = "c++filt"
for b in blocks:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip().replace("shg::", "").replace("__hip_bfloat16", "bf16")
    d = re.sub(r"\(.*", "", d)
    if flt and flt not in d:
        continue
    print("V%3d A%3d scratch %4d occ %d S%3d LDS %6d  %s" % (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                                                              g("SGPRs"), g(r"LDS Size \[bytes/block\]"), d[:170]))
