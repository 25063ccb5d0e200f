"""Resource usage of every kernel of libcrucible_hip.so as the compiler allocated it (-Rpass-analysis=kernel-resource-usage on the
same sources and flags as crucible_amd/csrc/Makefile): allocated VGPRs / SGPRs, scratch bytes, spill counts, waves per SIMD.
rocprofv3's `arch_vgpr_count` halves the allocation on gfx950 (64 for a 128-VGPR kernel); this is the figure bench.py reports.
usage: python scripts/kernel_resources.py > profiles/r03_kernel_resources.json   (CPU only, ~2 min)"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "crucible_amd", "csrc", "capi.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-Wno-unused-function", "--offload-device-only", "-c",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
txt = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
out = {}
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else None
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(cr::KernelArgs<\w+>\)$", "", d)
    out[d] = {"vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgprs": g("TotalSGPRs"), "scratch_bytes_per_lane": g(r"ScratchSize \[bytes/lane\]"),
              "sgpr_spills": g("SGPRs Spill"), "vgpr_spills": g("VGPRs Spill"), "waves_per_simd": g(r"Occupancy \[waves/SIMD\]")}
json.dump({"source": "hipcc -Rpass-analysis=kernel-resource-usage, flags of crucible_amd/csrc/Makefile", "kernels": out}, sys.stdout, indent=0, sort_keys=True)
