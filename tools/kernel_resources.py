"""Register / spill / occupancy table of the device code of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py egot2_amd/csrc/fused.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
cur = None; rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs','?'):>4} AGPR {v.get('AGPRs','?'):>4} spill {v.get('VGPRs Spill','?'):>4} scratch {v.get('ScratchSize [bytes/lane]','?'):>5} occ {v.get('Occupancy [waves/SIMD]','?')} lds {v.get('LDS Size [bytes/block]','?')}")
