#!/usr/bin/env python3
"""Per-kernel register / spill / LDS summary of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).
usage: kernel_resources.py reid-gan_amd/csrc/conv_igemm.hip [extra hipcc flags]"""
import re
import subprocess
import sys


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, universal_newlines=True).stderr
    cur = None
    rows = []
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "")
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
            cur = {"name": name}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"),
                         ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print("%-62s %5s %5s %5s %6s %7s %4s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "scratch", "occ", "lds"))
    for r in rows:
        print("%-62s %5d %5d %5d %6d %7d %4d %7d" % (r["name"][:62], r.get("vgpr", 0), r.get("agpr", 0), r.get("sgpr", 0), r.get("vspill", 0),
                                                 r.get("scratch", 0), r.get("occ", 0), r.get("lds", 0)))


if __name__ == "__main__":
    main()
