#!/usr/bin/env python3
"""Register / scratch use of every kernel of a HIP source (gfx950 device code, no GPU needed): compiles the file device-only, unbundles
the code object and reads the AMDGPU metadata notes.  usage: kernel_resources.py <file.hip> [--all]   (default: only kernels with
scratch or spills)"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
src = sys.argv[1]
d = tempfile.mkdtemp()
co, elf = os.path.join(d, "a.co"), os.path.join(d, "a.elf")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
                "--cuda-device-only", "-c", src, "-o", co], check=True, stderr=subprocess.DEVNULL)
subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co, "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], check=True)
out = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
for k in re.split(r"\n\s+- \.agpr_count", out)[1:]:
    g = lambda key: re.search(r"\." + key + r":\s+(\S+)", k)
    name, sp, ps, vg, lds = g("name"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("vgpr_count"), g("group_segment_fixed_size")
    ag = re.match(r":\s+(\d+)", k)          # (the split consumed the key of .agpr_count: its value leads the chunk)
    if not name:
        continue
    if "--all" in sys.argv or int(ps.group(1)) > 0 or int(sp.group(1)) > 0:
        dn = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(anonymous namespace\)::", "", dn)
        print("%-100s vgpr %3s agpr %3s spill %3s scratch %5s lds %6s" % (re.sub(r"\(.*", "", dn)[:100], vg.group(1), ag.group(1) if ag else "?", sp.group(1), ps.group(1), lds.group(1)))
