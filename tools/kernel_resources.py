"""Per-kernel register / scratch / LDS usage from hipcc -save-temps assembly (run: python tools/kernel_resources.py).
Flags what costs occupancy silently: VGPR counts above 128, scratch (private segment: spills or dynamically indexed arrays)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(ROOT, "boslam_amd", "csrc")
tmp = tempfile.mkdtemp()
for f in sorted(os.listdir(src)):
    if not f.endswith(".hip"):
        continue
    base = f[:-4]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + src,
                    "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(src, f), "-o", os.path.join(tmp, base + ".o"), "-save-temps"],
                   cwd=tmp, stderr=subprocess.DEVNULL)
    s = open(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for blk in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        m = re.match(r"_Z\d+(k_[a-z_0-9]+)", name)
        if not m:
            continue                                   # library kernels (hipcub sorts / scans inside ba.hip)
        short = (m.group(1) if m else name) + ("<" + ",".join(re.findall(r"L[ib](\d+)E", name)) + ">" if "IL" in name else "")
        print(f"{base:6s} {short:28s} vgpr {g('vgpr_count'):4d} spill {g('vgpr_spill_count'):4d} scratch {g('private_segment_fixed_size'):5d} B  lds {g('group_segment_fixed_size'):6d} B")
