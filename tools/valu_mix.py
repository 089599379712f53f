#!/usr/bin/env python3
"""Static VALU instruction mix of every kernel in libdrfe.so, priced with the measured issue costs of
profiles/<tag>_valu_issue.txt (tools/ubench_valu.hip: cycles per wave64 instruction per SIMD at 8 waves per SIMD):

    python tools/valu_mix.py [tag=r03] > profiles/<tag>_valu_mix.json

On gfx950 the plain 32-bit VOP1 / VOP2 forms (v_add_u32, v_and_b32, v_fma_f32, v_mov_b32, 16-bit scalar ops ...) issue a
wave64 instruction every ~2.3 cycles once two waves share the SIMD; everything the byte-wise image kernels are built from -
packed 16-bit ops, three-input min / max, v_perm, v_dot4, v_alignbit, DPP moves, 32-bit max / min, multiplies, conversions,
all f64 - takes ~4.1, transcendentals ~8.  A kernel's average cost per VALU instruction is therefore its own number:
mix-weighted from the disassembly (static counts stand in for dynamic ones: the hot loops dominate both).  Instructions the
table does not hold are priced by their encoding class (VOP3 / packed / DPP / f64 -> 4.1, plain e32 -> 2.3)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
cost = {}
issue = os.path.join(ROOT, "profiles", f"{tag}_valu_issue.txt")
if not os.path.exists(issue):              # the issue costs are the hardware's: a round that did not re-measure them prices with the newest table
    import glob
    issue = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_issue.txt")))[-1]
for ln in open(issue):
    f = ln.split()
    if len(f) >= 6 and f[0].startswith("v_") and f[1] == "8":
        cost[f[0]] = float(f[4])
cost.pop("v_cndmask_b32", None)          # its row measures a serial VCC chain, not the pipe

objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
import shutil
import tempfile
tmp = tempfile.mkdtemp(prefix="drfe_co_")
# objdump --offloading drops the extracted code objects NEXT TO its input: work on a copy in a scratch directory so that
# nothing lands in the source tree (round 3 committed twenty such files by accident)
so = shutil.copy(os.path.join(ROOT, "dr_slam_amd", "csrc", "libdrfe.so"), os.path.join(tmp, "libdrfe.so"))
# the device code object is embedded in .hip_fatbin: let objdump --offloading find it
dis = subprocess.run([objdump, "-d", "--offloading", so], capture_output=True, text=True, cwd=tmp).stdout
if "v_" not in dis:
    # extract the bundle by hand
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, f"{tmp}/fat.bin"], check=True)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", f"--input={tmp}/fat.bin", "--unbundle",
                          "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/dev.co"], capture_output=True, text=True)
    dis = subprocess.run([objdump, "-d", f"{tmp}/dev.co"], capture_output=True, text=True).stdout


def price(m):
    if m in cost:
        return cost[m]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if base in cost:
        return cost[base]
    if m.endswith("_e64") or m.endswith("_dpp") or m.startswith("v_pk_") or "f64" in m or "64" in base.split("_")[-1]:
        return 4.13
    if any(t in m for t in ("rcp", "rsq", "sqrt", "exp", "log", "sin", "cos")):
        return 8.1
    if any(t in m for t in ("cmp", "cndmask", "readlane", "readfirstlane", "writelane", "mad", "mul_lo", "mul_hi", "cvt", "bfe", "perm", "alignb", "dot", "max", "min", "med3", "lshl", "ashr", "add3", "lshl_add", "and_or", "bcnt", "mbcnt", "sad")):
        return 4.13
    return 2.3


kern = None
mix = collections.defaultdict(collections.Counter)
for ln in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
    if m:
        kern = m.group(1)
        continue
    t = ln.split()
    if kern and len(t) >= 1 and t[0].startswith("v_"):
        mix[kern][t[0]] += 1
res = {}
for k, c in mix.items():
    n = sum(c.values())
    if n < 20:
        continue
    cyc = sum(price(m) * v for m, v in c.items())
    fast = sum(v for m, v in c.items() if price(m) < 3.0)
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    res[name] = {"static_valu_instructions": n, "cycles_per_valu_instruction": round(cyc / n, 3), "share_at_2_cycle_rate": round(fast / n, 3),
                 "top": [[m, v] for m, v in c.most_common(6)]}
json.dump({"_about": f"static VALU mix of libdrfe.so priced with profiles/{tag}_valu_issue.txt (8 waves per SIMD); tools/valu_mix.py",
           "kernels": res}, sys.stdout, indent=1)
