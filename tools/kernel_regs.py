"""VGPR / spill table of the kernels in a built libmckernels.so:  python tools/kernel_regs.py [lib] [substring]"""
import os, re, subprocess, sys, tempfile, shutil
lib = sys.argv[1] if len(sys.argv) > 1 else "cuda-convnet_amd/libmckernels.so"
sub = sys.argv[2] if len(sys.argv) > 2 else "mc_igemm"
LLVM = "/opt/rocm/lib/llvm/bin/"
tmp = tempfile.mkdtemp()
shutil.copy(lib, os.path.join(tmp, "l.so"))
subprocess.run([LLVM + "llvm-objdump", "--offloading", "l.so"], cwd=tmp, capture_output=True)
rows = []
for f in sorted(os.listdir(tmp)):
    if "gfx950" not in f: continue
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
        if not m: continue
        if m.group(1) in cur:
            rows.append(cur); cur = {}
        cur[m.group(1)] = m.group(2)
    if cur: rows.append(cur)
shutil.rmtree(tmp)
names = subprocess.run(["c++filt"], input="\n".join(r.get("name", "?") for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    n = n.replace("mc::(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if sub in n:
        print("%-64s vgpr %4s vspill %3s sspill %3s" % (n, r.get("vgpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count")))
