"""Register / spill / scratch budget of every kernel in lib/libmocap_core.so (llvm-objdump --offloading + llvm-readelf --notes);
no GPU needed.  usage: python scripts/kernel_budget.py [substring ...]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("MOCAP_CORE_LIB") or os.path.join(ROOT, "low-cost-mocap_amd", "lib", "libmocap_core.so")
LLVM = "/opt/rocm/lib/llvm/bin"
d = tempfile.mkdtemp()
shutil.copy(LIB, os.path.join(d, "lib.so"))
subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
for f in sorted(x for x in os.listdir(d) if "amdgcn" in x):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=d, check=True, capture_output=True, text=True).stdout
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if sys.argv[1:] and not any(a in dem for a in sys.argv[1:]):
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
        print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):3d} spill {g('vgpr_spill_count'):3d} sgpr_spill {g('sgpr_spill_count'):3d} scratch {g('private_segment_fixed_size'):4d}")
shutil.rmtree(d)
