#!/usr/bin/env python3
"""One instantiation of decode_fused_kernel as a translation unit of its own (seconds instead of minutes per compile):
tools/one_kernel.py "false, 8, 32, 64, false, false, 1, true, 0, 16, 1, 16" [--asm OUT.s] [--src FILE.hip] [-D...]
Prints the resource line; with --asm also writes the assembly (for tools/isa_scratch_map.py / isa_phase_hist.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
targs = args.pop(0)
asm = None; srcf = os.path.join(ROOT, "slimt_amd/csrc/decode_fused.hip"); extra = []
while args:
    a = args.pop(0)
    if a == "--asm": asm = args.pop(0)
    elif a == "--src": srcf = args.pop(0)
    else: extra.append(a)
s = open(srcf).read()
cut = s.index("template <bool MG, int KSD, int KSF, int DH>\nstatic auto decode_fused_pick")
s = s[:cut] + "template __global__ void decode_fused_kernel<%s>(FusedDecodeArgs);\n}  // namespace slimt_hip\n" % targs
tmp = os.path.join(ROOT, "slimt_amd/csrc/_one_kernel.hip")
open(tmp, "w").write(s)
try:
    env = dict(os.environ, SLIMT_HIPCC_EXTRA=" ".join(extra))
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools/kernel_resources.py"), tmp], capture_output=True, text=True, env=env).stdout, end="")
    if asm:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", "-x", "hip",
                        tmp, "-I", os.path.join(ROOT, "slimt_amd/csrc"), "-I", os.path.join(ROOT, "include"), "-o", asm] + extra, stderr=subprocess.DEVNULL, check=True)
finally:
    os.remove(tmp)
