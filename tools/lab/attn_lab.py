"""[needs the lab switches: git apply tools/lab/attic/lab_switches.patch first — tools/lab/README.md]
Lab: how the MFMA, VALU, LDS and staging parts of attn_fwd_kernel<40> add up (B=4, H=8, N=4096).  Builds
csrc/attention.hip with -DVN_ATTN_LAB=<mask> into tools/lab/libvneti_attnlab_<mask>.so (results of those builds are garbage;
only their duration means anything) and times the forward launch of each.
    python tools/lab/attn_lab.py build          (in the container)
    python tools/lab/attn_lab.py                (on the GPU box; re-executes itself once per variant)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "view_neti_amd", "csrc")
VARIANTS = {0: "product", 1: "no softmax math (fma, exp)", 33: "no softmax math, no max pass", 2: "no MFMAs", 4: "no fragment reads",
            8: "no staging", 16: "no barrier", 24: "no staging, no barrier", 6: "no MFMAs, no fragment reads (VALU + staging)",
            30: "VALU only", 29: "MFMA only (no softmax math, reads, staging, barrier)", 61: "MFMA only, no max pass",
            25: "MFMA + fragment reads", 3: "reads + staging only"}
so = lambda m: os.path.join(ROOT, "tools", "lab", f"libvneti_attnlab_{m}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "attention.o"]
    for m in VARIANTS:
        obj = f"/tmp/attn_lab_{m}.o"
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-fno-fast-math",
                               "-Wno-pass-failed", f"-DVN_ATTN_LAB={m}", "-c", os.path.join(CS, "attention.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so(m), obj, *objs])
        print("built", so(m))
    sys.exit(0)
if "VNETI_LIB_PATH" not in os.environ:
    for m, name in VARIANTS.items():
        r = subprocess.run([sys.executable, __file__], env=dict(os.environ, VNETI_LIB_PATH=so(m), VN_LAB_NAME=f"{m:2d} {name}"),
                           capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-800:])
    sys.exit(0)
import torch
from view_neti_amd import ops
B, H, N, D = 4, 8, 4096, 40
dev = "cuda"
q, k, v = (torch.randn(B * N, H * D, device=dev).half() for _ in range(3))
o = torch.zeros_like(q); lse = torch.zeros(B, H, N, device=dev)
f = lambda: ops.attn_fwd(q, k, v, o, lse, B, H, N, N, D, D ** -0.5, False)
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f()
e.record(); torch.cuda.synchronize()
print(f"{os.environ.get('VN_LAB_NAME', 'product'):60s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
