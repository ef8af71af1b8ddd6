"""[needs the lab switches: git apply tools/lab/attic/lab_switches.patch first — tools/lab/README.md]
Lab: what the 8-phase tiles' main loop waits for on the step's big convolutions.  Builds csrc/gemm8.hip with
-DVN_GEMM8_LAB=<mask> (pieces of the loop removed; results are garbage, only durations mean anything) and times the
512^2 x 128 -> 128 conv under the 256x128 tile and the 256^2 x 256 -> 256 conv under the 256x256 tile.
    python tools/lab/gemm8_parts.py build          (in the container)
    python tools/lab/gemm8_parts.py                (on the GPU box; re-executes itself once per variant)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "view_neti_amd", "csrc")
VARIANTS = {0: "product", 1: "no MFMAs", 2: "no fragment reads", 4: "no staging DMAs", 8: "A gather folded into 256 KiB (L2 hits)",
            3: "no MFMAs, no fragment reads (staging only)", 6: "MFMAs only", 5: "fragment reads only", 9: "no MFMAs, A from L2",
            11: "staging only, A from L2", 7: "nothing (barriers, prologue, epilogue)"}
if os.environ.get("HALO"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in (0, 1, 2, 4, 3, 6, 5, 7)}
    # (round 3 also had switches that changed the synchronisation — 16 no vmcnt waits (a data race), 32 no s_setprio, 64 one
    #  phase per K-tile, 128 no stagger; their results are in DESIGN.md section 4 and the switches were removed in round 4)
if os.environ.get("ONLY"):
    VARIANTS = {int(k): VARIANTS.get(int(k), "?") for k in os.environ["ONLY"].split(",")}
so = lambda m: os.path.join(ROOT, "tools", "lab", f"libvneti_g8lab_{m}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "gemm8.o"]
    for m in VARIANTS:
        obj = f"/tmp/g8lab_{m}.o"
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-fno-fast-math",
                               "-Wno-pass-failed", f"-DVN_GEMM8_LAB={m}", "-c", os.path.join(CS, "gemm8.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so(m), obj, *objs])
        print("built", so(m))
    sys.exit(0)
if "VNETI_LIB_PATH" not in os.environ:
    for m, name in VARIANTS.items():
        r = subprocess.run([sys.executable, __file__], env=dict(os.environ, VNETI_LIB_PATH=so(m), VN_LAB_NAME=f"{m:2d} {name}"),
                           capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-800:], flush=True)
    sys.exit(0)
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(64 * 2 ** 20, dtype=torch.float32, device=dev)
out = []
CASES = [(4, 512, 512, 128, 128, 17), (4, 256, 256, 256, 256, 16), (4, 64, 64, 640, 640, 16)]
if os.environ.get("HALO"):  # the halo-patch form (tile 18, chunk-major K; random weights, the order does not matter here)
    hh = int(os.environ.get("HINT", "18"))  # 19: the 512x128 form
    CASES = [(4, 512, 512, 128, 128, hh), (4, 128, 128, 512, 512, hh), (4, 64, 64, 640, 640, hh)]
for (B, H, W, Ci, Co, hint) in CASES:
    x = torch.randn(B * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1 if hint >= 18 else 0)
    f = lambda: ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=hint, workspace=ws, split_k=1)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    out.append(f"{H}^2x{Ci}->{Co} tile {hint}: {s.elapsed_time(e) / 10 * 1e3:7.1f} us")
print(f"{os.environ.get('VN_LAB_NAME', 'product'):48s} " + " | ".join(out))
