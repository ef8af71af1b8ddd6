"""Build libvneti_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Usage: python view_neti_amd/csrc/build.py [--force]
Objects are rebuilt only when their source (or a header) is newer.  The .so stays in-tree
(git-ignored) so that it travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libvneti_hip.so")
# the same sources with -DVN_BF16 (common.h: half_t = __bf16, bf16 MFMA opcodes): the reference's mixed_precision=bf16 branch
SO_BF16 = os.path.join(HERE, "libvneti_hip_bf16.so")
ARCH = "gfx950"
# kernarg preload: the first 14 dwords of a kernel's explicit scalar / pointer parameters arrive in SGPRs with the wave
# instead of through an s_load round trip (the GEMM kernels list their prologue's operands that way; by-value structs
# are unaffected)
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
         "-ffast-math" if False else "-fno-fast-math", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "vneti.h"))
    return hs


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, bf16=False):
    obj = os.path.join(HERE, "build_bf16" if bf16 else "build", os.path.basename(src)[:-4] + ".o")
    if force or _stale(obj, [src] + _headers()):
        cmd = ["hipcc", *FLAGS, *(["-DVN_BF16"] if bf16 else []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    """both libraries (fp16: libvneti_hip.so, returned; bf16: libvneti_hip_bf16.so), all objects compiled in parallel"""
    for d in ("build", "build_bf16"):
        os.makedirs(os.path.join(HERE, d), exist_ok=True)
    srcs = sources()
    jobs = [(s, False) for s in srcs] + [(s, True) for s in srcs]
    # the big translation units first (gemm_conv.hip alone takes about a minute)
    jobs.sort(key=lambda j: -os.path.getsize(j[0]))
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
        objs = dict(zip(jobs, ex.map(lambda j: _compile(j[0], force, j[1]), jobs)))
    for so, bf in ((SO, False), (SO_BF16, True)):
        mine = [objs[(s, bf)] for s in srcs]
        if force or _stale(so, mine):
            cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", so, *mine]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
            if verbose:
                print(f"built {so}")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
