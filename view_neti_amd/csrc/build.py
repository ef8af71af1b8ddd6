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
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
         "-ffast-math" if False else "-fno-fast-math"]


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


def _compile(src, force):
    obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
    if force or _stale(obj, [src] + _headers()):
        cmd = ["hipcc", *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(SO, objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", SO, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {SO}")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
