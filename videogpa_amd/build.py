"""Build the C-ABI HIP library (libvgpa_hip.so) for gfx950, in-tree.

    python -m videogpa_amd.build            # incremental
    python -m videogpa_amd.build --force

hipcc cross-compiles without a GPU.  One object per csrc/*.hip (parallel), linked into
videogpa_amd/csrc/libvgpa_hip.so, which travels with the repo snapshot to the GPU box.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libvgpa_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    if force or _stale(obj, [src] + headers):
        # the MFMA kernels keep fp32 adds / multiplies unpacked: v_pk_*_f32 does not co-issue with the matrix pipe (attention.hip)
        extra = ["-fno-slp-vectorize"] if os.path.basename(src) in ("attention.hip", "attention_w1.hip", "attention_hd128.hip", "lora.hip") else []
        r = subprocess.run([HIPCC, *FLAGS, *extra, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
