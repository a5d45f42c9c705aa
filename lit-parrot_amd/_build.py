"""Compile the HIP sources of this package into ``libparrot_hip.so`` (gfx950 only, in-tree).

``hipcc`` cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
"""
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libparrot_hip.so"
SOURCES = ["core.hip", "w4.hip", "dense.hip", "w8.hip", "norm.hip", "attn.hip", "attn_prefill.hip", "misc.hip", "engine.hip", "gemm.hip", "gemm2.hip", "gptq.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    newest = max(p.stat().st_mtime for p in list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) +
                 [PKG_DIR.parent / "include" / "parrot_hip.h"])
    return newest > LIB_PATH.stat().st_mtime


def _compile_one(args):
    exe, src, obj, verbose, diag = args
    cmd = [exe, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
    if diag:  # diagnostic build: environment A/B switches and the stamp / tuning hooks of tools/ (never the shipped library)
        cmd.insert(1, "-DPARROT_DIAG")
    for d in os.environ.get("PARROT_BUILD_DEFINES", "").split():  # experiment builds of tools/ab_*.sh: e.g. ENG_THIN_PIECES=8
        cmd.insert(1, "-D" + d)
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    return src, proc.returncode, proc.stdout + proc.stderr


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> Path:
    """Build libparrot_hip.so if it is missing or older than its sources: one hipcc -c per source file, in parallel, then a link.
    ``diag`` compiles with -DPARROT_DIAG (A/B switches from the environment, stamp hooks) for the measurement tools."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor

    exe = _hipcc()
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    jobs = [(exe, str(src), str(objdir / (src.stem + ".o")), verbose, diag) for src in srcs]
    workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        results = list(pool.map(_compile_one, jobs))
    failed = [(src, out) for src, rc, out in results if rc != 0]
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {src}\n{out}" for src, out in failed))
    tmp = LIB_PATH.with_suffix(".so.tmp")
    cmd = [exe, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", str(tmp)] + [j[2] for j in jobs]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys

    print(build(force=True, verbose=True, diag="--diag" in sys.argv))
