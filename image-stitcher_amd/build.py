"""Build libisx_hip.so (gfx950 only) in-tree with hipcc.

    python image-stitcher_amd/build.py [--force]

The shared object lands in image-stitcher_amd/image_stitcher_amd/lib/ so that it travels to the GPU
box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "image_stitcher_amd", "lib")
OUT = os.path.join(OUT_DIR, "libisx_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "isx.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def _stale(obj, src, force):
    if force or not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.abspath(__file__), os.path.join(HERE, "..", "include", "isx.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """One object per source (compiled in parallel, only when stale), then one link."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(HERE, "build", "obj")
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]
    jobs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        jobs.append((src, obj, _stale(obj, src, force)))

    def compile_one(job):
        src, obj, stale = job
        if stale:
            cmd = [HIPCC] + compile_flags + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as pool:
        objs = list(pool.map(compile_one, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
