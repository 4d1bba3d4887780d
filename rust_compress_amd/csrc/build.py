"""Builds librcx.so (the C-ABI + gfx950 kernels) in-tree with hipcc.  Cross-compiles without a GPU.

The library is five translation units (rcx_api + one per codec family) compiled in parallel and linked once;
objects are cached under csrc/build/ and rebuilt when a source they include is newer.
`ab=True` (or RCX_AB=1 in the environment) builds librcx_ab.so with -DRCX_AB_VARIANTS: the earlier kernel
generations, profiling instantiations and experiments that benchmarks/ compares against.  The shipped librcx.so
holds the default kernel and one fallback per codec only."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
TUS = ["rcx_api", "tu_lz4", "tu_inflate", "tu_bwt", "tu_serial"]


def _out(ab):
    return os.path.join(HERE, "librcx_ab.so" if ab else "librcx.so")


def build(force=False, verbose=False, ab=None):
    if ab is None:
        ab = bool(os.environ.get("RCX_AB"))
    out = _out(ab)
    deps = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "..", "include", "rcx.h"))
    if ab:                                        # the experiment kernels live outside the product tree
        exp = os.path.join(HERE, "..", "..", "benchmarks", "experiments")
        deps += [os.path.join(exp, f) for f in sorted(os.listdir(exp)) if f.endswith(".hip")]
    newest = max(os.path.getmtime(d) for d in deps)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
    if ab:
        flags.append("-DRCX_AB_VARIANTS")
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    flags += os.environ.get("RCX_EXTRA_FLAGS", "").split()        # A/B experiments with compiler options
    # objects are cached per flag list: a rebuild with other RCX_EXTRA_FLAGS must not link objects compiled with the old ones
    import hashlib
    tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8]
    objdir = os.path.join(HERE, "build", ("ab" if ab else "ship") + "-" + tag)
    os.makedirs(objdir, exist_ok=True)
    stamp = out + ".flags"
    if not force and os.path.exists(out) and (not os.path.exists(stamp) or open(stamp).read() != tag):
        force_link = True
    else:
        force_link = False

    if not force and not force_link and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out

    def compile_tu(tu):
        obj = os.path.join(objdir, tu + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        subprocess.check_call([HIPCC] + flags + ["-c", os.path.join(HERE, tu + ".hip"), "-o", obj])
        return obj

    with ThreadPoolExecutor(len(TUS)) as ex:
        objs = list(ex.map(compile_tu, TUS))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,-rpath,/opt/rocm/lib", "-o", out] + objs)
    with open(stamp, "w") as fh:
        fh.write(tag)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv, ab="--ab" in sys.argv or None))
