"""Builds librcx.so (the C-ABI + gfx950 kernels) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "librcx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False, verbose=False):
    srcs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(HERE, "..", "..", "include", "rcx.h"))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-fno-gpu-rdc",
           "-Wno-unused-result", "-Wl,-rpath,/opt/rocm/lib", "-o", OUT, os.path.join(HERE, "rcx_api.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    cmd[1:1] = os.environ.get("RCX_EXTRA_FLAGS", "").split()        # A/B experiments with compiler options
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
