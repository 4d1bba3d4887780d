"""Builds tests/wavesim/build/libwavesim_kernels.so (g++ only; TEST INFRASTRUCTURE)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "build", "libwavesim_kernels.so")


def build(force=False, sanitize=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(HERE, "sim_kernels.cpp"), os.path.join(HERE, "wavesim.cpp")]
    deps = srcs + [os.path.join(HERE, "wavesim.h")]
    csrc = os.path.join(HERE, "..", "..", "rust_compress_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-std=c++17", "-O1", "-g"] + (["-DRCX_SIM_TRACE"] if os.environ.get("RCX_SIM_TRACE") else []) + \
          (["-fsanitize=address", "-fno-omit-frame-pointer"] if (sanitize or os.environ.get("RCX_SIM_ASAN")) else []) + [ "-fPIC", "-shared", "-x", "c++", "-include", os.path.join(HERE, "wavesim.h"),
           "-Wall", "-DRCX_V8_WHY_STATS", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-variable", "-Wno-attributes",
           "-o", OUT] + srcs
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
