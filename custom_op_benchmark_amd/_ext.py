"""The compiled PyTorch-ROCm C++ extension `graphop_cpp` (csrc/torch_ext.cpp): the reference's
pybind11 boundary (graphop/graphop.cpp) re-created over the C ABI, plus TORCH_LIBRARY(graphop).

build()  compiles it in-tree with g++ against the installed torch headers (no GPU needed; ~30 s);
load()   imports it (running its PYBIND11_MODULE init and its static TORCH_LIBRARY registration)
         or returns None when it has not been built -- the ctypes binding then registers
         torch.ops.graphop.* itself (graphop.py)."""
import importlib.util
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "graphop_cpp.so")
SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
_mod = None


def build(force=False):
    import pybind11
    import sysconfig
    import torch
    lib = os.path.join(_HERE, "libgraphop_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("build libgraphop_hip.so first (make -C custom_op_benchmark_amd/csrc)")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "graphop_hip.h")
    if not force and os.path.exists(EXT_PATH) and os.path.getmtime(EXT_PATH) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        return EXT_PATH
    ti = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", EXT_PATH,
           "-DTORCH_EXTENSION_NAME=graphop_cpp", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ti, "include"), "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(os.path.dirname(_HERE), "include"),
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip",
           "-ltorch_hip", "-L" + _HERE, "-lgraphop_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ti, "lib"),
           "-Wno-deprecated-declarations"]
    subprocess.check_call(cmd)
    return EXT_PATH


def load():
    """-> the graphop_cpp module, or None if it is not built (or GRAPHOP_NO_CPP_EXT=1)."""
    global _mod
    if _mod is not None:
        return _mod
    if os.environ.get("GRAPHOP_NO_CPP_EXT", "0") == "1" or not os.path.exists(EXT_PATH):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("graphop_cpp", EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules.setdefault("graphop_cpp", mod)
    _mod = mod
    return mod
