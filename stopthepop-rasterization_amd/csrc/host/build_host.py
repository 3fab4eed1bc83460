#!/usr/bin/env python3
"""Builds diff_gaussian_rasterization/_stp_host<EXT_SUFFIX>: the native torch binding of libstp_raster.so (stp_torch_binding.cpp).

A torch CppExtension in everything but the driver: ONE g++ command with torch's own include / library paths and ABI flag
(torch.utils.cpp_extension reports them), no .hip / .cu source and therefore no hipify pass; the result lands IN-TREE next
to _C.py so that it travels with the repository snapshot (a JIT cache under ~/.cache would not).  Host code only: it
compiles on a box without a GPU.  pybind11 comes from torch's own bundled headers (ce.include_paths()).   usage: python build_host.py [-v]"""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.normpath(os.path.join(HERE, "..", "..", "diff_gaussian_rasterization"))
NAME = "_stp_host"
SRC = os.path.join(HERE, "stp_torch_binding.cpp")
OUT = os.path.join(PKG, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def command():
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [SRC, "-o", OUT, f"-L{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64", "-ldl",
            f"-Wl,-rpath,{torch_lib}"]
    return cmd


def up_to_date():
    return os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(HERE, "..", "..", "..", "include", "stp_raster.h")),
                                                                os.path.getmtime(__file__))


def main():
    if up_to_date() and "-f" not in sys.argv:
        return
    cmd = command()
    if "-v" in sys.argv:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


if __name__ == "__main__":
    main()
