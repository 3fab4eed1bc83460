"""pip-installable form of the drop-in package (the reference is `pip install`-able too, reference setup.py).

    pip install --no-build-isolation -e .        # or: pip install --no-build-isolation .

builds libstp_raster.so (and the second library, libstp_raster_fma.so: INTEGRATION.md section 5) with hipcc for gfx950
(make -C stopthepop-rasterization_amd/csrc [FMA_DEPTH=1]) and the native torch binding
_stp_host with g++ (csrc/host/build_host.py: host code only -- no kernel goes through a torch extension, hence no hipify
pass) and installs the package `diff_gaussian_rasterization` with both as package data.
The repository's test-side directories (oracle/, tests/, tools/) are not installed."""
import os
import subprocess
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_PARENT = "stopthepop-rasterization_amd"


class BuildWithLibrary(build_py):
    def run(self):
        jobs = str(min(8, os.cpu_count() or 1))
        subprocess.check_call(["make", "-C", os.path.join(ROOT, PKG_PARENT, "csrc"), "-j", jobs, "ARCH=gfx950"])
        # the second library is optional (STP_RASTER_LIB=fma / _C.use_library("fma") select it): a failure to build it must not keep the default from installing
        if subprocess.call(["make", "-C", os.path.join(ROOT, PKG_PARENT, "csrc"), "-j", jobs, "ARCH=gfx950", "FMA_DEPTH=1"]) != 0:
            print("warning: libstp_raster_fma.so (the fma-depth build) did not build; installing without it", file=sys.stderr)
        subprocess.check_call([sys.executable, os.path.join(ROOT, PKG_PARENT, "csrc", "host", "build_host.py")])
        super().run()


setup(
    name="diff_gaussian_rasterization",
    version="0.4.0",
    description="MI355X-native sorted Gaussian-splat rasterizer behind the StopThePop diff_gaussian_rasterization API",
    packages=["diff_gaussian_rasterization"],
    package_dir={"": PKG_PARENT},
    package_data={"diff_gaussian_rasterization": ["libstp_raster.so", "libstp_raster_fma.so", "_stp_host*.so"]},
    cmdclass={"build_py": BuildWithLibrary},
    python_requires=">=3.9",
    install_requires=[],   # torch (ROCm build) is expected in the environment, as with the reference
)
