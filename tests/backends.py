"""Backends the parity tests run against.

  "hip"     : the product, parallel-ddp_amd/lib/libpddp.so through its C ABI on a real MI355X  (-m gpu)
  "hostsim" : TEST TOOL -- the same kernel bodies compiled for the host with a 1-lane wave
              (tests/hostsim), used by the CPU suite to check kernel arithmetic and indexing without a GPU.
"""
import os
import subprocess

import pytest

import pyddp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM = os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim.so")


def hostsim_path():
    if os.environ.get("PDDP_HOSTSIM_SAN") == "1":          # the AddressSanitizer / UBSan build of the test tool (tools/sanitizers.sh; the suite then runs with libasan preloaded)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s", "-j8", "SAN=1"])
        return os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim_san.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s", "-j8"])
    return HOSTSIM


def lib_path(backend):
    if backend == "hip":
        return pyddp.library_path()
    return hostsim_path()


BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]

# Kernel families are pinned per handle through pddp_config.kernels: make_solver(..., kernels=dict(bp="mx", fp="tl")).  Neither the library nor this suite reads or
# writes the process environment for it.


def make_solver(backend, plant, **kw):
    path = lib_path(backend)
    cfg = pyddp.default_config(plant, _lib_path=path, **kw)
    return pyddp.Solver(cfg, _lib_path=path)
