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
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s"])
    return HOSTSIM


def lib_path(backend):
    if backend == "hip":
        return pyddp.library_path()
    return hostsim_path()


BACKENDS = [pytest.param("hostsim", id="hostsim"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]

# ---- kernel selection by environment variable: TEST PLUMBING.  Until round 4 libpddp itself read PDDP_BP / PDDP_FP / ... when a handle was created; the library now takes
# the selection as data (pddp_config.kernels, include/pddp.h) and reads no environment.  The comparison tests and the measurement scripts under tools/ still name a family
# by setting one of these variables around the creation of a handle; install_env_selection() makes pyddp.default_config translate them into the explicit record.
ENV_SELECTION = {"PDDP_BP": "bp", "PDDP_FP": "fp", "PDDP_SWEEP": "sweep", "PDDP_LS": "ls", "PDDP_AB": "ab", "PDDP_CF": "cf", "PDDP_CF_BP": "cf_bp", "PDDP_CF_FP": "cf_fp",
                 "PDDP_CF_NIS": "cf_nis"}


def selection_from_env():
    return {field: os.environ[var] for var, field in ENV_SELECTION.items() if os.environ.get(var)}


def install_env_selection():
    if getattr(pyddp.default_config, "_env_selection", False):
        return
    plain = pyddp.default_config

    def default_config(plant, _lib_path=None, kernels=None, **kw):
        sel = selection_from_env()
        sel.update(kernels or {})
        return plain(plant, _lib_path=_lib_path, kernels=sel, **kw)
    default_config._env_selection = True
    pyddp.default_config = default_config
    pyddp.binding.default_config = default_config


def make_solver(backend, plant, **kw):
    path = lib_path(backend)
    cfg = pyddp.default_config(plant, _lib_path=path, **kw)
    return pyddp.Solver(cfg, _lib_path=path)
