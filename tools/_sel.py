"""Measurement-script plumbing (tools/ only): name a kernel family on the command line of a probe as PDDP_BP=mx PDDP_FP=tl ... -- install() makes pyddp.default_config
translate those variables into the explicit record pddp_config.kernels.  Neither the library nor the test suite reads the environment for its kernel selection; this keeps
the round-2..5 one-liners under tools/ working."""
import os

import pyddp

ENV_SELECTION = {"PDDP_BP": "bp", "PDDP_FP": "fp", "PDDP_SWEEP": "sweep", "PDDP_LS": "ls", "PDDP_AB": "ab", "PDDP_CF": "cf", "PDDP_CF_BP": "cf_bp", "PDDP_CF_FP": "cf_fp", "PDDP_CF_NIS": "cf_nis"}


def install():
    if getattr(pyddp.default_config, "_env_selection", False):
        return
    plain = pyddp.default_config

    def default_config(plant, _lib_path=None, kernels=None, **kw):
        sel = {field: os.environ[var] for var, field in ENV_SELECTION.items() if os.environ.get(var)}
        sel.update(kernels or {})
        return plain(plant, _lib_path=_lib_path, kernels=sel, **kw)
    default_config._env_selection = True
    pyddp.default_config = default_config
    pyddp.binding.default_config = default_config
