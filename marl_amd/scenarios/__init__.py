"""Scenario loader (reference: multiagent/scenarios/__init__.py:5-7 loads a file by path with
imp.load_source; here built-in names resolve to modules of this package and any other path is
loaded from the file)."""
import importlib
import importlib.util
import os.path as osp


def load(name):
    base = name[:-3] if name.endswith('.py') else name
    if osp.sep not in base and '/' not in base:
        from . import hierarchy
        if base in hierarchy.PRESETS:       # the fork's L1F*/S1L* files: one parametric scenario
            return hierarchy.bound_module(base)
        try:
            return importlib.import_module('marl_amd.scenarios.' + base)
        except ModuleNotFoundError:
            pass
    pathname = name if osp.isabs(name) else osp.join(osp.dirname(__file__), name)
    spec = importlib.util.spec_from_file_location('marl_amd_user_scenario', pathname)
    if spec is None or not osp.exists(pathname):
        raise FileNotFoundError('scenario %r not found' % name)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
