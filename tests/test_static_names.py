"""CPU: every global name a function of the product (and bench.py) loads must exist.

The HIP path cannot run in the GPU-less build container, so a typo in a rarely taken branch
(an undefined local that Python compiles as a global load) would otherwise surface only on the
MI355X box.  This walks the bytecode of every function and checks LOAD_GLOBAL / LOAD_NAME targets
against the module's globals and the builtins -- a poor man's pyflakes (no linter in the image)."""

import builtins
import dis
import importlib
import importlib.util
import os
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODULES = [
    "momlevel_amd", "momlevel_amd._lib", "momlevel_amd.core", "momlevel_amd.engine",
    "momlevel_amd.parallel", "momlevel_amd.steric", "momlevel_amd.reference",
    "momlevel_amd.derived", "momlevel_amd.util", "momlevel_amd.dynamic", "momlevel_amd.labeled",
    "momlevel_amd.adapters", "momlevel_amd.synthetic", "momlevel_amd.test_data",
    "momlevel_amd.cftime_lite", "momlevel_amd.eos._dispatch", "momlevel_amd.eos.wright",
    "momlevel_amd.eos.linear", "momlevel_amd.csrc.build", "oracle.momlevel_numpy",
    "oracle.wright_c", "oracle.cpu_worker", "oracle.host_abi",
]
FILES = ["bench.py", "__graft_entry__.py"]


def _code_objects(code):
    yield code
    for const in code.co_consts:
        if isinstance(const, types.CodeType):
            yield from _code_objects(const)


def _undefined(module):
    bad = []
    for obj in list(vars(module).values()):
        funcs = []
        if isinstance(obj, types.FunctionType) and obj.__module__ == module.__name__:
            funcs.append(obj)
        elif isinstance(obj, type) and obj.__module__ == module.__name__:
            for m in vars(obj).values():
                f = getattr(m, "__func__", m)
                f = getattr(f, "fget", f) if isinstance(m, property) else f
                if isinstance(f, types.FunctionType):
                    funcs.append(f)
        inner = [f.__wrapped__ for f in funcs if isinstance(getattr(f, "__wrapped__", None),
                                                            types.FunctionType)]
        for f in funcs + inner:
            known = set(f.__globals__) | set(vars(builtins))  # a decorator's wrapper lives elsewhere
            for code in _code_objects(f.__code__):
                for ins in dis.get_instructions(code):
                    if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in known:
                        bad.append(f"{module.__name__}.{f.__qualname__}: {ins.argval} "
                                   f"(line {code.co_firstlineno}+)")
    return bad


@pytest.mark.parametrize("name", MODULES)
def test_no_undefined_globals_in_module(name):
    assert _undefined(importlib.import_module(name)) == []


@pytest.mark.parametrize("rel", FILES)
def test_no_undefined_globals_in_script(rel):
    spec = importlib.util.spec_from_file_location("_static_" + rel.replace(".", "_"),
                                                  os.path.join(ROOT, rel))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    assert _undefined(module) == []


def test_exec_masking_asm_declares_its_scalar_side_effects():
    """s_and_saveexec_b64 writes SCC besides EXEC.  An inline-asm block that uses it must list
    "scc" (and "vcc", which carries the lane mask) among its clobbers, or the compiler may keep a
    scalar compare's result live across the block: round 4 found exactly that -- an `s_cmp ...
    s_cselect` pair scheduled around the predicated accumulate of the float32 one-pass kernel, wrong
    sums, no crash.  Checked on the source: every asm statement with `saveexec` names both."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = 0
    for rel in ("momlevel_amd/csrc/eos_device.hpp", "momlevel_amd/csrc/momlevel_hip.hip",
                "momlevel_amd/csrc/momlevel_promote.hip", "momlevel_amd/csrc/eos_promote.hpp"):
        text = open(os.path.join(root, rel)).read()
        for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(", text):
            depth, i = 1, m.end()
            while depth and i < len(text):
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            stmt = text[m.start():i]
            if "saveexec" in stmt:
                found += 1
                assert '"scc"' in stmt and '"vcc"' in stmt, f"{rel}: {stmt[:120]}"
    assert found >= 2  # add_skipna and accumulate
