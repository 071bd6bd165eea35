"""Builds / loads the torch C++ extension ``crender_torch`` (csrc/crender_torch.cpp): the
tensor-unwrapping front end of the C ABI that the filler uses on its per-frame path.

In-tree build (``__graft_entry__.build()`` -> ``build()`` here): plain C++ against torch's
headers and ``c10/hip/HIPStream.h``, linked to ``libcrender_hip.so`` next to it (rpath $ORIGIN);
no device code, so it needs no GPU and no hipcc.  ``load()`` imports the built module and never
builds; a missing module is an error on the paths that need it (no fallback compute path: the
ctypes binding calls the same C ABI and remains available for everything else)."""
from __future__ import annotations

import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
NAME = "crender_torch"
BUILD_DIR = os.path.join(_HERE, "_torch_ext")
SOURCE = os.path.join(_HERE, "csrc", "crender_torch.cpp")

_mod = None


def _built_path():
    hits = glob.glob(os.path.join(BUILD_DIR, NAME + "*.so"))
    return hits[0] if hits else None


def needs_build() -> bool:
    so = _built_path()
    if so is None:
        return True
    deps = [SOURCE, os.path.join(_HERE, "..", "include", "crender_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the extension into cython3dmodelrenderer_amd/_torch_ext/ (ninja + g++)."""
    from . import _build
    _build.build()                    # libcrender_hip.so must exist to link against
    if not force and not needs_build():
        return _built_path()
    from torch.utils import cpp_extension as ce
    os.makedirs(BUILD_DIR, exist_ok=True)
    rocm = ce.ROCM_HOME or "/opt/rocm"
    ce.load(name=NAME, sources=[SOURCE], build_directory=BUILD_DIR, verbose=verbose,
            extra_include_paths=[os.path.join(rocm, "include"), os.path.join(_HERE, "..", "include")],
            extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
            extra_ldflags=[f"-L{_HERE}", "-lcrender_hip", "-Wl,-rpath,'$$ORIGIN/..'",   # ($$: ninja, quotes: sh)
                           f"-L{os.path.join(os.path.dirname(__import__('torch').__file__), 'lib')}", "-lc10_hip", "-ltorch_hip"],
            with_cuda=False, is_python_module=False)
    return _built_path()


def load():
    """The built module (raises if it has not been built)."""
    global _mod
    if _mod is not None:
        return _mod
    so = _built_path()
    if so is None:
        raise ImportError(f"{NAME} is not built: run __graft_entry__.build() "
                          "(or python -m cython3dmodelrenderer_amd._torch_ext)")
    import torch  # noqa: F401  (libtorch must be loaded first)
    from . import _capi
    _capi.load()                      # libcrender_hip.so (also found through the extension's rpath)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.abi_version() != _capi.ABI_VERSION:
        raise ImportError(f"{NAME} is bound to ABI {mod.abi_version()}, this package speaks {_capi.ABI_VERSION}: rebuild")
    sys.modules.setdefault(NAME, mod)
    _mod = mod
    return mod


if __name__ == "__main__":
    print(build(force=True, verbose=True))
