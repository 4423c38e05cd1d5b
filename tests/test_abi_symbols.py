"""The C-ABI shared library loads in a GPU-less container and exports every symbol that
include/lrvb_hip.h declares; the ctypes table binds exactly that set (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'lrvb_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = re.findall(r'\b(lrvb_[a-z_0-9]+)\s*\(', text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    import lrvb_amd
    lib_path = lrvb_amd._hip.LIB_PATH
    assert os.path.exists(lib_path), 'liblrvb_hip.so missing: run __graft_entry__.build()'
    lib = ctypes.CDLL(lib_path)
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), 'symbol {} declared in include/lrvb_hip.h but not exported'.format(name)
    bound = set(lrvb_amd._hip._SIGNATURES) | {'lrvb_last_error'}
    assert set(declared) == bound, (set(declared) ^ bound)


def test_nothing_but_the_header_is_exported():
    """`nm -D`: the dynamic symbol table of the library defines exactly the header's functions -- no kernel stubs, no
    C++ helpers, no undeclared entry points (timing-lab code is not part of the product)."""
    import subprocess
    import lrvb_amd
    out = subprocess.run(['nm', '-D', '--defined-only', lrvb_amd._hip.LIB_PATH], capture_output=True, text=True, check=True)
    exported = sorted({ln.split()[-1].split('@')[0] for ln in out.stdout.splitlines() if ln.strip()})
    assert exported == _declared(), set(exported) ^ set(_declared())


def test_library_reports_version_and_fails_loudly_without_gpu():
    import lrvb_amd
    lib = lrvb_amd._hip.load()
    assert lib.lrvb_version() == 1
    if lrvb_amd._hip.device_count() == 0:
        import numpy as np
        import pytest
        par = lrvb_amd.VectorParam('x', 2)
        with pytest.raises(RuntimeError):
            lrvb_amd.QuadraticObjective(par, A=np.ones(2))       # no silent CPU fallback


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    """Without the built HIP library every device object refuses to construct: there is no CPU route."""
    import numpy as np
    import pytest
    import lrvb_amd
    monkeypatch.setattr(lrvb_amd._hip, '_lib', None)
    monkeypatch.setattr(lrvb_amd._hip, 'LIB_PATH', str(tmp_path / 'liblrvb_hip.so'))
    par = lrvb_amd.VectorParam('x', 2)
    with pytest.raises(OSError, match='no CPU fallback'):
        lrvb_amd.QuadraticObjective(par, A=np.ones(2))
    with pytest.raises(OSError, match='no CPU fallback'):
        lrvb_amd.DeviceContext([dict(kind=0, free_size=2, vec_size=2, dim0=2, dim1=0, lb=-np.inf, ub=np.inf)], quad_kind=1)
    # host-only pieces (packing, exponential families) do not need it
    par.set_free(np.array([0.5, -0.5]))
    assert np.allclose(par.get(), [0.5, -0.5])
