"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pfst_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

from pfst_amd import _lib


def test_header_parses_and_library_exports_every_symbol():
    decls = _lib.parse_header()
    assert len(decls) >= 40
    assert os.path.exists(_lib.LIB_PATH), 'run python -m pfst_amd.build'
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in decls if not hasattr(L, n)]
    assert not missing, missing
    assert _lib.lib().pfst_abi_version() == 1


def test_argument_validation_without_gpu():
    # bad arguments are rejected on the host before any launch
    L = _lib.lib()
    assert L.pfst_fill_f32(None, 10, 0.0, None) == -1
    assert b'spatial.hip' in L.pfst_last_error()
    assert L.pfst_conv_igemm(None, 0, None, None, None, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, None, None, None) == -1
