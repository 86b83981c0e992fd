"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pfst_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

import pytest

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


def test_f16x3_chain_grid_policy():
    """pfst_f16x3_chain_grid (host logic, no GPU): every tile is covered by chains of at most 8, never fewer workgroups than resident slots
    when there are that many tiles, no chain for launches that cannot chain, and the last round of workgroups is never mostly empty when a
    better split exists (9216 tiles on 512 slots: 1536 workgroups x 6 = three full rounds, not 1152 x 8 = two and a quarter)."""
    from pfst_amd._lib import lib
    L = lib()
    assert L.pfst_f16x3_set_slots(512) == 0
    try:
        for total in (1, 100, 512, 513, 600, 2304, 4096, 9216, 16384, 100000):
            g = L.pfst_f16x3_chain_grid(total, 1)
            assert 0 < g <= total and -(-total // g) <= 8, (total, g)
            assert g == total if total <= 512 else g >= 512, (total, g)
            assert L.pfst_f16x3_chain_grid(total, 0) == total
        assert L.pfst_f16x3_chain_grid(9216, 1) == 1536 and L.pfst_f16x3_chain_grid(16384, 1) == 2048
        assert L.pfst_f16x3_chain_grid(0, 1) == 0
    finally:
        L.pfst_f16x3_set_slots(0)


def test_call_refuses_a_wrong_argument_count():
    """ctypes lets surplus arguments through on cdecl functions; _lib.call does not (an appended argument at the wrong call site would move
    the stream out of its slot and the launch onto the default stream)"""
    from pfst_amd import _lib
    _lib.lib()
    n = len(_lib._decls['pfst_fill_f32'][1])
    with pytest.raises(TypeError, match='takes'):
        _lib.call('pfst_fill_f32', *([0] * (n + 1)))
    with pytest.raises(TypeError, match='takes'):
        _lib.call('pfst_fill_f32', *([0] * (n - 1)))
