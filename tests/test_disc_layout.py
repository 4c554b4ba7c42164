"""The DISC layout of the slab transpose (what FFTPower keeps of the half spectrum after the k_y pass,
power_spectrum_3d.py:189-195): the library's table against its Python restatement in astrild_amd/slab.py.
Host arithmetic only - no GPU."""
import ctypes as ct

import numpy as np
import pytest


@pytest.mark.parametrize("n", [256, 512, 1024])
@pytest.mark.parametrize("parts", [1, 2, 4, 8])
def test_library_table_equals_python_layout(hip, n, parts):
    from astrild_amd import slab
    from astrild_amd._lib import check
    r1 = 32 if n == 1024 else 16
    lay = slab.disc_layout(n, parts, r1, 16)
    S = (ct.c_uint * parts)()
    owner = (ct.c_ubyte * (n // r1))()
    r1_out = ct.c_int()
    check(hip.ast_fft_tile_disc_layout(n, parts, S, owner, ct.byref(r1_out)), "ast_fft_tile_disc_layout")
    assert r1_out.value == r1 and [int(v) for v in S] == lay["S"] and [int(v) for v in owner] == lay["part_of"]
    tab = (ct.c_int * (lay["tiles"] * lay["nblk"] * 4))()
    check(hip.ast_fft_tile_disc_table(n, parts, tab, len(tab)), "ast_fft_tile_disc_table")
    tab = np.frombuffer(tab, dtype=np.int32).reshape(lay["tiles"], lay["nblk"], 4)
    lohi = tab[:, :, 3].astype(np.int64)
    lo, hi, part = lohi & 255, (lohi >> 8) & 255, lohi >> 16
    assert np.array_equal(lo, lay["lo"]) and np.array_equal(hi, lay["hi"])
    assert np.array_equal(part, np.broadcast_to(np.array(lay["part_of"]), part.shape))
    live = hi > lo
    assert np.array_equal(tab[:, :, 0][live], lay["offb"][live])
    assert np.array_equal(tab[:, :, 1], np.array(lay["S"])[part]) and np.array_equal(tab[:, :, 2], np.array(lay["cumS"])[part])
    # what the layout is for: 21.5 % of the half plane gone, owners within a few per cent of each other
    full = n * ((n // 2 + 1 + 15) // 16 * 16)
    assert lay["total"] < 0.79 * full
    assert max(lay["S"]) <= (1.05 if n >= 512 else 1.12) * lay["total"] / parts      # (16 blocks on 8 parts: two each)


def test_layout_refuses_bad_geometry(hip):
    S = (ct.c_uint * 8)()
    assert hip.ast_fft_tile_disc_layout(1024, 3, S, None, None) != 0          # 32 row blocks do not deal out to 3 parts
    assert hip.ast_fft_tile_disc_layout(128, 2, S, None, None) != 0           # no hand-written passes at this size
    assert hip.ast_fft_tile_disc_layout(256, 32, S, None, None) != 0          # 16 row blocks
