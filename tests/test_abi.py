"""CPU-only: the C-ABI library loads and exports exactly what include/astrild_hip.h
declares, and the ctypes table in astrild_amd/_lib.py covers every declaration.
No compute calls (there is no GPU here)."""
import ctypes as ct
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "astrild_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_\s\*]*?\b([a-z][A-Za-z0-9_]*)\s*\(", src, flags=re.M)
    return sorted({n for n in names if n.startswith("ast_") or n.startswith("kappa0_")})


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for required in ("ast_paint", "ast_paint_tiled", "ast_ngp_assign", "ast_fft_exec", "ast_power_bin_1d",
                     "ast_kappa_stack", "ast_kappa_to_alphas", "kappa0_to_alphas", "kappa0_to_phi",
                     "ast_gaussian_smooth", "ast_histogram", "ast_shell_filter", "ast_slab_pack"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from astrild_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    handle = ct.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} is declared in the header but not exported"


def test_ctypes_table_matches_header():
    from astrild_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.lib()                      # sets argtypes/restype for all of them
    assert lib.ast_version() >= 100       # host-only call, no GPU needed


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "astrild_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


def test_missing_library_fails_loudly(monkeypatch):
    from astrild_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libastrild_hip.so")
    with pytest.raises(_lib.AstrildHipError):
        _lib.lib()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from astrild_amd import device, _lib
    with pytest.raises(_lib.AstrildHipError):
        device.device()


def test_order_probe_sample_stays_inside_the_particle_array():
    """device.sample_is_unordered reads 32 particles from each start: every start is a multiple of 32 and leaves room for
    its run at every particle count - in particular at 2^30, where a float32 linspace of the starts rounds past the end -
    and on CPU tensors the probe tells lattice order from shuffled order."""
    import numpy as np
    import torch
    from astrild_amd import device as dev
    for npart in (32, 33, 64, 65, 1000, 1 << 20, (1 << 30) - 1, 1 << 30, (1 << 30) + 31, (1 << 32) - 66):
        st = dev.sample_run_starts(npart)
        assert len(st) == 256 and st[0] == 0 and all(s % 32 == 0 or s == npart - 32 for s in st)
        assert all(0 <= s <= npart - 32 for s in st) and st == sorted(st)
    n, L = 64, 100.0
    ijk = np.stack(np.meshgrid(*(np.arange(n),) * 3, indexing="ij"), axis=-1).reshape(-1, 3)
    rng = np.random.default_rng(3)
    nat = torch.from_numpy((((ijk + 0.5 + 0.5 * rng.standard_normal(ijk.shape)) * (L / n)) % L).astype(np.float32))
    assert not dev.sample_is_unordered(nat, n, L)
    assert dev.sample_is_unordered(nat[torch.from_numpy(rng.permutation(len(nat)))].contiguous(), n, L)
