"""On-disk formats of SURVEY.md §8f-4 (CPU: framing, header layout, multi-file results, ASCII -> HDF5)."""
import numpy as np
import pandas as pd
import pytest

from astrild_amd import formats


def test_density_header_layout_matches_the_file_format():
    dt = formats.DENSITY_HEADER_DTYPE
    assert dt.itemsize == 1024
    # offsets of the reference's record (particles/hutils/density.py:173-196)
    assert [dt.fields[k][1] for k in ("gridSize", "totalGrid", "fileType", "noDensityFiles", "densityFileGrid",
                                       "indexDensityFile", "box", "npartTotal", "mass", "time")] == \
        [0, 24, 32, 36, 40, 52, 56, 104, 152, 200]
    assert dt.fields["FILE_ID"][1] == 1016 and dt.fields["fill"][1] == 256


def test_density_grid_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    grid = rng.standard_normal((6, 5, 4)).astype(np.float32)
    path = str(tmp_path / "snap_012.a_den")
    formats.write_density_grid(path, grid, 250.0, redshift=0.5)
    raw = np.fromfile(path, np.uint8)
    assert raw.size == 8 + 1024 + 8 + 8 + grid.nbytes + 8
    assert np.frombuffer(raw[:8].tobytes(), np.uint64)[0] == 1024
    header, got = formats.read_density_grid(path, to_device=False)
    assert np.array_equal(got, grid) and got.shape == (6, 5, 4)            # x slowest, like dtfe.py:72
    assert header["BoxSize"] == 250.0 and header["redshift"] == 0.5 and int(header["totalGrid"]) == 120
    vel = rng.standard_normal((4, 4, 4, 3)).astype(np.float32)
    formats.write_density_grid(str(tmp_path / "v.a_vel"), vel, 100.0, file_type=11)
    assert np.array_equal(formats.read_density_grid(str(tmp_path / "v.a_vel"), to_device=False)[1], vel)
    # truncated payload, corrupted framing
    open(str(tmp_path / "bad1"), "wb").write(raw[:-20].tobytes())
    with pytest.raises(formats.DensityFileError):
        formats.read_density_grid(str(tmp_path / "bad1"), to_device=False)
    bad = raw.copy()
    bad[8 + 1024] ^= 1
    open(str(tmp_path / "bad2"), "wb").write(bad.tobytes())
    with pytest.raises(formats.DensityFileError):
        formats.read_density_grid(str(tmp_path / "bad2"), to_device=False)
    with pytest.raises(formats.DensityFileError):
        formats.read_density_header(str(tmp_path / "missing"))


def test_density_grid_split_over_several_files_is_refused(tmp_path):
    """The reference's reader puts every piece of a multi-file result at offset 0 (density.py:437): there is no
    behaviour to mirror, so such files are refused with a message that says so."""
    grid = np.arange(4 * 3 * 2, dtype=np.float32).reshape(4, 3, 2)
    root = str(tmp_path / "split.den")
    for i, part in enumerate((grid[:2], grid[2:])):
        formats.write_density_grid(f"{root}.{i}", part, 10.0, noDensityFiles=2, indexDensityFile=i)
    header, _ = formats.read_density_header(root)
    assert int(header["noDensityFiles"]) == 2
    with pytest.raises(formats.DensityFileError, match="split over 2 files"):
        formats.read_density_grid(root, to_device=False)


def test_compress_rayramses_outputs(tmp_path):
    fields = ["rayid", "chi_co", "the_co", "phi_co", "kappa_2", "shear_x", "shear_y"]
    rng = np.random.default_rng(3)
    table = pd.DataFrame(rng.uniform(0.1, 1.0, size=(40, len(fields))), columns=fields)
    table["rayid"] = rng.permutation(40)
    files = []
    for i, part in enumerate((table.iloc[:17], table.iloc[17:])):
        path = str(tmp_path / f"Ray_maps_00007.out{i:05d}")
        part.to_csv(path, sep=" ", header=False, index=False, float_format="%.17g")
        files.append(path)
    out = formats.compress_rayramses_outputs(files, fields)
    ref = table.sort_values("rayid").set_index("rayid")
    assert list(out.index) == list(range(40))
    np.testing.assert_allclose(out.values, ref.values, rtol=2e-15)      # pandas' default float parser: 1 ulp
    conv = formats.compress_rayramses_outputs(files, fields, convert=True, hubble_h=0.7)
    sy = ref["shear_y"] * 2.0 * np.sin(ref["the_co"])
    np.testing.assert_allclose(conv["chi_co"], ref["chi_co"] / 0.7, rtol=1e-15)
    np.testing.assert_allclose(conv["shear_x"], -ref["shear_x"] * np.cos(2 * ref["phi_co"]) - sy * np.sin(2 * ref["phi_co"]),
                               rtol=1e-14)
    np.testing.assert_allclose(conv["shear_y"], -ref["shear_x"] * np.sin(2 * ref["phi_co"]) + sy * np.sin(2 * ref["phi_co"]),
                               rtol=1e-14)
    with pytest.raises(ValueError):
        formats.compress_rayramses_outputs(files, fields, convert=True)
