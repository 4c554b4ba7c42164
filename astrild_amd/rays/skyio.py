"""``SkyIO`` (src/astrild/rays/skyio.py): ray vector -> square map, file names."""
import numpy as np


class SkyIO:
    @staticmethod
    def transform_RayRamsesOutput_to_NumpyNdarray(values: np.ndarray) -> np.ndarray:
        """skyio.py:32-48: the zip/sort key is ``arange`` so the result is the
        row-major reshape of the first npix^2 values."""
        values = np.asarray(values)
        _npix = int(np.sqrt(len(values)))
        return np.array(values[: _npix * _npix], dtype=np.float64).reshape(_npix, _npix)

    @staticmethod
    def _create_filename(file_in: str, quantity: str, on: str, extension: str) -> str:
        """skyio.py:70-94."""
        quantity = quantity.replace("_", "")
        file_out = file_in.split("/")[-1].replace("Ray", quantity)
        file_out = file_out.replace(".h5", f"_lt.{extension}")
        if ("_lc" not in file_in) and ("zrange" not in file_in):
            file_out = file_out.split("_")
            box_string = [string for string in file_in.split("/") if "test" in string][0]
            idx, string = [
                (idx, "%s_" % box_string + string) for idx, string in enumerate(file_out) if "output" in string
            ][0]
            file_out[idx] = string
            file_out = "_".join(file_out)
        _file = file_out.split(".")[:-1] + [on]
        _file = _file + [extension]
        return ".".join(_file)
