"""IO helpers with the reference's names (src/astrild/io.py:10-31)."""
import os

import numpy as np


class IO:
    @staticmethod
    def _remove_existing_file(filename) -> None:
        if os.path.exists(filename):
            os.remove(filename)

    @staticmethod
    def save_skymap(data, filename: str) -> None:
        IO._remove_existing_file(filename)
        print("Save in:\n   %s" % filename)
        if isinstance(data, np.ndarray):
            np.save(filename, data)
        else:                                   # astropy PrimaryHDU
            data.writeto(filename)

    @staticmethod
    def save_dataFrame(direct: str, filename: str, df) -> None:
        file_path = direct + filename
        IO._remove_existing_file(file_path)
        print("Save to -> ", file_path)
        df.to_hdf(file_path, key="df", mode="w")
