"""ctypes loader of experiments/libunidisc_exp.so (built on demand from experiments/csrc): used by experiments/scripts only."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def load_experiments():
    path = os.path.join(HERE, "libunidisc_exp.so")
    if not os.path.exists(path):
        out = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j4"], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("building libunidisc_exp.so failed:\n" + out.stderr[-3000:])
    lib = ctypes.CDLL(path)
    lib.udm_last_error.restype = ctypes.c_char_p
    return lib
