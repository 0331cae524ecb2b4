"""Identity of the library build: a hash of every source file libpolyfem_amd.so is compiled from.  Profiles record it, and
the benchmark only quotes a recorded counter figure whose hash equals that of the sources it is running."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_sha256() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(_HERE, "csrc", "Makefile"),
                                                                         os.path.join(_HERE, "..", "include", "polyfem_amd.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
