"""rel(): relative Frobenius error, recorded per call site so that the tolerances in the GPU tests can be (and stay) set at <= 10x what the
engine actually achieves (written to gpurun_out/parity_observed.json by conftest at the end of a GPU session)."""
import inspect
import os

import numpy as np

RECORD = {}


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    v = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    fr = inspect.stack()[1]
    key = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    if not (RECORD.get(key, -1.0) >= v):
        RECORD[key] = v
    return v
