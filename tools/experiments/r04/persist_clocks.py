"""Development view of the resident NAT decoder kernel: shader-clock cycles per phase (a library built with -DVTTS_NAT_PERSIST_CLOCKS)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np

from bench import pipeline_256  # noqa: E402
from viettts_amd import _lib  # noqa: E402

r = pipeline_256(256, passes=2)
lib = C.CDLL(str(_lib.default_lib_path()))
buf = (C.c_ulonglong * (256 * 8))()
rc = lib.vtts_nat_debug_persist_clocks(buf)
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
fr = a[:, 6:7]
names = ["LSTM1", "barrier1", "LSTM2", "barrier2", "proj+prenet", "barrier3"]
print("rc", rc, "acoustic_model_ms", round(r["acoustic_model_ms"], 2))
for tile in range(4):
    t = a[64 * tile : 64 * tile + 64]
    per = t[:, :6] / np.maximum(t[:, 6:7], 1)
    print(f"tile {tile}: frames {int(t[0, 6])}: cycles per frame, mean over the tile's workgroups:", {n: int(v) for n, v in zip(names, per.mean(0))}, "sum", int(per.sum(1).mean()))
    print("   proj workgroups (slice < 16):", {n: int(v) for n, v in zip(names, per[:16].mean(0))})
    print("   other workgroups:            ", {n: int(v) for n, v in zip(names, per[16:].mean(0))})
