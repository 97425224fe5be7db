#!/usr/bin/env python
"""Does a HIP graph help batch-1 latency?    One forward of B = 1 x T = 512, bf16 and fp32, three ways: the
library with its own graph cache off (`graph` = 0: every launch enqueued by the host), on (the default: replays a hipGraph it captured on the
8th call with the same buffers), and captured from OUTSIDE into a torch.cuda.CUDAGraph (the library notices the caller's capture and just
enqueues)."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viettts_amd.hifigan.config import V1  # noqa: E402
from viettts_amd.hifigan.generator import Generator  # noqa: E402
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params  # noqa: E402

dev = torch.device("cuda", 0)
params = synthetic_params(V1, 4321, "scaled")
res = {}
for dtype in ("bf16", "f32"):
    g = Generator(V1, device=dev, dtype=dtype)
    g.load_params(params)
    mel = torch.from_numpy(synthetic_mel(1, 512, 1234)).to(dev)
    out = torch.empty((1, 256 * 512), dtype=torch.float32, device=dev)
    for _ in range(3):
        g(mel, out)
    torch.cuda.synchronize()

    def timed(fn, n=30):
        lat = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        return statistics.median(lat) * 1e3

    g.set_option("graph", 0)
    eager = timed(lambda: g(mel, out))
    ref = out.clone()
    g.set_option("graph", 1)
    for _ in range(10):
        g(mel, out)
    entry = {"eager_ms": eager, "library_graph_ms": timed(lambda: g(mel, out)), "library_graphs_cached": g.get_option("graphs_cached")}
    g.set_option("graph", 0)
    try:
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g(mel, out)  # warm-up on the capture stream (lazy events / streams)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            g(mel, out)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        entry["torch_graph_ms"] = timed(graph.replay)
        entry["graph_output_equal"] = bool(torch.equal(out, ref))
    except Exception as e:  # noqa: BLE001
        entry["graph_error"] = repr(e)[:300]
    res[dtype] = entry
    g.close()
print(json.dumps(res))
