"""BASELINE.json configs[3] on ONE GPU, stage by stage: 256 synthetic sentences (sentence-like token ids, synthetic
checkpoints) through viettts_amd.pipeline.synthesize_sentences (duration model -> frame rules -> acoustic model -> HiFi-GAN
bf16, ragged batches).  Prints one JSON line with per-stage wall times — a development view of bench.py's `pipeline_256`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import pipeline_256  # noqa: E402

if __name__ == "__main__":
    # usage: pipeline_bench.py [sentences] [passes] [x3]     (x3 = the acoustic model's bf16x3 option)
    r = pipeline_256(int(sys.argv[1]) if len(sys.argv) > 1 else 256, passes=int(sys.argv[2]) if len(sys.argv) > 2 else 2,
                     nat_bf16x3=len(sys.argv) > 3 and sys.argv[3] == "x3")
    r["samples_per_s"] = r["samples"] / (r["total_ms"] * 1e-3)
    print(json.dumps(r))
