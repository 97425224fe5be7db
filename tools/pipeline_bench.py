"""BASELINE.json configs[3] on ONE GPU, stage by stage: 256 transcript sentences (synthetic checkpoints) through
viettts_amd.pipeline.synthesize_sentences (duration model -> frame rules -> acoustic model -> HiFi-GAN in ragged passes).  Prints one JSON line with
per-stage wall times — a development view of bench.py's `pipeline_256` (and, with `parity`, of `pipeline_256.parity_grade`).
    usage: pipeline_bench.py [sentences] [passes] [x3 | fp32 | parity]
      x3     = throughput configuration: bf16 vocoder, the acoustic model's bf16x3 option
      fp32   = bf16 vocoder, every acoustic product in fp32 (the default)
      parity = north_star's tolerance: the bf16x3 vocoder in ragged passes, the acoustic model in fp32"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import pipeline_256  # noqa: E402

if __name__ == "__main__":
    mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    gen = None
    if mode == "parity":
        from viettts_amd.hifigan.config import V1
        from viettts_amd.hifigan.generator import Generator
        from viettts_amd.hifigan.synth import synthetic_params

        gen = Generator(V1, device="cuda:0", dtype="bf16x3")
        gen.load_params(synthetic_params(V1, 4321, "scaled"))
    r = pipeline_256(int(sys.argv[1]) if len(sys.argv) > 1 else 256, gen, passes=int(sys.argv[2]) if len(sys.argv) > 2 else 2, nat_bf16x3=mode == "x3")
    r["samples_per_s"] = r["samples"] / (r["total_ms"] * 1e-3)
    r["mode"] = mode
    print(json.dumps(r))
