#!/usr/bin/env python
"""Run this WHERE jax (and optionally dm-haiku) ARE installed — they are not in the build image — to pin the two things this repo could only restate:

  1. the prenet dropout keep masks the reference draws at inference (vietTTS/nat/text2mel.py:65-73 -> model.py:95-100,134-142): jax.random under
     dm-haiku's PRNGSequence, in whichever threefry layout the installed JAX uses (`jax_threefry_partitionable`: False before 0.5, True from 0.5 on);
  2. what a checkpoint pickled with Haiku / jax objects looks like to viettts_amd.nat.ckpt's tolerant loader.

    python tools/jax_verify_masks.py [--frames 40] [--out tests/golden/jax_masks_golden.npz]

It draws the masks with jax itself, compares them with oracle/nat_oracle.py::haiku_prenet_keep_masks in the matching mode and, with --out, writes them as a
golden fixture (keys: `key`, `partitionable`, `masks`, `jax_version`) that tests can then pin the restatement against.  Exit status 0 = restatement == JAX.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--key", type=int, nargs=2, default=[123456789, 42])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    try:
        import jax
        import jax.numpy as jnp
    except ImportError:
        raise SystemExit("jax is not installed here: run this on a box that has it (the build image has no network)")
    from oracle.nat_oracle import haiku_prenet_keep_masks

    part = bool(getattr(jax.config, "jax_threefry_partitionable", False))
    key = jnp.asarray(a.key, dtype=jnp.uint32)  # a raw uint32[2] key, as the reference's checkpoints hold it
    try:
        import haiku as hk

        seq = hk.PRNGSequence(key)
        draw = lambda: next(seq)
        how = f"dm-haiku {hk.__version__} PRNGSequence"
    except ImportError:
        state = {"k": key}

        def draw():  # PRNGSequence with reserve size 1: (key, sub) = split(key); hand out sub
            state["k"], sub = jax.random.split(state["k"])
            return sub

        how = "jax.random.split chain (dm-haiku absent)"
    masks = np.empty((a.frames, 2, 256), dtype=bool)
    for t in range(a.frames):
        for layer in range(2):
            masks[t, layer] = np.asarray(jax.random.bernoulli(draw(), 0.5, (1, 256)))[0]
    want = haiku_prenet_keep_masks(np.asarray(a.key, dtype=np.uint32), a.frames, 256, partitionable=part)
    same = bool(np.array_equal(masks, want))
    print(f"jax {jax.__version__}, jax_threefry_partitionable = {part}, key chain via {how}: restatement {'==' if same else '!='} JAX "
          f"({int((masks != want).sum())} of {masks.size} bits differ)")
    if a.out:
        np.savez_compressed(a.out, key=np.asarray(a.key, dtype=np.uint32), partitionable=part, masks=masks, jax_version=jax.__version__)
        print("wrote", a.out)
    raise SystemExit(0 if same else 1)


if __name__ == "__main__":
    main()
