#!/usr/bin/env python
"""bench.py — throughput of the mel->waveform hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the HiFi-GAN V1 generator over one batch of synthetic mels already resident
in HBM: per GPU ``--batch`` utterances x ``--frames`` mel frames (defaults 64 x 1024 = BASELINE.json
configs[2], the shape the samples/sec metric is quoted on).  Weak scaling: every rank runs the same
per-GPU batch on its own shard; the packed weights are broadcast once from rank 0 (RCCL) and there is
no data-path collective.  Rank 0 prints ONE JSON line with:
  value        whole-job audio samples/sec (all N GPUs), inputs resident in HBM
  rtf_b1       latency / RTF of a single 512-frame utterance (BASELINE configs[1]), rank 0
  roofline     dominant ResBlock-conv kernel: algorithmic FLOPs / HIP-event time, vs the MFMA peak
  parity_bf16  max-abs / SNR of the bf16 engine against the reference generator's fp64 output (tests/golden)
  cpu_baseline the REFERENCE's own torch generator (oracle/_ref, built by oracle/build_ref.py) on this box's host cores;
               the numpy port of the oracle when that archive is absent (kind says which)

With ``--gpus N`` (N > 1) and no WORLD_SIZE in the environment the script re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``: one rank per GPU, backend
"nccl" (= RCCL), exactly the launch the driver uses.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

# RCCL hands device buffers between the ranks' processes through HIP IPC; these hosts' driver only offers dmabuf IPC, and the HSA runtime reads
# the switch when it initialises — so it is set before torch (and with it the HIP runtime) is imported, for this process and for the ranks the
# self-launch below starts (viettts_amd/dist.py::ipc_env).  A value the caller exported wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from viettts_amd import dist as vdist  # noqa: E402
from viettts_amd.hifigan.config import V1  # noqa: E402
from viettts_amd.hifigan.generator import Generator  # noqa: E402
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params  # noqa: E402

FLOP_PER_SAMPLE = 2398848  # 2 x MAC of all convolutions per output sample
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0 / 3.0}  # MI355X_MICROARCH.md: dense MFMA peaks (bf16x3: three bf16 MFMAs per product)


def _host_threads() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_reference(budget_s: float = 25.0):
    """the reference's own PyTorch generator (vietTTS/hifigan/torch_model.py:156-218, weight norm removed)
    on ALL host cores, fp32 — compiled to a TorchScript archive by oracle/build_ref.py where /root/reference exists and
    carried to this box under oracle/_ref/.  Same synthetic weights (W_scaled seed 4321) and mels (seed 1234) as the GPU
    run.  B=1 x T=512: 1 warm-up + median of 5 (RTF); B=4 x T=1024: median of 2 (throughput) while the time budget lasts.
    Returns None when the archive is absent."""
    from oracle.build_ref import load_reference_archive  # the checker's loader (cpu_baseline leg only)
    from viettts_amd.hifigan.weights import haiku_to_state_dict

    ncpu = _host_threads()
    sd = {k: torch.from_numpy(v) for k, v in haiku_to_state_dict(V1, synthetic_params(V1, 4321, "scaled")).items()}
    ts = load_reference_archive(sd)
    if ts is None:
        return None
    # BASELINE.md §4 says torch.set_num_threads(os.cpu_count()); on a 256-thread host that measured 52 s per 512-frame utterance
    # (2.5e3 samples/s: OpenMP oversubscription on 32-channel convolutions), 70x slower than 8 threads.  A baseline that
    # handicaps the reference is no baseline: calibrate the thread count on a short input (ascending, stop once it gets
    # clearly slower) and report the count actually used as `cores`.
    xc = torch.from_numpy(synthetic_mel(1, 96, 1234)).permute(0, 2, 1).contiguous()
    calib, best = {}, None
    for nt in sorted({c for c in (4, 8, 16, 32, 64, 128, ncpu) if c <= ncpu}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            ts(xc)
            t0 = time.perf_counter()
            ts(xc)
            calib[nt] = time.perf_counter() - t0
        if best is None or calib[nt] < calib[best]:
            best = nt
        elif calib[nt] > 1.5 * calib[best]:
            break
    cores = best
    torch.set_num_threads(cores)
    t_begin = time.perf_counter()

    def run(B, T, reps, warm):
        x = torch.from_numpy(synthetic_mel(B, T, 1234)).permute(0, 2, 1).contiguous()
        times = []
        with torch.no_grad():
            for i in range(warm + reps):
                t0 = time.perf_counter()
                y = ts(x)
                dt = time.perf_counter() - t0
                if i >= warm:
                    times.append(dt)
                if time.perf_counter() - t_begin > budget_s and times:
                    break
        assert y.shape == (B, 1, 256 * T) and bool(torch.isfinite(y).all())
        return statistics.median(times), len(times)

    med1, n1 = run(1, 512, 5, 1)
    out = {
        "value": 256 * 512 / med1,
        "unit": "samples/s",
        "cores": int(cores),
        "kind": "reference",
        "impl": "reference-torch: vietTTS/hifigan/torch_model.py::Generator (TorchScript archive oracle/_ref/torch_generator_v1.pt.gz), "
                "fp32, torch CPU threads = cores; the reference's JAX-CPU path is not installable offline (no jax/jaxlib/haiku)",
        "sample": f"B=1 x T=512 frames (131072 samples), 1 warm-up + median of {n1}",
        "ms": med1 * 1e3,
        "rtf_16000": med1 / (131072 / 16000.0),
        "rtf_22050": med1 / (131072 / 22050.0),
        "host_threads_available": int(ncpu),
        "thread_calibration_ms_T96": {str(k): round(v * 1e3, 2) for k, v in calib.items()},
    }
    if time.perf_counter() - t_begin < budget_s * 0.5:
        med4, n4 = run(4, 1024, 2, 0)
        out["b4_T1024"] = {"samples_per_s": 4 * 256 * 1024 / med4, "ms": med4 * 1e3, "reps": n4}
    return out


def cpu_baseline(reps: int = 3, T: int = 512):
    """The reference's torch generator when its archive travelled here (kind "reference"); else the oracle (CPU restatement
    of the reference generator, fp32, BLAS-threaded) as a labelled port.  Bounded sample.  Baseline only."""
    ref = cpu_baseline_reference()
    if ref is not None:
        return ref
    from oracle.hifigan_oracle import generator_forward

    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    params = synthetic_params(V1, 4321, "scaled")
    mel = synthetic_mel(1, T, 1234)
    generator_forward(params, mel[:, :32], V1, np.float32)  # warm-up (BLAS threads, page-in)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        generator_forward(params, mel, V1, np.float32)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {
        "value": 256 * T / med,
        "unit": "samples/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"oracle/hifigan_oracle.py fp32 (numpy+BLAS restatement of the reference generator; oracle/_ref archive absent), B=1 x T={T} frames, median of {reps}",
        "ms": med * 1e3,
    }


def parity_bf16(gen, dev):
    """The bf16 engine against the reference generator's fp64 output (tests/golden, minted by oracle/make_golden.py from the
    reference's torch generator): B=1 x T=512 and rows 0, 37, 63 of THIS benchmark's batch (64 x 1024).  Strided samples."""
    out = {}
    gdir = os.path.join(REPO, "tests", "golden")
    try:
        with open(os.path.join(gdir, "golden_meta.json")) as f:
            meta = json.load(f)["cases"]
        for case, B, T in (("v1_scaled_T512", 1, 512), ("v1_scaled_B64_T1024", 64, 1024)):
            rec, g = meta[case], np.load(os.path.join(gdir, case + ".npz"))
            rows = rec.get("rows", list(range(B)))
            mel = torch.from_numpy(synthetic_mel(B, T, rec["mseed"])).to(dev)
            wav, pre = gen.forward_tap(mel, "pre_tanh")
            torch.cuda.synchronize()
            idx = torch.from_numpy(g["idx"]).to(dev)
            y = wav[rows][:, idx].double().cpu().numpy()
            p = pre[rows][:, idx].double().cpu().numpy()
            out[f"B{B}xT{T}"] = {
                "max_abs_wav": float(np.abs(y - g["y64"]).max()),
                "max_abs_pre_tanh": float(np.abs(p - g["pre64"]).max()),
                "snr_db_pre_tanh": float(10 * np.log10((g["pre64"] ** 2).mean() / ((p - g["pre64"]) ** 2).mean())),
                "rows": rows, "samples_compared": int(y.size),
            }
            del wav, pre, mel
        out["reference"] = "fp64 output of vietTTS/hifigan/torch_model.py::Generator on the same seeded weights and mels"
        out["tolerance"] = "north_star fixes 1e-4 for fp32 only (fp32_path below); bf16 bound asserted in tests: max-abs < 0.03, SNR > 38 dB"
    except Exception as e:  # a side report must not take the headline down
        out = {"error": f"{type(e).__name__}: {e}"}
    return out


def pmc_counters(kernel: str, args, B: int, T: int):
    """Counter-derived numbers for the roofline object: HBM bytes per launch of the dominant kernel (`traffic`) and MFMA
    utilisation (dominant kernel; time-weighted over all ResBlock kernels).  PMC counters cannot be read from inside a run (and
    never in the same pass as a timed measurement): they come from the committed rocprofv3 passes of THIS command on THIS build —
    `tools/profile_final.sh` -> `profiles/counters_bf16.json`, which records the digest of the sources it profiled — and are
    reported only while that digest equals the digest of the sources the loaded library was built from (its build stamp) and
    kernel and launch problem match.  Otherwise: None plus the reason."""
    path = os.path.join(REPO, "profiles", f"counters_{args.dtype}.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except OSError:
        return None, None, f"profiles/counters_{args.dtype}.json absent"
    from viettts_amd import _lib
    from viettts_amd.csrc.build import _digest

    stamp_file = str(_lib.default_lib_path()) + ".sha256"
    try:
        with open(stamp_file) as f:
            stamp = f.read().strip()
    except OSError:
        stamp = None
    if stamp is None or stamp != _digest():
        return None, None, "the loaded library's build stamp does not match the sources in the tree"
    if rec.get("source_digest") != stamp:
        return None, None, f"profiles/counters_{args.dtype}.json was measured on another build ({str(rec.get('source_digest'))[:12]} vs {stamp[:12]})"
    if (B, T) != (64, 1024) or args.microbatch not in (0, 64):
        return None, None, "counters were collected at B=64 x T=1024 only"
    hit = [(k, v) for k, v in rec["kernels"].items() if k.startswith(kernel)]
    if len(hit) != 1:
        return None, None, f"kernel {kernel!r} not in the counter file"
    k = hit[0][1]
    traffic = k["hbm_read_bytes"] + k["hbm_write_bytes"] if k.get("hbm_read_bytes") else None
    util = {"dominant_kernel": k.get("mfma_util"), "time_weighted_resblock_kernels": rec.get("time_weighted_mfma_util_resblock_kernels"),
            "lds_bank_conflict_cycles_dominant": k.get("lds_bank_conflict_cycles"),
            "source": f"profiles/{rec.get('tag')}_pmc.md, profiles/{rec.get('tag')}_kernel_stats.md (source digest {stamp[:16]})"}
    return traffic, util, None


def _parity_golden():
    """tests/golden/bench_parity_grade.npz (minted by oracle/make_bench_golden.py, a committed fixture: the oracle itself is imported by the
    cpu_baseline leg only): the fp64 oracle CHAIN's waveforms for three transcript sentences of the pipeline workload and the fp64 oracle
    generator's samples on three windows of the long-form utterance."""
    try:
        return np.load(os.path.join(REPO, "tests", "golden", "bench_parity_grade.npz"))
    except OSError:
        return None


def pipeline_256(n=256, gen=None, rank=0, world=1, barrier=None, passes=3, nat_bf16x3=False, golden=None):
    """BASELINE.json configs[3]: n sentences cycled from the reference's demo transcript x the InfoRe lexicon (SURVEY.md §8d;
    fixtures tests/golden/text/, token ids pinned to the reference's own text2tokens) with synthetic checkpoints -> NAT
    duration model -> frame rules -> NAT acoustic model (prenet dropout on, masks drawn on the device) -> HiFi-GAN bf16 in
    ragged batches; this rank's shard of the sentences (viettts_amd.dist.shard_utterances; no exchange step).  Second pass
    timed, device-synchronised per stage, host work included.  Returns this rank's numbers; main() combines the ranks."""
    import time

    import torch

    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params
    from viettts_amd.nat.acoustic import AcousticModel
    from viettts_amd.nat.duration import DurationModel
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences
    from viettts_amd.pipeline import synthesize_sentences

    own = gen is None
    if own:
        gen = Generator(V1, device="cuda:0", dtype="bf16")
        gen.load_params(synthetic_params(V1, 4321, "scaled"))
    from viettts_amd import dist as vdist

    # rank 0 loads and packs each checkpoint, the other ranks receive the packed blobs: one broadcast per model
    dm = vdist.setup_model_dp(DurationModel(device=str(gen.device)), lambda m: m.load_params(*synthetic_duration_checkpoint()))
    am = vdist.setup_model_dp(AcousticModel(device=str(gen.device)), lambda m: m.load_params(*synthetic_acoustic_checkpoint()))
    if nat_bf16x3:  # the acoustic model's split-precision option (include/vtts_nat.h): for a bf16-class vocoder the mel's 1e-5 is noise
        am.set_option("bf16x3", 1)
    if os.environ.get("VTTS_NAT_PP_SPLIT"):  # development A/B against an experiment build (VTTS_NAT_PP_EXP, tools/r06_nat_ab.sh): the decoder's projection + prenet step cut along its weights
        am.set_option("pp_split", int(os.environ["VTTS_NAT_PP_SPLIT"]))
    tdir = os.path.join(REPO, "tests", "golden", "text")
    sents = transcript_sentences(n, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
    out = {}
    wavs = None
    for _ in range(passes):  # steady state: the first passes warm allocators and code objects; the last one is reported
        tm = {}
        del wavs  # the previous pass's waveforms live in pinned host buffers: released here, the caching host allocator hands them to this pass
                  # (page-locking ~60 MB afresh costs ~20 ms: with 2 passes and the first one's result still referenced that was inside the timed pass)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        t0 = time.perf_counter()
        wavs = synthesize_sentences(sents, dm, am, gen, silence_duration=0.05, dropout_seed=7, rank=rank, world=world, timing=tm)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        nsamp = int(sum(w.shape[0] for w in wavs.values()))
        out = {"workload": f"{n} sentences cycled from assets/transcript.txt (26 lines) x assets/infore/lexicon.txt, synthetic NAT + HiFi-GAN weights, "
                           f"text tokens -> 16 kHz waveform, sharded over {world} GPU(s) with no exchange step",
               "sentences": n, "tokens": tm.get("tokens", 0), "frames": tm.get("frames", 0), "frames_max": tm.get("frames_max", 0), "samples": nsamp,
               "duration_model_ms": tm.get("duration_s", 0.0) * 1e3, "host_rules_ms": tm.get("host_rules_s", 0.0) * 1e3,
               "acoustic_model_ms": tm.get("acoustic_s", 0.0) * 1e3,
               "acoustic_precision": "bf16x3 option" if nat_bf16x3 else "fp32",
               "pinned_alloc_ms": tm.get("pinned_alloc_s", 0.0) * 1e3,
               "generator_ms": tm.get("generator_s", 0.0) * 1e3, "total_ms": total * 1e3}
    if golden is not None and wavs is not None and "total_ms" in out:
        # in-run check of the timed pass's own waveforms against the fp64 oracle chain (text tokens -> duration model -> frame rules -> acoustic model
        # on the same dropout masks -> generator; vietTTS/synthesizer.py:33-39) for the fixture's sentences this rank owns; equal integer frame counts
        worst, checked, frames_equal = 0.0, 0, True
        for i in (int(v) for v in golden["pipe_sentences"]):
            if i in wavs:
                want = golden[f"pipe_{i}_wave"]
                nfr, trail = (int(v) for v in golden[f"pipe_{i}_frames"])
                if wavs[i].shape[0] != 256 * (nfr - trail) or list(sents[i]) != [int(v) for v in golden[f"pipe_{i}_tokens"]]:
                    frames_equal = False
                    continue
                worst = max(worst, float(np.abs(wavs[i].astype(np.float64) - want).max()))
                checked += 1
        out["max_abs_vs_oracle_chain"] = worst if checked else None
        out["oracle_chain_sentences_checked"] = checked
        out["integer_frame_counts_equal"] = frames_equal
    dm.close()
    am.close()
    if own:
        gen.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=1024, help="mel frames per utterance")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "bf16x3"],
                    help="bf16 = BASELINE configs[2] (throughput); f32 = the 1e-4-parity path; bf16x3 = that path's split-operand engine (development runs)")
    ap.add_argument("--no-f32", action="store_true", help="skip the fp32 side measurement")
    ap.add_argument("--microbatch", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="micro-batches in flight on separate HIP streams (0 = engine default)")
    ap.add_argument("--chains", type=int, default=-1, help="engine option 'chains' (0 / 1 / 2; -1 = engine default)")
    ap.add_argument("--fuse", type=int, default=-1, help="engine option 'fuse' (0..3; -1 = engine default)")
    ap.add_argument("--tail", type=int, default=-1, help="engine option 'tail' (bf16: conv_post inside the last pair launch; 0 / 1; -1 = engine default)")
    ap.add_argument("--stage", type=int, default=-1, help="engine option 'stage' (bf16: the whole last stage in one launch; 0 / 1; -1 = engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rtf", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU under torch.distributed.run, the launch line the driver itself uses
        port = vdist.free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        os.execve(sys.executable, cmd, vdist.ipc_env(dict(os.environ)))

    # VTTS_DIST_BACKEND=gloo + VTTS_SHARE_GPU=1: a dry run of the N > 1 code path on a ONE-GPU box (every rank on cuda:0,
    # collectives staged through the host) — a development check, never a measurement
    info = vdist.init_process_group(os.environ.get("VTTS_DIST_BACKEND") or None)
    if info.world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={info.world}: launch one rank per GPU (or unset WORLD_SIZE and let "
                         f"bench.py launch them)")
    n_gpus = info.world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev_index = 0 if os.environ.get("VTTS_SHARE_GPU") else info.local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    gen = Generator(V1, device=dev, dtype=args.dtype)
    if args.microbatch:
        gen.set_option("microbatch", args.microbatch)
    if args.streams:
        gen.set_option("streams", args.streams)
    if args.chains >= 0:
        gen.set_option("chains", args.chains)
    if args.fuse >= 0:
        gen.set_option("fuse", args.fuse)
    if args.tail >= 0:
        gen.set_option("tail", args.tail)
    if args.stage >= 0:
        gen.set_option("stage", args.stage)
    bstats = {}
    vdist.setup_generator_dp(gen, lambda: synthetic_params(V1, 4321, "scaled"), info, bstats)

    B, T = args.batch, args.frames
    # per-rank shard of the global batch: distinct seeded mels, resident in HBM before timing
    mel = torch.from_numpy(synthetic_mel(B, T, 1234 + info.rank)).to(dev)
    out = torch.empty((B, 256 * T), dtype=torch.float32, device=dev)

    def barrier():
        if n_gpus > 1:
            dist.barrier()

    for _ in range(args.warmup):
        gen(mel, out)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gen(mel, out)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    per_rank_ms = None
    if n_gpus > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(n_gpus)]
        dist.all_gather(allt, t)  # every rank's own clock over the same K steps: a straggler shows up here
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline calibration pass (rank 0's kernel timing) ----
    # The product schedule runs a large batch as two half-size passes side by side on two streams (engine.hip: auto_streams), so a kernel shares
    # the GPU and its HIP-event duration is not its own.  The dominant kernel is therefore timed in a short SINGLE-STREAM pass of the same build
    # over the same inputs (options streams = 1, microbatch = the whole batch: round 3's default schedule), HIP events on the launch stream
    # around every launch of that class; rocprofv3 (tools/profile_final.sh -> profiles/) profiles that same schedule.  Not part of `value`.
    calib_steps = max(2, min(args.steps, 5))
    saved = (gen.get_option("streams"), gen.get_option("microbatch"))
    gen.set_option("streams", 1)
    gen.set_option("microbatch", B)
    gen(mel, out)  # warm-up of the schedule (workspace of the full-size pass)
    torch.cuda.synchronize()
    gen.set_option("profile", 1)
    gen.profile_read(reset=True)
    tc0 = time.perf_counter()
    for _ in range(calib_steps):
        gen(mel, out)
    torch.cuda.synchronize()
    calib_ms_per_step = (time.perf_counter() - tc0) / calib_steps * 1e3
    prof = gen.profile_read(reset=True)
    gen.set_option("profile", 0)
    gen.set_option("streams", saved[0])
    gen.set_option("microbatch", saved[1])

    def run_pipeline(g, **kw):
        """One pipeline_256 job on generator g, this rank's shard; the ranks' numbers combined (times: max, counts: sum, errors: max)."""
        try:
            p = pipeline_256(256, g, info.rank, n_gpus, barrier, **kw)
        except Exception as e:  # a side measurement must not take the headline line down with it
            p = {"error": f"{type(e).__name__}: {e}"}
        if n_gpus > 1:  # the ranks agree on whether to combine (a failed rank would otherwise leave the others in a collective)
            ok = torch.tensor([0.0 if "error" in p else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0 and "error" not in p:
                p = {"error": "another rank failed"}
        if n_gpus > 1 and "error" not in p:
            keys_max = ["duration_model_ms", "host_rules_ms", "acoustic_model_ms", "generator_ms", "total_ms", "frames_max", "pinned_alloc_ms"]
            keys_sum = ["tokens", "frames", "samples"]
            if "oracle_chain_sentences_checked" in p:
                p["max_abs_vs_oracle_chain"] = p["max_abs_vs_oracle_chain"] or 0.0
                p["integer_frame_counts_unequal"] = 0.0 if p.pop("integer_frame_counts_equal") else 1.0
                keys_max += ["max_abs_vs_oracle_chain", "integer_frame_counts_unequal"]
                keys_sum += ["oracle_chain_sentences_checked"]
            tmax = torch.tensor([float(p[k]) for k in keys_max], dtype=torch.float64, device=dev)
            tsum = torch.tensor([float(p[k]) for k in keys_sum], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            for k, v in zip(keys_max, tmax.tolist()):
                p[k] = int(v) if k == "frames_max" else v
            for k, v in zip(keys_sum, tsum.tolist()):
                p[k] = int(v)
            if "integer_frame_counts_unequal" in p:
                p["integer_frame_counts_equal"] = p.pop("integer_frame_counts_unequal") == 0.0
            if p.get("oracle_chain_sentences_checked", None) == 0:  # no rank compared a sentence: "0.0" would read as a perfect match
                p["max_abs_vs_oracle_chain"] = None
        if "error" not in p:
            p["samples_per_s"] = p["samples"] / (p["total_ms"] * 1e-3)
            p["sentences_per_s"] = p["sentences"] / (p["total_ms"] * 1e-3)
        return p

    # ---- text -> waveform (BASELINE configs[3]): 256 sentences sharded over the ranks; whole job = max over ranks ----
    # Two peers: `pipeline_256` = the throughput configuration (bf16 vocoder ~1e-2 of a sample's range, so the acoustic model runs with
    # its bf16x3 option, 1e-5 of the mel's range: include/vtts_nat.h), and `pipeline_256.parity_grade` = the configuration that answers to
    # north_star's 1e-4: the split-operand (bf16x3) vocoder in ragged passes + the acoustic model in fp32, the mode pinned to the reference's code.
    pipe = None
    gx_side = None  # the split-operand generator of the parity-grade legs (every rank: the legs shard like their bf16 peers)
    golden = _parity_golden()
    if not args.no_rtf and args.dtype == "bf16" and not args.no_f32:
        try:
            gx_side = Generator(V1, device=dev, dtype="bf16x3")
            vdist.setup_generator_dp(gx_side, lambda: synthetic_params(V1, 4321, "scaled"), info, {})
        except Exception:
            gx_side = None
    if not args.no_rtf and args.dtype == "bf16":
        pipe = run_pipeline(gen, nat_bf16x3=True)
        if "error" not in pipe:
            pipe["vocoder"] = "bf16 engine"
        if gx_side is not None and "error" not in pipe:
            pg = run_pipeline(gx_side, nat_bf16x3=False, golden=golden)
            if "error" not in pg:
                pipe["parity_grade"] = {
                    "what": "the same 256 sentences with the vocoder on the split-operand engine (bf16x3: <= 5e-5 of the reference generator on identical mels, "
                            "tests/test_gpu_nat.py) in ragged passes and every acoustic product in fp32 (the mode pinned to the reference's code at 5e-5)",
                    "vocoder": "bf16x3 engine", "acoustic_precision": "fp32",
                    "total_ms": pg["total_ms"], "acoustic_model_ms": pg["acoustic_model_ms"], "generator_ms": pg["generator_ms"],
                    "samples_per_s": pg["samples_per_s"], "sentences_per_s": pg["sentences_per_s"],
                    # in-run, the timed pass's own waveforms: the fp64 oracle CHAIN end to end (acoustic model included; tests check the vocoder alone on
                    # the same GPU mel at 5e-5), tests/golden/bench_parity_grade.npz
                    "max_abs_vs_oracle_chain": pg.get("max_abs_vs_oracle_chain"), "oracle_chain_sentences_checked": pg.get("oracle_chain_sentences_checked"),
                    "integer_frame_counts_equal": pg.get("integer_frame_counts_equal"),
                    "reference": "oracle/make_bench_golden.py: nat_oracle.duration_model -> frame rules -> nat_oracle.acoustic_inference (fp64, same threefry masks) -> hifigan_oracle (fp64)",
                }
            else:
                pipe["parity_grade"] = pg
        if n_gpus == 1 and "error" not in pipe:
            # the throughput configuration with every acoustic product in fp32 (round 4's `acoustic_fp32`): what the acoustic model's option is worth
            try:
                pf = pipeline_256(256, gen, info.rank, n_gpus, barrier)
                pipe["acoustic_fp32"] = {"acoustic_model_ms": pf["acoustic_model_ms"], "generator_ms": pf["generator_ms"], "total_ms": pf["total_ms"],
                                         "samples_per_s": pf["samples"] / (pf["total_ms"] * 1e-3)}
            except Exception as e:
                pipe["acoustic_fp32"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- long-form (BASELINE configs[4]): 10 min of 16 kHz audio, exact 512-frame chunks + 13-frame halo, chunk c -> rank c mod N ----
    longform = None
    if not args.no_rtf:
        from viettts_amd.longform import synthesize_chunked

        T10 = 37500  # 600 s * 16000 / 256
        m10 = torch.from_numpy(synthetic_mel(1, T10, 99)[0]).to(dev)  # every rank holds the utterance's mel (12 MB); no exchange step
        synthesize_chunked(gen, m10, 512, rank=info.rank, world=n_gpus)  # warm-up of the chunk shapes (workspace of the full-size pass)
        torch.cuda.synchronize()
        barrier()
        tm = {}
        synthesize_chunked(gen, m10, 512, rank=info.rank, world=n_gpus, timing=tm)
        tl = torch.tensor([tm.get("first_chunk_s", 0.0), tm["total_s"], float(tm["chunks"])], dtype=torch.float64, device=dev)
        if n_gpus > 1:
            tmax, tsum = tl.clone(), tl.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            tl = torch.stack([tmax[0], tmax[1], tsum[2]])
        first_s, total_s, nchunks = (float(v) for v in tl.tolist())
        longform = {
            "workload": f"one 37500-frame utterance (600 s @16 kHz), 512-frame chunks + 13-frame halo, the first chunk alone then full-size passes, "
                        f"chunk c -> rank c mod {n_gpus} (no exchange step), max over ranks",
            "first_chunk_ms": first_s * 1e3,
            "total_ms": total_s * 1e3,
            "rtf_16000": total_s / 600.0,
            "samples_per_s": 256 * T10 / total_s,
            "chunks": int(nchunks),
        }
        if gx_side is not None:
            try:
                synthesize_chunked(gx_side, m10, 512, rank=info.rank, world=n_gpus)  # warm-up
                torch.cuda.synchronize()
                barrier()
                tmx = {}
                wx = synthesize_chunked(gx_side, m10, 512, rank=info.rank, world=n_gpus, timing=tmx)
                # in-run: three windows of THIS pass's output (utterance start, the chunk seam at 512 * 37, utterance end) against the fp64 oracle generator
                # (tests/golden/bench_parity_grade.npz); a rank checks the frames of the chunks it owns (chunk c -> rank c mod N)
                err, nchk = 0.0, 0
                if golden is not None:
                    for name in ("start", "seam", "end"):
                        lo = int(golden[f"lf_{name}_lo"])
                        want = torch.from_numpy(golden[f"lf_{name}_wave"]).to(dev)
                        got = wx[256 * lo : 256 * lo + want.numel()].double()
                        fr = torch.arange(lo, lo + want.numel() // 256, device=dev)
                        own = (((fr // 512) % n_gpus) == info.rank).repeat_interleave(256)
                        if bool(own.any()):
                            err = max(err, float((got - want).abs()[own].max()))
                            nchk += int(own.sum())
                tx = torch.tensor([tmx.get("first_chunk_s", 0.0), tmx["total_s"], err, float(nchk)], dtype=torch.float64, device=dev)
                if n_gpus > 1:
                    tmax, tsum = tx.clone(), tx.clone()
                    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                    tx = torch.stack([tmax[0], tmax[1], tmax[2], tsum[3]])
                fx, totx, errx, nx = (float(v) for v in tx.tolist())
                longform["parity_grade"] = {
                    "what": "the same utterance and schedule on the split-operand engine (bf16x3; chunked == un-chunked bit for bit and <= 5e-5 of the fp64 oracle: tests/test_gpu_longform.py)",
                    "vocoder": "bf16x3 engine", "first_chunk_ms": fx * 1e3, "total_ms": totx * 1e3, "rtf_16000": totx / 600.0, "samples_per_s": 256 * T10 / totx,
                    "max_abs_vs_fp64_oracle_windows": errx if nx else None, "samples_compared": int(nx), "bar": 1e-4,
                    "reference": "oracle/make_bench_golden.py: hifigan_oracle.generator_forward (fp64) on frames [lo - 13, hi + 13) of the utterance's mel, windows at the start, at the seam 512 x 37 and at the end",
                }
                del wx
            except Exception as e:
                longform["parity_grade"] = {"error": f"{type(e).__name__}: {e}"}
        longform["vocoder"] = f"{args.dtype} engine"
        del m10

    assert bool(torch.isfinite(out).all()), "non-finite output"
    assert float(out.abs().max()) <= 1.0, "waveform outside tanh's range"

    if info.rank == 0:
        samples_per_step = n_gpus * B * 256 * T
        value = samples_per_step * args.steps / elapsed
        res = {
            "metric": "audio samples/sec whole-node, HiFi-GAN V1 mel2wave",
            "value": value,
            "unit": "samples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded random weights of the V1 architecture, seeded log-mel-like inputs)",
            "config": {
                "workload": f"HiFi-GAN V1 generator, {B} utterances x {T} mel frames per GPU per step (BASELINE configs[2] shape), {args.dtype}",
                "batch_per_gpu": B,
                "frames": T,
                "samples_per_step": samples_per_step,
                "parallelism": f"dp{n_gpus} utterance-sharded, weights broadcast once, no data-path collective",
                "microbatch": gen.get_option("microbatch"),
                "streams": gen.get_option("streams"),
                "schedule": "engine default: micro-batches of <= 32768 frames on two HIP streams" if not (args.microbatch or args.streams) else "as given by --microbatch / --streams",
            },
            "weights_broadcast": {"backend": bstats.get("backend"), "ranks_in_group": bstats.get("world"), "bytes": bstats.get("bytes"),
                                  "ms_rank0": bstats.get("broadcast_ms"),
                                  # after the broadcast every rank checksums its blob, MIN / MAX all-reduced: the job stops if they differ (null at N = 1)
                                  "blob_checksum": bstats.get("blob_checksum"), "blob_checksum_equal": bstats.get("blob_checksum_equal")},
            "ms_per_step_per_rank": per_rank_ms,
            "tflops_whole_job": value * FLOP_PER_SAMPLE / 1e12,
            "frac_of_mfma_peak_whole_forward": value * FLOP_PER_SAMPLE / 1e12 / (PEAK_TFLOPS[args.dtype] * n_gpus),
        }
        # ---- roofline of the dominant kernel, HIP events on the launch stream, timed region only ----
        if prof["launches"] > 0 and prof["ms"] > 0:
            ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
            traffic, util, why_not = pmc_counters(prof["kernel"], args, B, T)
            res["roofline"] = {
                "bound": "mfma",
                "achieved": ach,
                "peak": PEAK_TFLOPS[args.dtype],
                "unit": "TFLOP/s",
                "frac": ach / PEAK_TFLOPS[args.dtype],
                "traffic": traffic,  # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) or null
                # calibration, not the contract's peak: a pure MFMA stream (operands in registers, no LDS / memory traffic) sustains 1750 TF/s on pseudo-random
                # bf16 operands and 2460 on constant ones on this chip's power budget (profiles/r03_e_mfma_peak.txt, tools/kbench/coissue 400 3)
                "peak_sustained_mfma_only_random_operands": 1750.0 if args.dtype == "bf16" else None,
                # recorded, not measured in this run (round 5, profiles/r05_i_kstep_asm_findings.md): the dominant kernel's OWN k-step loop cut loose from staging and
                # epilogues (weights from L2, activations from LDS, two workgroups per CU) sustains 1703 TF/s over 0.7 s at 1.33-1.35 kW / 1.74-1.79 GHz, and the
                # MfmaUtil counter formula below reads 0.87 for it — the ceilings `achieved` and `mfma_util_dominant_kernel` are to be read against
                "recorded_sustained_own_kstep_loop_tflops": 1703.0 if args.dtype == "bf16" else None,
                "recorded_mfma_util_of_own_kstep_loop": 0.87 if args.dtype == "bf16" else None,
                # next to `frac`, never instead of it: `achieved` against that recorded ceiling of the kernel's own loop
                "frac_of_recorded_ceiling": ach / 1703.0 if args.dtype == "bf16" else None,
                "traffic_algorithmic": 2.0 * B * (T * 64) * 128 * 2 if prof["kernel"].startswith("resblock_pair_g_bf16_k<GTile<128, 11") else None,
                # flat numeric keys (the driver's `parsed.roofline` keeps numbers): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
                "mfma_util_dominant_kernel": util.get("dominant_kernel") if util else None,
                "mfma_util_time_weighted_resblock_kernels": util.get("time_weighted_resblock_kernels") if util else None,
                "lds_bank_conflict_cycles_dominant_kernel": util.get("lds_bank_conflict_cycles_dominant") if util else None,
                "mfma_util": util,
                "counters_unavailable": why_not,
                "kernel": prof["kernel"],
                "launches": prof["launches"],
                "avg_launch_ms": prof["ms"] / prof["launches"],
                "flops_per_launch": prof["flops"] / prof["launches"],
                "timed_in": f"calibration pass: {calib_steps} single-stream passes (streams = 1, microbatch = {B}) of the same build over the same inputs, "
                            f"right after the timed region; the timed region itself runs the engine's default schedule "
                            f"(two half-size passes side by side), under which a kernel's duration is not its own",
                "calibration_ms_per_step": calib_ms_per_step,
            }
        else:
            res["roofline"] = None

        if args.dtype == "bf16" and not args.no_rtf:
            res["parity_bf16"] = parity_bf16(gen, dev)
        # ---- RTF at batch 1 (BASELINE configs[1]: B=1, T=512) ----
        if not args.no_rtf:
            m1 = torch.from_numpy(synthetic_mel(1, 512, 1234)).to(dev)
            o1 = torch.empty((1, 256 * 512), dtype=torch.float32, device=dev)
            for _ in range(3):
                gen(m1, o1)
            torch.cuda.synchronize()
            lat = []
            for _ in range(20):
                t1 = time.perf_counter()
                gen(m1, o1)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            med = statistics.median(lat)
            gen.set_option("graph", 0)  # the same call with every launch enqueued by the host (no hipGraph replay)
            lat_e = []
            for _ in range(10):
                t1 = time.perf_counter()
                gen(m1, o1)
                torch.cuda.synchronize()
                lat_e.append(time.perf_counter() - t1)
            gen.set_option("graph", 1)
            res["rtf_b1"] = {
                "workload": "B=1, T=512 frames (131072 samples), the same device buffers every call (steady-state serving): the stage's ResBlocks "
                            "run on parallel streams and, from the 8th call on, the library replays the hipGraph it captured",
                "latency_ms": med * 1e3,
                "latency_ms_eager_launches": statistics.median(lat_e) * 1e3,
                "rtf_16000": med / (131072 / 16000.0),
                "rtf_22050": med / (131072 / 22050.0),
                "samples_per_s": 131072 / med,
            }
        # ---- the fp32 (1e-4 parity) path, same run: throughput on a smaller batch + batch-1 RTF ----
        if args.dtype != "f32" and not args.no_f32:
            g32 = Generator(V1, device=dev, dtype="f32")
            g32.load_params(synthetic_params(V1, 4321, "scaled"))
            Bf = B  # the headline shape (64 x 1024): two half-size passes side by side (engine.hip: auto_streams)
            o32 = torch.empty((Bf, 256 * T), dtype=torch.float32, device=dev)
            g32(mel[:Bf], o32)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                g32(mel[:Bf], o32)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / 2
            m1 = torch.from_numpy(synthetic_mel(1, 512, 1234)).to(dev)
            o1 = torch.empty((1, 256 * 512), dtype=torch.float32, device=dev)
            for _ in range(3):
                g32(m1, o1)
            torch.cuda.synchronize()
            lat = []
            for _ in range(20):
                t2 = time.perf_counter()
                g32(m1, o1)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t2)
            med = statistics.median(lat)
            v32 = Bf * 256 * T / dt32
            par32 = None
            try:  # the <= 1e-4 of BASELINE.json at THIS shape and schedule, in-run: rows 0, 37, 63 of the batch against the reference generator's fp64 output
                gdir = os.path.join(REPO, "tests", "golden")
                with open(os.path.join(gdir, "golden_meta.json")) as f:
                    rec = json.load(f)["cases"]["v1_scaled_B64_T1024"]
                if (Bf, T) == (rec["B"], rec["T"]) and info.rank == 0 and rec["mseed"] == 1234:
                    g = np.load(os.path.join(gdir, "v1_scaled_B64_T1024.npz"))
                    idx = torch.from_numpy(g["idx"]).to(dev)
                    y = o32[rec["rows"]][:, idx].double().cpu().numpy()
                    par32 = {"max_abs_wav_vs_fp64_reference": float(np.abs(y - g["y64"]).max()), "max_abs_wav_vs_fp32_reference": float(np.abs(y - g["y32"]).max()),
                             "rows": rec["rows"], "samples_compared": int(y.size), "bar": 1e-4,
                             "reference": "vietTTS/hifigan/torch_model.py::Generator on the same seeded weights and mels (tests/golden/v1_scaled_B64_T1024.npz)"}
            except Exception as e:
                par32 = {"error": f"{type(e).__name__}: {e}"}
            res["fp32_path"] = {
                "workload": f"{Bf} x {T} frames (the engine's default schedule: micro-batches of {g32.get_option('microbatch') or min(Bf, -(-32768 // T))} on two streams), fp32 MFMA kernels (parity <= 1e-4 vs the reference)",
                "samples_per_s": v32,
                "tflops": v32 * FLOP_PER_SAMPLE / 1e12,
                "frac_of_f32_mfma_peak": v32 * FLOP_PER_SAMPLE / 1e12 / PEAK_TFLOPS["f32"],
                "parity": par32,
                "b1_T512_latency_ms": med * 1e3,
                "rtf_16000": med / (131072 / 16000.0),
                "rtf_22050": med / (131072 / 22050.0),
            }
            g32.close()
        # ---- the split-operand engine (VTTS_BF16X3): fp32-grade parity on the bf16 matrix pipe, same shape, same checks as fp32_path ----
        if args.dtype == "bf16" and not args.no_f32:
            try:
                gx = Generator(V1, device=dev, dtype="bf16x3")
                gx.load_params(synthetic_params(V1, 4321, "scaled"))
                ox = torch.empty((B, 256 * T), dtype=torch.float32, device=dev)
                gx(mel, ox)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    gx(mel, ox)
                torch.cuda.synchronize()
                dtx = (time.perf_counter() - t1) / 3
                parx = None
                gdir = os.path.join(REPO, "tests", "golden")
                with open(os.path.join(gdir, "golden_meta.json")) as f:
                    rec = json.load(f)["cases"]["v1_scaled_B64_T1024"]
                if (B, T) == (rec["B"], rec["T"]) and info.rank == 0:
                    g = np.load(os.path.join(gdir, "v1_scaled_B64_T1024.npz"))
                    y = ox[rec["rows"]][:, torch.from_numpy(g["idx"]).to(dev)].double().cpu().numpy()
                    parx = {"max_abs_wav_vs_fp64_reference": float(np.abs(y - g["y64"]).max()), "rows": rec["rows"], "samples_compared": int(y.size), "bar": 1e-4}
                # dominant pair class, one-stream calibration as for the headline
                gx.set_option("streams", 1)
                gx.set_option("microbatch", B)
                gx(mel, ox)
                torch.cuda.synchronize()
                gx.set_option("profile", 1)
                gx.profile_read(reset=True)
                for _ in range(2):
                    gx(mel, ox)
                torch.cuda.synchronize()
                px = gx.profile_read(reset=True)
                m1 = torch.from_numpy(synthetic_mel(1, 512, 1234)).to(dev)
                o1 = torch.empty((1, 256 * 512), dtype=torch.float32, device=dev)
                gx.set_option("profile", 0)
                gx.set_option("streams", 0)
                gx.set_option("microbatch", 0)
                for _ in range(3):
                    gx(m1, o1)
                torch.cuda.synchronize()
                lat = []
                for _ in range(20):
                    t2 = time.perf_counter()
                    gx(m1, o1)
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t2)
                vx = B * 256 * T / dtx
                res["bf16x3_path"] = {
                    "workload": f"{B} x {T} frames, the fp32 engine's layouts and schedule with the ResBlock convolutions on the bf16 matrix pipe, every product "
                                f"three bf16 x bf16 terms of two-term operand splits (kernels_x3.hip); answers to the fp32 bar (1e-4)",
                    "samples_per_s": vx, "ms_per_step": dtx * 1e3,
                    "speedup_vs_fp32_engine": vx / res["fp32_path"]["samples_per_s"] if "fp32_path" in res else None,
                    "algorithmic_tflops": vx * FLOP_PER_SAMPLE / 1e12,
                    "parity": parx,
                    "dominant_kernel": {"kernel": px["kernel"], "launches": px["launches"], "avg_launch_ms": px["ms"] / max(px["launches"], 1),
                                        "algorithmic_tflops": px["flops"] / max(px["ms"], 1e-9) / 1e9,
                                        "bf16_mfma_tflops_issued": 3.0 * px["flops"] / max(px["ms"], 1e-9) / 1e9,
                                        "frac_of_bf16_mfma_peak_issued": 3.0 * px["flops"] / max(px["ms"], 1e-9) / 1e9 / PEAK_TFLOPS["bf16"],
                                        "timed_in": "single-stream calibration pass"},
                    "b1_T512_latency_ms": statistics.median(lat) * 1e3,
                    "rtf_16000": statistics.median(lat) / (131072 / 16000.0),
                }
                gx.close()
                # the headline is bf16-grade (configs[2] names bf16; parity_bf16 says how far from the reference); the rate AT north_star's 1e-4 bar is the
                # split engine's — flat, beside the headline's own roofline figures, so that the one-line record carries both
                if res.get("roofline"):
                    res["roofline"]["parity_grade_samples_per_s"] = vx * n_gpus if n_gpus == 1 else None  # measured on rank 0's GPU only
                    res["roofline"]["parity_grade_ms_per_step"] = dtx * 1e3
                    res["roofline"]["parity_grade_max_abs_vs_fp64_reference"] = parx["max_abs_wav_vs_fp64_reference"] if parx else None
            except Exception as e:  # a side report must not take the headline down
                res["bf16x3_path"] = {"error": f"{type(e).__name__}: {e}"}
        if longform is not None:
            res["longform_10min"] = longform
        if pipe is not None:
            res["pipeline_256"] = pipe
        if n_gpus == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        else:  # the key is always there: the CPU baseline is timed on rank 0 of the N = 1 run only (it would otherwise compete with the ranks' host threads)
            res["cpu_baseline"] = None
            res["cpu_baseline_skipped"] = "--no-cpu-baseline" if n_gpus == 1 else f"N = {n_gpus} > 1: the reference's CPU path is timed in the N = 1 run only"
        print(json.dumps(res), flush=True)

    barrier()
    gen.close()
    if gx_side is not None:
        gx_side.close()
    if n_gpus > 1 and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
