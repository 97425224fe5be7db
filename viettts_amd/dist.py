"""Data-parallel sharding of the mel->waveform path across the GPUs of one node.

The reference has no inference parallelism at all (``mel2wave`` is single-utterance, single-device;
its only parallel code is the TPU ``pmap`` trainer, vietTTS/nat/acoustic_tpu_trainer.py:38-53, out of
scope).  The path shards naturally: utterances are independent, and time chunks of a
long utterance are independent given a 13-frame mel halo.  So: one process per GPU, the packed weight
blob is broadcast ONCE from rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests),
and there is NO data-path collective.  Results stay on the rank that produced them unless the caller
asks for a gather.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

HALO_FRAMES = 13  # receptive field of the generator is +-12.71 mel frames


@dataclass(frozen=True)
class RankInfo:
    rank: int
    world: int
    local_rank: int


def rank_info() -> RankInfo:
    """RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run exports them (1-process default)."""
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def ipc_env(env=None):
    """``HSA_ENABLE_IPC_MODE_LEGACY=0`` in ``env`` (default: this process's environment) unless the caller already chose a value.
    RCCL's intra-node transport hands device buffers between the ranks' processes through HIP IPC handles; on hosts whose kernel driver
    only offers dmabuf IPC (the MI355X boxes this build runs on) the legacy mode fails with ``hipIpcGetMemHandle: invalid argument``
    at the first collective.  The HSA runtime reads the variable when it initialises, so it has to be in the environment BEFORE the
    process's first HIP call: the launchers (``bench.py`` before it imports torch, and its self-launch) set it for their children; calling this
    from :func:`init_process_group` covers a host that initialises HIP lazily after it."""
    e = os.environ if env is None else env
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init_process_group(backend: Optional[str] = None) -> RankInfo:
    """One process per GPU.  backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
    The rendezvous address comes from the launcher (``torch.distributed.run`` exports MASTER_ADDR / MASTER_PORT); without one the address
    defaults to 127.0.0.1, and a missing MASTER_PORT is an error for world > 1 — ranks that each picked "a free port" would never meet, and a
    fixed default collides with whatever else runs on the node (use :func:`launch_env` to build the children's environment)."""
    info = rank_info()
    if info.world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        ipc_env()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("viettts_amd.dist.init_process_group: WORLD_SIZE > 1 but MASTER_PORT is not set — launch the ranks with "
                               "torch.distributed.run (bench.py --gpus N does) or export the port viettts_amd.dist.launch_env() picked")
        if backend == "nccl":
            torch.cuda.set_device(info.local_rank)
        dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world)
    return info


def launch_env(world: int, rank: int, local_rank: Optional[int] = None, port: Optional[int] = None, env=None) -> dict:
    """Environment of one rank for a launcher that does not go through ``torch.distributed.run``: RANK / WORLD_SIZE / LOCAL_RANK, the
    rendezvous on 127.0.0.1 at ``port`` (the PARENT takes a free one once — :func:`free_port` — and hands the same number to every rank)
    and the IPC mode RCCL needs (:func:`ipc_env`)."""
    e = dict(os.environ if env is None else env)
    if port is None:
        raise ValueError("launch_env: the parent picks ONE port (viettts_amd.dist.free_port()) and passes it to every rank")
    e.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if local_rank is None else local_rank), MASTER_ADDR="127.0.0.1",
             MASTER_PORT=str(port))
    return ipc_env(e)


free_port = _free_port


# ---------------------------------------------------------------------------------------------------
# partitioning (pure functions; identical on every rank, so no communication is needed to agree)
# ---------------------------------------------------------------------------------------------------
def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-first assignment of utterance indices to ranks by total frame count.
    Deterministic (ties broken by index); every index appears exactly once."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += int(lengths[i])
    for r in range(world):
        out[r].sort()
    return out


@dataclass(frozen=True)
class Chunk:
    index: int
    t0: int  # first mel frame whose samples this chunk keeps
    t1: int  # one past the last
    lo: int  # first mel frame fed to the generator (t0 - halo, clipped at the utterance start)
    hi: int  # one past the last frame fed (t1 + halo, clipped at the end)

    @property
    def keep_from(self) -> int:
        """Frames to drop from the front of the chunk's output."""
        return self.t0 - self.lo

    @property
    def frames(self) -> int:
        return self.hi - self.lo


def plan_chunks(T: int, chunk_frames: int, halo: int = HALO_FRAMES) -> List[Chunk]:
    """Cut ``T`` frames into chunks of ``chunk_frames`` kept frames + ``halo`` context frames per
    side.  At true utterance edges there is no halo: the generator's own zero padding
    (get_padding, vietTTS/hifigan/model.py:8-10) applies there, exactly as un-chunked."""
    if T < 1 or chunk_frames < 1 or halo < 0:
        raise ValueError("T, chunk_frames must be >= 1 and halo >= 0")
    out = []
    t0, i = 0, 0
    while t0 < T:
        t1 = min(T, t0 + chunk_frames)
        out.append(Chunk(i, t0, t1, max(0, t0 - halo), min(T, t1 + halo)))
        t0, i = t1, i + 1
    return out


def shard_chunks(chunks: Sequence[Chunk], world: int) -> List[List[Chunk]]:
    """chunk c -> rank c mod world."""
    return [[c for c in chunks if c.index % world == r] for r in range(world)]


# ---------------------------------------------------------------------------------------------------
# the one collective: weight broadcast at start-up
# ---------------------------------------------------------------------------------------------------
def broadcast_packed_weights(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Broadcast the packed weight blob (uint8 tensor, identical size on every rank) in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def blob_checksum(blob: torch.Tensor) -> int:
    """A 64-bit checksum of a packed weight blob (uint8 tensor), computed where the blob lives: sum over the 8-byte words of
    ``word * (2 * index + 1)`` modulo 2^64 — position-weighted, so swapped or shifted words change it as well as flipped bits."""
    b = blob.reshape(-1)
    pad = (-b.numel()) % 8
    if pad:
        b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
    w = b.view(torch.int64)
    k = torch.arange(w.numel(), dtype=torch.int64, device=w.device) * 2 + 1
    return int((w * k).sum().item())  # int64 arithmetic wraps modulo 2^64


def verify_blob_on_all_ranks(blob: torch.Tensor, what: str = "packed weights") -> int:
    """After the start-up broadcast: every rank checksums the blob it holds, the checksums are all-reduced with MIN and MAX, and
    every rank fails fast if they differ — a rank that computed on a damaged copy of the weights would otherwise produce
    plausible-looking audio forever.  A start-up collective (two 8-byte all-reduces per model), not a data-path one.
    Returns the checksum."""
    cs = blob_checksum(blob)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        lo = torch.tensor([cs], dtype=torch.int64, device=blob.device)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if int(lo.item()) != int(hi.item()):
            raise RuntimeError(f"{what}: the ranks hold different blobs after the start-up broadcast (this rank's checksum {cs & 0xFFFFFFFFFFFFFFFF:#018x}; "
                               f"min {int(lo.item()) & 0xFFFFFFFFFFFFFFFF:#018x}, max {int(hi.item()) & 0xFFFFFFFFFFFFFFFF:#018x} over {dist.get_world_size()} ranks)")
    return cs


def setup_model_dp(model, load_fn, info: Optional[RankInfo] = None, stats: Optional[dict] = None):
    """Rank 0 loads + packs the weights (``load_fn(model)`` calls its ``load_params`` and runs on rank 0 only), every other
    rank receives the packed blob: ONE broadcast per model at start-up (HiFi-GAN generator 27.9 MB bf16 / 55.7 MB fp32,
    NAT duration model, NAT acoustic model).  ``model`` offers ``packed_bytes``, ``packed_blob()``, ``adopt_packed(blob)``
    and ``device`` (Generator, DurationModel, AcousticModel).  ``stats`` (a dict) receives ``bytes``, ``broadcast_ms`` (device-
    synchronised wall time of the one collective on this rank), ``backend`` and ``world`` as the process group reports them, and
    ``blob_checksum`` / ``blob_checksum_equal`` (:func:`verify_blob_on_all_ranks`: the job stops here if a rank's copy differs)."""
    import time

    info = info or rank_info()
    if info.world == 1 or info.rank == 0:
        load_fn(model)
        blob = model.packed_blob()
    else:
        blob = torch.empty(model.packed_bytes, dtype=torch.uint8, device=model.device)
    if stats is not None:
        stats.update(bytes=int(blob.numel()), broadcast_ms=0.0, backend=None, world=1)
    if info.world > 1:
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        t0 = time.perf_counter()
        broadcast_packed_weights(blob, 0)
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        if stats is not None:
            stats.update(broadcast_ms=(time.perf_counter() - t0) * 1e3, backend=dist.get_backend(), world=dist.get_world_size())
        cs = verify_blob_on_all_ranks(blob, type(model).__name__ + " weights")  # raises on every rank if any rank's copy differs
        if stats is not None:
            stats.update(blob_checksum=f"{cs & 0xFFFFFFFFFFFFFFFF:#018x}", blob_checksum_equal=True)
        if info.rank != 0:
            model.adopt_packed(blob)
    elif stats is not None:
        stats.update(blob_checksum=f"{blob_checksum(blob) & 0xFFFFFFFFFFFFFFFF:#018x}", blob_checksum_equal=None)  # one rank: nothing to compare
    return model


def setup_generator_dp(gen, params_fn, info: Optional[RankInfo] = None, stats: Optional[dict] = None):
    """:func:`setup_model_dp` for the generator: ``params_fn()`` returns the Haiku parameter dict (rank 0 only)."""
    return setup_model_dp(gen, lambda g: g.load_params(params_fn()), info, stats)


def gather_to_rank0(local: torch.Tensor, info: Optional[RankInfo] = None) -> Optional[List[torch.Tensor]]:
    """Optional end-of-job gather for a single-writer CLI (not on the data path)."""
    info = info or rank_info()
    if info.world == 1:
        return [local]
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(info.world)]
    dist.all_gather(sizes, torch.tensor([local.numel()], dtype=torch.int64, device=local.device))
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local.reshape(-1)
    bufs = [torch.empty_like(pad) for _ in range(info.world)] if info.rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if info.rank != 0:
        return None
    return [b[: int(s.item())] for b, s in zip(bufs, sizes)]
