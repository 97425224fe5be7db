"""The N>1 path on CPU: pure partitioning functions, and a world_size-2 gloo run of the one collective
the design has (the start-up weight broadcast) plus the optional end-of-job gather."""
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from viettts_amd.dist import (HALO_FRAMES, broadcast_packed_weights, gather_to_rank0, plan_chunks, shard_chunks,
                              shard_utterances)


def test_shard_utterances_partition_and_balance():
    rng = np.random.default_rng(0)
    lengths = rng.integers(50, 2000, size=257).tolist()
    for world in (1, 2, 4, 8):
        shards = shard_utterances(lengths, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))  # every utterance exactly once
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lengths)  # greedy LPT bound
        assert shards == shard_utterances(lengths, world)  # deterministic: every rank computes the same plan
    assert shard_utterances([], 4) == [[], [], [], []]
    with pytest.raises(ValueError):
        shard_utterances([1, 2], 0)


@pytest.mark.parametrize("T,chunk", [(1, 512), (512, 512), (513, 512), (4096, 512), (37500, 512), (100, 7)])
def test_plan_chunks_cover_exactly(T, chunk):
    chunks = plan_chunks(T, chunk)
    assert chunks[0].t0 == 0 and chunks[-1].t1 == T
    for a, b in zip(chunks, chunks[1:]):
        assert a.t1 == b.t0  # kept ranges tile [0, T) with no gap / overlap
    for c in chunks:
        assert 0 <= c.lo <= c.t0 < c.t1 <= c.hi <= T
        assert c.t0 - c.lo == min(HALO_FRAMES, c.t0)  # full halo inside, none at the true edge
        assert c.hi - c.t1 == min(HALO_FRAMES, T - c.t1)
        assert c.keep_from == c.t0 - c.lo
    for world in (1, 2, 8):
        sh = shard_chunks(chunks, world)
        assert sorted(c.index for s in sh for c in s) == [c.index for c in chunks]
        for r, s in enumerate(sh):
            assert all(c.index % world == r for c in s)


def test_blob_checksum_sees_flips_swaps_and_length():
    from viettts_amd.dist import blob_checksum, verify_blob_on_all_ranks

    g = torch.Generator().manual_seed(3)
    a = torch.randint(0, 256, (4099,), dtype=torch.uint8, generator=g)  # not a multiple of 8: the tail is zero-padded
    base = blob_checksum(a)
    assert base == blob_checksum(a.clone()) == verify_blob_on_all_ranks(a)  # one process: nothing to compare, the checksum itself
    b = a.clone()
    b[4098] ^= 1
    assert blob_checksum(b) != base
    c = a.clone()
    c[0:8], c[8:16] = a[8:16].clone(), a[0:8].clone()  # two whole words swapped: a plain sum would not notice
    assert blob_checksum(c) != base
    assert blob_checksum(torch.cat([a, torch.ones(8, dtype=torch.uint8)])) != base


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from viettts_amd import dist as vdist

    info = vdist.init_process_group("gloo")
    assert (info.rank, info.world) == (rank, world)
    # rank 0 "packs" the weights; every other rank allocates an empty blob of the same size and receives it
    n = 1 << 16
    ref = (torch.arange(n, dtype=torch.int64) * 2654435761 % 251).to(torch.uint8)
    blob = ref.clone() if rank == 0 else torch.zeros(n, dtype=torch.uint8)
    broadcast_packed_weights(blob, 0)
    ok_bcast = bool(torch.equal(blob, ref))
    # independent shards, no collective on the data path: each rank "synthesises" its own utterances
    lengths = [5, 9, 3, 7, 2]
    mine = shard_utterances(lengths, world)[rank]
    local = torch.cat([torch.full((lengths[i],), float(i)) for i in mine]) if mine else torch.zeros(0)
    got = gather_to_rank0(local, info)
    if rank == 0:
        plan = shard_utterances(lengths, world)
        ok_gather = all(torch.equal(got[r], torch.cat([torch.full((lengths[i],), float(i)) for i in plan[r]])) for r in range(world))
    else:
        ok_gather = got is None
    # the start-up protocol itself (viettts_amd.dist.setup_model_dp) on a stand-in model: rank 0 loads and packs, rank 1
    # must never call load_params and ends up with rank 0's bytes
    class FakeModel:
        device = torch.device("cpu")
        packed_bytes = 4096

        def __init__(self):
            self._blob, self.loaded, self.adopted = None, False, False

        def load_params(self, seed):
            self.loaded = True
            self._blob = (torch.arange(4096, dtype=torch.int64) * seed % 253).to(torch.uint8)

        def packed_blob(self):
            return self._blob

        def adopt_packed(self, blob):
            self.adopted, self._blob = True, blob

    m = vdist.setup_model_dp(FakeModel(), lambda mm: mm.load_params(7 + rank), info)  # a rank-dependent seed: only rank 0's may win
    want = (torch.arange(4096, dtype=torch.int64) * 7 % 253).to(torch.uint8)
    ok_bcast = ok_bcast and bool(torch.equal(m._blob, want)) and m.loaded == (rank == 0) and m.adopted == (rank != 0)
    # every rank checksums the blob it ended up with and the checksums are all-reduced (MIN, MAX): equal here ...
    st = {}
    m2 = vdist.setup_model_dp(FakeModel(), lambda mm: mm.load_params(11), info, st)
    ok_sum = st.get("blob_checksum_equal") is True and st.get("backend") == "gloo" and st["blob_checksum"] == f"{vdist.blob_checksum(m2._blob) & 0xFFFFFFFFFFFFFFFF:#018x}"
    # ... and a transfer that damages ONE byte on ONE rank stops EVERY rank at start-up
    real_bcast = vdist.broadcast_packed_weights

    def damaged_bcast(blob, src=0):
        real_bcast(blob, src)
        if rank == 1:
            blob[1234] ^= 0x10
        return blob

    vdist.broadcast_packed_weights = damaged_bcast
    try:
        vdist.setup_model_dp(FakeModel(), lambda mm: mm.load_params(13), info)
        ok_sum = False
    except RuntimeError as e:
        ok_sum = ok_sum and "different blobs" in str(e)
    finally:
        vdist.broadcast_packed_weights = real_bcast
    q.put((rank, ok_bcast and ok_sum, ok_gather))
    dist.barrier()
    dist.destroy_process_group()


def test_generator_batches_cover_every_sentence_once_and_balance_by_frames():
    """viettts_amd/pipeline.py::_generator_batches — the ragged batches the pipeline hands the generator: every row exactly once, in
    ascending-length order, as few passes as PASS_FRAMES of real frames allow (but two for a large job that would fit one), the passes balanced by
    frames; gen_batch > 0 caps the count."""
    import random

    from viettts_amd.pipeline import PASS_FRAMES, SPLIT_MIN_FRAMES, TAIL_SHARE, _generator_batches

    rnd = random.Random(5)
    for n in (1, 7, 256, 1024, 3000):
        fr = sorted(rnd.randint(40, 300) for _ in range(n))
        rows = list(range(100, 100 + n))
        out = _generator_batches(rows, fr)
        assert [r for b in out for r in b] == rows and all(out)
        sums = [sum(fr[r - 100] for r in b) for b in out]
        k = max(1, -(-sum(fr) // int(PASS_FRAMES * 1.25)))
        if k == 1 and n >= 32 and sum(fr) >= SPLIT_MIN_FRAMES:
            # a large job that fits one pass leaves in two (the first pass's read-back under the second's compute), the tail holding TAIL_SHARE of the frames
            assert len(out) == 2 and abs(sums[1] - TAIL_SHARE * sum(fr)) <= 300
        else:
            assert len(out) == k
            assert len(out) == 1 or max(sums) - min(sums) <= 2 * 300
        assert max(sums) <= PASS_FRAMES * 1.25 + 300
        capped = _generator_batches(rows, fr, 64)
        assert [r for b in capped for r in b] == rows and max(len(b) for b in capped) <= 64
    assert _generator_batches([], []) == []


def test_generator_batches_bound_the_padded_size():
    """a long-tailed shard (500 sentences of ~100 frames and one of 1000) must not become one pass of 500 000 padded frames."""
    from viettts_amd.pipeline import PASS_FRAMES, _generator_batches

    fr = sorted([100] * 500 + [1000])
    out = _generator_batches(list(range(501)), fr)
    assert [r for b in out for r in b] == list(range(501))
    assert all(len(b) * fr[b[-1]] <= 2 * PASS_FRAMES for b in out) and len(out) == 2


def test_gloo_world2_broadcast_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_bench_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with
    N ranks on 127.0.0.1 (the driver's own launch line); with WORLD_SIZE set it does not, and a rank count that disagrees
    with --gpus is an error (`--gpus 8` used to measure ONE GPU with a warning)."""
    import sys

    import bench

    seen = {}

    def fake_execve(exe, argv, env):
        seen["exe"], seen["argv"], seen["env"] = exe, list(argv), dict(env)
        raise SystemExit(0)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(bench.os, "execve", fake_execve)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and "--nproc-per-node=4" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and a[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # the ranks' RCCL needs the dmabuf IPC mode
    # a launcher that started the wrong number of ranks is refused before any GPU work
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not seen and "WORLD_SIZE=1" in str(e.value)


def test_launch_env_and_ipc_mode_and_missing_port(monkeypatch):
    """The N > 1 launch contract: every launcher hands its ranks HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL's HIP-IPC hand-off of device
    buffers needs the dmabuf mode on these hosts; a value the caller exported wins), the rendezvous port is ONE free port the parent picked — never
    a fixed default — and a rank that was launched without one fails with a message instead of meeting nobody."""
    from viettts_amd import dist as vdist

    port = vdist.free_port()
    assert 1024 < port < 65536
    e0, e1 = (vdist.launch_env(8, r, port=port, env={"PATH": "/bin"}) for r in (0, 7))
    assert e0["MASTER_PORT"] == e1["MASTER_PORT"] == str(port) and e0["MASTER_ADDR"] == "127.0.0.1"
    assert (e0["RANK"], e1["RANK"], e1["LOCAL_RANK"], e1["WORLD_SIZE"]) == ("0", "7", "7", "8")
    assert e0["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert vdist.launch_env(2, 0, port=port, env={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"  # the caller's choice stands
    with pytest.raises(ValueError):
        vdist.launch_env(2, 0)  # no port: every rank would pick its own
    for k in ("MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        vdist.init_process_group("gloo")


def test_bench_sets_the_ipc_mode_before_importing_torch():
    """bench.py is what the driver launches under torch.distributed.run: the switch has to be in the environment before the HIP runtime starts."""
    src = (Path(__file__).resolve().parents[1] / "bench.py").read_text()
    assert 0 < src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < src.index("import torch")
