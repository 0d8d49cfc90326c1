"""world_size-2 and world_size-8 `gloo` tests of the N>1 path on CPU: contiguous slice sharding, the single end-of-path
all-gather (equal and ragged shards), max-over-ranks timing, and that sharded per-slice work is
identical to the unsharded result (the reference has no multi-device path)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_slice_work(x, slice_id0):
    """Stand-in for the per-slice sample: a deterministic function of (global slice id, slice data)."""
    ids = torch.arange(slice_id0, slice_id0 + x.shape[0], dtype=x.dtype).view(-1, 1, 1, 1)
    return torch.sin(x * (1.0 + ids)) + ids


def _oracle_slices(lo, hi):
    """The per-slice sample itself, restated by the CPU oracle (tiny UNet, proj-domain guided reverse process with
    adaptive guidance, 2+2 steps): slice s depends on (s, its data, draws keyed by s) only."""
    import numpy as np
    from oracle import diffusion as od, unet as ou
    from ipdm_pytorch_amd import synth
    from tests.golden.cases import LOOP_CFG
    cfg = ou.UNetConfig(**LOOP_CFG)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(ou.param_shapes(cfg), seed=41).items()}
    sch = od.Schedule(1000, 5)
    outs = []
    for sid in range(lo, hi):
        x = torch.from_numpy(synth.hash_uniform((1, 1, 24, 16), 600 + sid)) * 0.6
        k = [0]

        def noise_fn():
            z = torch.from_numpy(synth.hash_normal((1, 1, 24, 16), (700 + sid) * 1000 + k[0]))
            k[0] += 1
            return z
        res, _ = od.guided_reverse_process_slice(sch, lambda xx, t: ou.unet_forward(cfg, sd, xx, t), x, t_start=[2, 2],
                                                 clip=False, lambda_ratio=1, eta=0.5, mode="proj", constant_guidance=None,
                                                 noise_fn=noise_fn, kernel_size=4, amplitude=7)
        outs.append(res[-1])
    return torch.cat(outs, 0) if outs else torch.empty((0, 1, 24, 16))


def _worker_pipeline(rank, world, port, n_slices, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from ipdm_pytorch_amd import dist as idist
    r, w, _ = idist.init_from_env(backend="gloo")
    lo, hi = idist.shard_range(n_slices, r, w)
    out = idist.all_gather_slices(_oracle_slices(lo, hi), n_slices, r, w)
    idist.barrier()
    q.put((rank, out))
    torch.distributed.destroy_process_group()


def test_sharded_pipeline_output_world2():
    """The gathered result of two ranks, each running the per-slice sample (the oracle's restatement of it -- no GPU
    here) on its contiguous shard of 3 slices, equals the one-process result bit for bit: pipeline output through
    shard_range + the single all-gather, ragged shards (2 + 1)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.set_num_threads(2)
    want = _oracle_slices(0, 3)
    assert torch.equal(res[0], want) and torch.equal(res[1], want)


def _worker(rank, world, port, n_slices, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ipdm_pytorch_amd import dist as idist
    r, w, _ = idist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    full = torch.arange(n_slices * 6, dtype=torch.float32).view(n_slices, 1, 2, 3) / 7.0
    lo, hi = idist.shard_range(n_slices, r, w)
    local = _per_slice_work(full[lo:hi], lo)
    out = idist.all_gather_slices(local, n_slices, r, w)
    want = _per_slice_work(full, 0)
    t = idist.max_over_ranks(1.0 + rank, "cpu")
    per_rank = idist.gather_over_ranks(10.0 + rank, "cpu")          # what bench.py prints beside the max: every rank's step time
    assert per_rank == [10.0 + r for r in range(world)], per_rank
    # adaptive pass schedule (t_start_proj=None): every rank must take the branch of the GLOBAL Delta-map maximum
    import types
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
    hook = progressive_domain_denoiser._rank_max(types.SimpleNamespace(proj_device="cpu"))
    assert hook is not None and hook(3.0 if rank == 0 else 40.0) == 40.0
    assert hook(float((rank * 5) % world)) == float(max((r * 5) % world for r in range(world)))      # the maximum sits on a middle rank
    idist.barrier()
    q.put((rank, bool(torch.equal(out, want)), tuple(out.shape), t, (lo, hi)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_slices", [4, 5, 1])
def test_shard_and_all_gather_world2(n_slices):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = []
    for rank, same, shape, t, rng in res:
        assert same, "rank %d: gathered result differs from the unsharded one" % rank
        assert shape == (n_slices, 1, 2, 3)
        assert t == 2.0                       # max over ranks of (1.0, 2.0)
        ranges.append(rng)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n_slices


def test_shard_and_all_gather_world8_ragged_batch():
    """EIGHT ranks (the node the headline scales to: BASELINE config C5) over a ragged global batch of 61 slices -- five
    ranks with 8, three with 7: contiguous shards, the padded all-gather, per-rank timings, the MAX all-reduce of the timing
    and of the adaptive t_start=None branch (every rank takes the branch of the global Delta-map maximum)."""
    n_slices, world = 61, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_slices, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    for rank, same, shape, t, rng in res:
        assert same, "rank %d: gathered result differs from the unsharded one" % rank
        assert shape == (n_slices, 1, 2, 3) and t == float(world)
    sizes = [hi - lo for _, _, _, _, (lo, hi) in res]
    assert sizes == [8] * 5 + [7] * 3 and res[0][4][0] == 0 and res[-1][4][1] == n_slices
    for a, b in zip(res, res[1:]):
        assert a[4][1] == b[4][0]


def test_shard_range_partitions():
    from ipdm_pytorch_amd.dist import shard_range
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for a, b in zip(cuts, cuts[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
