"""world_size-2 tests of the data-parallel path: `gloo` on CPU (the model itself needs a GPU, so the
per-rank gradients come from the CPU oracle here; what is under test is transformertts_amd/dp.py:
sharding, the single flat-buffer all-reduce, parameter broadcast, and the claim that averaging the
per-rank gradients of equal-shape shards equals the gradient of the global-batch loss)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, args, nprocs, seconds=240):
    """mp.spawn with a deadline: a collective that deadlocks must fail the test, not hang the suite (the workers are
    killed by PID)."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + seconds
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail(f'{fn.__name__}: {nprocs} ranks did not finish within {seconds} s')


def _flat(grads):
    return torch.cat([g.reshape(-1) for g in grads.values()])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import ft_oracle as fo
    from transformertts_amd import dp
    r, local, w = dp.init_process_group(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    batch = fo.synthetic_batch(4, 16, 48, seed=3)                 # global batch 4, equal shapes
    shard = dp.shard_batch(batch, rank, world)
    assert shard[0].shape[0] == 2
    model = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    g = _flat(model.train_step(*shard, apply=False)['grads']).clone()
    sync = dp.GradAllReduce()
    assert sync.world == 2 and not sync.use_avg                   # gloo: SUM then 1/world
    g2 = g.clone()
    sync(g)
    # two-bucket path: the decoder half is launched early (as the backward hook does), the head and the
    # join follow - same result as the single all-reduce
    assert sync.overlap
    split = g2.numel() // 3
    sync.start_tail(g2, split)
    assert sync._tail is not None
    g2[:split] += 0.0                                             # "encoder backward" still writing the head
    sync(g2)
    assert sync._tail is None
    torch.testing.assert_close(g2, g, rtol=0, atol=0)
    # broadcast: rank 1 starts from garbage and must end with rank 0's parameters
    params = torch.arange(10, dtype=torch.float32) if rank == 0 else torch.full((10,), -1.0)
    dp.broadcast_parameters(params, 0)
    if rank == 0:
        np.save(os.path.join(out_dir, 'avg.npy'), g.numpy())
    np.save(os.path.join(out_dir, f'params{rank}.npy'), params.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_equals_global_batch_gradient(tmp_path):
    from oracle import ft_oracle as fo
    port = _free_port()
    _spawn(_worker, (2, port, str(tmp_path)), 2)
    avg = np.load(tmp_path / 'avg.npy')
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    batch = fo.synthetic_batch(4, 16, 48, seed=3)
    model = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    want = _flat(model.train_step(*batch, apply=False)['grads']).numpy()
    np.testing.assert_allclose(avg, want, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(np.load(tmp_path / 'params1.npy'), np.arange(10, dtype=np.float32))


def test_shard_batch_rejects_uneven_split():
    from transformertts_amd import dp
    with pytest.raises(AssertionError):
        dp.shard_batch([np.zeros((5, 3))], 0, 2)
    a, = dp.shard_batch([np.arange(8).reshape(4, 2)], 1, 2)
    np.testing.assert_array_equal(a, [[4, 5], [6, 7]])


def test_world_size_one_is_degenerate():
    from transformertts_amd import dp
    sync = dp.GradAllReduce()
    g = torch.ones(4)
    sync(g)
    assert torch.equal(g, torch.ones(4))


def _gpu_worker(rank, world, port, out_dir, backend='gloo'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), TTSMI_DIST_BACKEND=backend,
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    from oracle import ft_oracle as fo
    from transformertts_amd import dp
    from transformertts_amd.model.models import ForwardTransformer
    r, local, w = dp.init_process_group()
    assert dist.get_backend() == backend and w == world
    dev = rank if backend == 'nccl' else 0                    # gloo: both ranks share the one GPU of the test box
    torch.cuda.set_device(dev)
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    model = ForwardTransformer.from_config(dict(cfg, device=f'cuda:{dev}', seed=100 + rank, precision='bf16'))
    if rank == 0:
        model.load_weights_dict({k: np.asarray(v) for k, v in W.items()})   # rank 1 keeps its own random init
    model._compile(learning_rate=1e-3)
    wrapped = dp.DataParallel(model)                          # broadcast must overwrite rank 1's weights
    assert wrapped.sync.overlap and model.grad_sync is not None
    batch = fo.synthetic_batch(4, 16, 48, seed=3)
    shard = dp.shard_batch(batch, rank, world)
    for _ in range(2):
        wrapped.train_step(*shard)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f'w{rank}.npy'), model.params.data.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_two_ranks_match_the_global_batch_step(tmp_path, backend):
    """The whole DP path with the real model - two ranks sharing the GPU over gloo, and (on a box with >= 2 GPUs,
    skipped otherwise) one rank per GPU over RCCL (backend "nccl"): parameter broadcast, the backward hook, the
    overlapped two-bucket gradient all-reduce and the replicated Adam step must leave both ranks with identical
    weights, equal to a single process stepping on the global batch."""
    from oracle import ft_oracle as fo
    from transformertts_amd.model.models import ForwardTransformer
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one GPU per rank: fewer than 2 GPUs visible')
    port = _free_port()
    _spawn(_gpu_worker, (2, port, str(tmp_path), backend), 2)
    w0, w1 = np.load(tmp_path / 'w0.npy'), np.load(tmp_path / 'w1.npy')
    np.testing.assert_array_equal(w0, w1)
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    single = ForwardTransformer.from_config(dict(cfg, device='cuda:0', seed=0, precision='bf16'))
    single.load_weights_dict({k: np.asarray(v) for k, v in W.items()})
    single._compile(learning_rate=1e-3)
    batch = fo.synthetic_batch(4, 16, 48, seed=3)
    for _ in range(2):
        single.train_step(*batch)
    ws = single.params.data.cpu().numpy()
    # same math up to the summation order of the two half-batch gradients (bf16 GEMM operands): Adam's
    # lr * sign-like step bounds the difference by a few lr
    assert np.abs(w0 - ws).max() < 5e-3 and np.abs(w0 - ws).mean() < 2e-4
