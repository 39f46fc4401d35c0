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


# ---------------------------------------------------------------------------------------------------------------------
# Ragged global batches: ranks with DIFFERENT local maxima and DIFFERENT local batch sizes must
# reproduce the single-device step on the concatenated, globally padded batch - the reference's loss is an unmasked mean
# over [B, T_max, C] (utils/losses.py:41-49), so padded positions carry loss and gradient and the divisor is global.
# ---------------------------------------------------------------------------------------------------------------------
RAGGED = dict(B=4, Tp=16, Tm=48, seed=5)


def _ragged_shards():
    """Global ragged batch of 4 (sample 0 at both maxima) -> rank 0: samples 1..3 trimmed to THEIR maxima,
    rank 1: sample 0 alone (1 sample, global maxima)."""
    from oracle import ft_oracle as fo
    tokens, mel, durs, pitch = fo.synthetic_batch(RAGGED['B'], RAGGED['Tp'], RAGGED['Tm'], seed=RAGGED['seed'], ragged=True)

    def trim(sl):
        tp = int((tokens[sl] != 0).sum(1).max())
        tm = int(durs[sl].sum(1).max())
        return tokens[sl, :tp], mel[sl, :tm], durs[sl, :tp], pitch[sl, :tp]
    shards = [trim(slice(1, 4)), trim(slice(0, 1))]
    assert shards[0][0].shape[1] < RAGGED['Tp'] and shards[0][1].shape[1] < RAGGED['Tm']
    # the single-device batch in rank order
    order = [1, 2, 3, 0]
    return (tokens, mel, durs, pitch), shards, tuple(a[order] for a in (tokens, mel, durs, pitch))


def _ragged_cpu_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import torch.nn.functional as F
    from oracle import ft_oracle as fo
    from transformertts_amd import dp
    dp.init_process_group(backend='gloo')
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    _, shards, _ = _ragged_shards()
    x, ts, td, tp = shards[rank]

    class _Stub:                      # DataParallel only needs these of the model on the CPU
        device = 'cpu'
        params = type('P', (), {'data': torch.zeros(4), 'offsets': {}})()
        grad_sync = None
    wrapped = dp.DataParallel(_Stub(), broadcast=False)
    Bg, Tpg, Tmg = wrapped.global_shape(x.shape[0], x.shape[1], ts.shape[1])
    assert (Bg, Tpg, Tmg) == (RAGGED['B'], RAGGED['Tp'], RAGGED['Tm'])
    x, ts, td, tp = (wrapped._pad_to(a, n) for a, n in ((x, Tpg), (ts, Tmg), (td, Tpg), (tp, Tpg)))
    # the oracle's Expand stops at the shard's own max sum(dur): pad it to the global length like the model does
    real = fo.expand_torch
    fo.expand_torch = lambda h, d: F.pad(real(h, d), (0, 0, 0, Tmg - real(h, d).shape[1]))
    model = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    tr = model.train_step(x, ts, td, tp, apply=False)
    share = x.shape[0] / Bg                                   # local mean * B_local / B_global = local sum / global count
    g = _flat(tr['grads']) * share
    wrapped.sync.scale = False                                # what DataParallel.train_step sets: plain SUM
    wrapped.sync(g)
    loss = torch.tensor([float(tr['loss']) * share], dtype=torch.float64)
    dist.all_reduce(loss)
    if rank == 0:
        np.save(os.path.join(out_dir, 'sum.npy'), g.numpy())
        np.save(os.path.join(out_dir, 'loss.npy'), loss.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_shards_with_global_counts_equal_the_single_device_step(tmp_path):
    """gloo world 2 on the CPU oracle: shards of 3 and 1 samples, trimmed to their own maxima, go through
    DataParallel.global_shape / _pad_to and the un-scaled SUM all-reduce; the result is the gradient and the loss of
    ONE step on the whole padded batch (reference model/models.py:464-482)."""
    from oracle import ft_oracle as fo
    port = _free_port()
    _spawn(_ragged_cpu_worker, (2, port, str(tmp_path)), 2)
    _, _, whole = _ragged_shards()
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    tr = fo.ForwardTransformerOracle(cfg, W, torch.float64).train_step(*whole, apply=False)
    np.testing.assert_allclose(np.load(tmp_path / 'sum.npy'), _flat(tr['grads']).numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.load(tmp_path / 'loss.npy')[0], float(tr['loss']), rtol=1e-12)


def _ragged_gpu_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TTSMI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from oracle import ft_oracle as fo
    from transformertts_amd import dp
    from transformertts_amd.model.models import ForwardTransformer
    dp.init_process_group()
    torch.cuda.set_device(0)
    cfg = dict(fo.tiny_config(), dropout_rate=0.0, predictors_dropout=0.0)     # ranks draw their own dropout streams
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    model = ForwardTransformer.from_config(dict(cfg, device='cuda:0', seed=100 + rank, precision='f32'))
    model.load_weights_dict({k: np.asarray(v) for k, v in W.items()})
    model._compile(learning_rate=1e-3)
    wrapped = dp.DataParallel(model)
    _, shards, _ = _ragged_shards()
    out = wrapped.train_step(*shards[rank])                   # global shape found by the all-gather
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f'w{rank}.npy'), model.params.data.cpu().numpy())
    np.save(os.path.join(out_dir, f'g{rank}.npy'), model.params.grad.cpu().numpy())
    np.save(os.path.join(out_dir, f'loss{rank}.npy'), np.array([float(out['loss']), float(out['losses']['mel']),
                                                                float(out['losses']['duration']), float(out['losses']['pitch'])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ragged_two_rank_step_on_the_gpu_equals_the_single_process_step(tmp_path):
    """The real model (exact-fp32 path, dropout off) as two gloo ranks on one GPU with ragged shards (3 samples with
    shorter maxima / 1 sample at the global maxima) against ONE process stepping on the whole padded batch: gradients,
    the four loss values and the post-Adam weights."""
    from oracle import ft_oracle as fo
    from transformertts_amd.model.models import ForwardTransformer
    port = _free_port()
    _spawn(_ragged_gpu_worker, (2, port, str(tmp_path)), 2)
    w0, w1 = np.load(tmp_path / 'w0.npy'), np.load(tmp_path / 'w1.npy')
    np.testing.assert_array_equal(w0, w1)
    np.testing.assert_array_equal(np.load(tmp_path / 'loss0.npy'), np.load(tmp_path / 'loss1.npy'))
    _, _, whole = _ragged_shards()
    cfg = dict(fo.tiny_config(), dropout_rate=0.0, predictors_dropout=0.0)
    W = fo.init_weights(cfg, seed=1, perturb=0.02)
    single = ForwardTransformer.from_config(dict(cfg, device='cuda:0', seed=0, precision='f32'))
    single.load_weights_dict({k: np.asarray(v) for k, v in W.items()})
    single._compile(learning_rate=1e-3)
    out = single.train_step(*whole)
    torch.cuda.synchronize()
    want = np.array([float(out['loss']), float(out['losses']['mel']), float(out['losses']['duration']), float(out['losses']['pitch'])])
    np.testing.assert_allclose(np.load(tmp_path / 'loss0.npy'), want, rtol=2e-6)
    gs = single.params.grad.cpu().numpy()
    g0 = np.load(tmp_path / 'g0.npy')
    assert np.abs(g0 - gs).max() < 2e-5 * np.abs(gs).max()           # fp32 summation order of the two shards only
    # Adam turns a gradient into a step of about lr whatever its size, so weights can differ by ~2 lr where the
    # gradient is at the noise floor; everywhere else they agree closely
    ws = single.params.data.cpu().numpy()
    assert np.abs(w0 - ws).max() < 2.5e-3 and np.abs(w0 - ws).mean() < 2e-5


# ---------------------------------------------------------------------------------------------------------------------
# The collective behind the C ABI (ttsmi_comm_* / ttsmi_allreduce_sum_f32: a thin RCCL wrapper for bindings without
# torch.distributed).  One rank on a 1-GPU box (sum over one rank = identity, but the call goes through RCCL's stream
# ordering); one rank per GPU when >= 2 are visible.
# ---------------------------------------------------------------------------------------------------------------------
def _abi_rank(rank, world, id_bytes, out_dir):
    import ctypes
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    from transformertts_amd import _lib
    from transformertts_amd._lib import check
    torch.cuda.set_device(rank)
    l = _lib.lib()
    comm = ctypes.c_void_p()
    idbuf = ctypes.create_string_buffer(bytes(id_bytes), 128)
    check(l.ttsmi_comm_init_rank(ctypes.byref(comm), world, idbuf, rank), 'comm_init_rank')
    g = torch.arange(1 << 20, dtype=torch.float32, device=f'cuda:{rank}') * (rank + 1)
    st = torch.cuda.current_stream().cuda_stream
    check(l.ttsmi_allreduce_sum_f32(comm, g.data_ptr(), g.numel(), st), 'allreduce')
    g2 = g * 2                                   # consumer on the same stream: ordered after the collective
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f'abi{rank}.npy'), g2[:4096].cpu().numpy())
    check(l.ttsmi_comm_destroy(comm), 'comm_destroy')


@pytest.mark.gpu
@pytest.mark.parametrize('world', [1, 2])
def test_c_abi_allreduce(tmp_path, world):
    import ctypes
    from transformertts_amd import _lib
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs')
    l = _lib.lib()
    idbuf = ctypes.create_string_buffer(128)
    rc = l.ttsmi_comm_unique_id(idbuf)
    if rc == -3:
        pytest.skip('librccl not loadable here: ' + l.ttsmi_last_error().decode())
    assert rc == 0, l.ttsmi_last_error()
    if world == 1:
        _abi_rank(0, 1, idbuf.raw, str(tmp_path))
    else:
        _spawn(_abi_rank, (world, idbuf.raw, str(tmp_path)), world)
    want = np.arange(4096, dtype=np.float32) * sum(r + 1 for r in range(world)) * 2
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f'abi{r}.npy'), want)
    # argument checks
    assert l.ttsmi_allreduce_sum_f32(None, None, 4, None) == -1 and b'bad argument' in l.ttsmi_last_error()
    assert l.ttsmi_comm_init_rank(None, 1, idbuf, 0) == -1
