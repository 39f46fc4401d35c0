"""ctypes binding of libttsmi.so (the C ABI declared in include/ttsmi.h).

There is NO fallback: if the shared library is missing or a symbol is absent this raises, so a GPU
test can never silently pass on some other code path."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

# torch MUST be imported before libttsmi.so is dlopen'ed: the PyTorch-ROCm wheel bundles its own
# libamdhip64.so.7 and libttsmi must bind to THAT runtime instance (same SONAME -> the loader
# reuses the already-loaded copy), otherwise the process holds two HIP runtimes and the stream
# handles / device pointers torch hands us belong to the other one ("no ROCm-capable device").
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TTSMI_LIB (measurement only): another build of the same library, for same-box A/Bs of a source change
# (tools/kbench.py --variants base TTSMI_LIB=<path>).  Honoured only together with TTSMI_ALLOW_LIB_OVERRIDE=1 - the
# product and the tests always load the in-tree build - and any library, in-tree or not, must report the ABI
# version this binding was written for (EXPECTED_VERSION, checked in lib()).
_override = os.environ.get('TTSMI_LIB') if os.environ.get('TTSMI_ALLOW_LIB_OVERRIDE') == '1' else None
LIB_PATH = _override or os.path.join(_HERE, 'lib', 'libttsmi.so')
EXPECTED_VERSION = 109            # include/ttsmi.h: TTSMI_VERSION

P = c_void_p          # every device pointer
I = c_int
L = c_int64
F = c_float
S = c_void_p          # stream

# name -> (restype, argtypes); order and meaning exactly as in include/ttsmi.h
SIGNATURES = {
    'ttsmi_version': (I, []),
    'ttsmi_last_error': (c_char_p, []),
    'ttsmi_last_kernel': (c_char_p, []),
    'ttsmi_linear_fwd': (I, [P, L, P, L, I, P, L, P, P, L, I, I, I, I, I, S]),
    'ttsmi_linear_dgrad': (I, [P, L, P, L, P, L, P, L, I, I, I, I, I, S]),
    'ttsmi_linear_wgrad_ws_bytes': (c_size_t, [I, I, I]),
    'ttsmi_linear_wgrad': (I, [P, L, P, L, P, L, P, I, I, I, P, c_size_t, I, S]),
    'ttsmi_conv1d_fwd': (I, [P, P, P, P, I, I, I, I, I, I, I, S]),
    'ttsmi_conv1d_dgrad': (I, [P, P, P, P, I, I, I, I, I, I, S]),
    'ttsmi_conv1d_wgrad_ws_bytes': (c_size_t, [I, I, I, I, I]),
    'ttsmi_conv1d_wgrad': (I, [P, P, P, P, I, I, I, I, I, P, c_size_t, I, S]),
    'ttsmi_attention_fwd': (I, [P, P, P, P, P, I, I, I, I, F, c_uint64, P, c_uint32, I, S]),
    'ttsmi_attention_bwd_ws_bytes': (c_size_t, [I, I, I, I]),
    'ttsmi_attention_bwd': (I, [P, P, P, P, P, P, P, I, I, I, I, F, c_uint64, P, c_uint32, P,
                                c_size_t, I, S]),
    'ttsmi_attention_weights': (I, [P, P, P, P, I, I, I, I, F, c_uint64, P, c_uint32, I, S]),
    'ttsmi_attention_weights_masked': (I, [P, P, P, P, I, I, I, I, F, P, I, S]),
    'ttsmi_attention_dropmask_bytes': (c_size_t, [I, I, I]),
    'ttsmi_attention_dropmask': (I, [P, I, I, I, F, c_uint64, P, c_uint32, S]),
    'ttsmi_attention_dropmask_stack': (I, [P, P, I, I, I, I, F, c_uint64, P, S]),
    'ttsmi_attention_fwd_masked': (I, [P, P, P, P, P, I, I, I, I, F, P, I, S]),
    'ttsmi_attention_fwd_splitkeys_ws_bytes': (c_size_t, [I, I, I, I]),
    'ttsmi_attention_fwd_splitkeys': (I, [P, P, P, P, P, I, I, I, I, P, c_size_t, S]),
    'ttsmi_attention_bwd_masked': (I, [P, P, P, P, P, P, P, I, I, I, I, F, P, P, c_size_t, I, S]),
    'ttsmi_add_layernorm_fwd': (I, [P, P, P, P, P, P, I, P, F, c_uint32, F, c_uint32, c_uint64, P,
                                    F, P, P, P, I, I, P, S]),
    'ttsmi_add_layernorm_bwd_ws_bytes': (c_size_t, [I, I]),
    'ttsmi_add_layernorm_bwd': (I, [P, P, P, P, P, P, P, P, I, P, F, c_uint32, F, c_uint32,
                                    c_uint64, P, I, P, P, P, P, P, I, I, P, c_size_t, P, S]),
    'ttsmi_layernorm_param_reduce_batched': (I, [P, P, P, P, P, P, I, S]),
    'ttsmi_token_pad_mask': (I, [P, P, P, I, I, S]),
    'ttsmi_length_pad_mask': (I, [P, P, P, I, I, S]),
    'ttsmi_embedding_fwd': (I, [P, P, P, I, I, I, S]),
    'ttsmi_embedding_bwd': (I, [P, P, P, I, I, I, S]),
    'ttsmi_pitch_embed_fwd': (I, [P, P, P, P, P, I, I, S]),
    'ttsmi_pitch_embed_bwd_ws_bytes': (c_size_t, [I, I]),
    'ttsmi_pitch_embed_bwd': (I, [P, P, P, P, P, P, P, I, I, P, c_size_t, S]),
    'ttsmi_rowdot_fwd': (I, [P, P, P, P, P, I, I, I, S]),
    'ttsmi_rowdot_bwd_ws_bytes': (c_size_t, [I, I]),
    'ttsmi_rowdot_bwd': (I, [P, P, P, P, P, P, P, P, I, I, I, P, c_size_t, S]),
    'ttsmi_rowmask_mul': (I, [P, P, P, I, I, S]),
    'ttsmi_lenreg_index': (I, [P, I, P, P, P, I, I, I, S]),
    'ttsmi_lenreg_fwd': (I, [P, P, P, I, I, I, I, S]),
    'ttsmi_lenreg_bwd': (I, [P, P, P, I, I, I, I, S]),
    'ttsmi_l1_loss_ws_bytes': (c_size_t, [L]),
    'ttsmi_l1_loss': (I, [P, L, P, I, L, L, F, P, L, P, P, c_size_t, S]),
    'ttsmi_l1_losses_weighted_ws_bytes': (c_size_t, [I]),
    'ttsmi_l1_losses_weighted': (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, S]),
    'ttsmi_adam_tf': (I, [P, P, P, P, L, P, P, F, F, F, P, S]),
    'ttsmi_step_increment': (I, [P, S]),
    'ttsmi_stft_logmel': (I, [P, P, P, I, L, I, I, P, I, P, P, P, P, I, F, P, S]),
    'ttsmi_mel_nnls': (I, [P, P, P, P, P, P, I, P, I, I, I, F, I, F, S]),
    'ttsmi_griffinlim_ws_bytes': (c_size_t, [I]),
    'ttsmi_griffinlim': (I, [P, P, P, P, I, I, I, I, F, P, P, c_size_t, S]),
    'ttsmi_cast_f32_to_bf16': (I, [P, P, L, S]),
    'ttsmi_hgemm_tn': (I, [P, I, L, P, L, I, P, L, P, P, L, P, L, I, I, I, I, I, I, I, I, S]),
    'ttsmi_hgemm_k256_split': (I, [P, L, P, L, P, L, I, P, L, I, I, S]),
    'ttsmi_hgemm_wgrad_ws_bytes': (c_size_t, [I, I, I]),
    'ttsmi_hgemm_wgrad': (I, [P, P, L, P, L, P, I, I, I, P, c_size_t, S]),
    'ttsmi_cast_transpose_bf16': (I, [P, L, P, L, I, I, I, I, I, S]),
    'ttsmi_hgemm_wgrad_rows_ws_bytes': (c_size_t, [I, I, I]),
    'ttsmi_hgemm_wgrad_rows': (I, [P, I, L, P, I, L, P, L, P, I, I, I, I, I, I, I, P, c_size_t, S]),
    'ttsmi_conv_wdgrad_layout_bf16': (I, [P, P, I, I, I, I, S]),
    'ttsmi_cast_transpose_bf16_batched': (I, [P, I, I, S]),
    'ttsmi_hgemm_ln_fwd': (I, [P, L, P, L, I, P, L, P, P, P, P, P, F, c_uint32, c_uint64, P, F, P, P, P, P, I, I, I, S]),
    'ttsmi_layernorm_partials_bytes': (c_size_t, [I, I]),
    'ttsmi_hgemm_ln_bwd_nparts': (I, [I]),
    'ttsmi_hgemm_ln_bwd': (I, [P, L, P, L, P, P, P, P, P, F, c_uint32, c_uint64, P, P, P, P, c_size_t, I, I, I, S]),
    'ttsmi_relu_bits_bytes': (c_size_t, [I, I]),
    'ttsmi_hgemm_k256_relu_bits': (I, [P, L, P, L, P, P, L, P, I, I, S]),
    'ttsmi_hgemm_k256_masked_bits': (I, [P, L, P, L, P, P, L, I, I, S]),
    'ttsmi_hgemm_ln_fwd_h': (I, [P, L, P, L, I, P, L, P, P, P, P, P, F, c_uint32, c_uint64, P, F, P, P, P, P, I, I, I, S]),
    'ttsmi_hgemm_ln_bwd_dual_h': (I, [P, L, P, L, I, P, L, P, L, P, P, P, P, P, F, c_uint32, c_uint64, P, P, P, I, P, c_size_t,
                                      I, I, I, S]),
    'ttsmi_layernorm_bwd_xhat_h': (I, [P, P, P, P, P, F, c_uint32, c_uint64, P, P, P, P, c_size_t, I, I, S]),
    'ttsmi_hgemm_ln_bwd_dual': (I, [P, L, P, L, I, P, L, P, L, P, P, P, P, P, F, c_uint32, c_uint64, P, P, P, P, c_size_t,
                                    I, I, I, S]),
    'ttsmi_layernorm_bwd_xhat_nparts': (I, [I]),
    'ttsmi_layernorm_bwd_xhat': (I, [P, P, P, P, P, F, c_uint32, c_uint64, P, P, P, P, c_size_t, I, I, S]),
    'ttsmi_add_layernorm_bwd_nparts': (I, [I]),
    'ttsmi_layernorm_param_reduce_batched_nw': (I, [P, P, P, P, P, P, I, S]),
    'ttsmi_comm_unique_id': (I, [P]),
    'ttsmi_comm_init_rank': (I, [P, I, P, I]),
    'ttsmi_comm_destroy': (I, [P]),
    'ttsmi_allreduce_sum_f32': (I, [P, P, L, S]),
    'ttsmi_set_launch_observer': (I, [P]),
    'ttsmi_debug_stream_create_cu_mask': (I, [P, I, P]),
    'ttsmi_debug_xcc_census': (I, [P, I, S]),
    'ttsmi_dense_chain_pack_bytes': (c_size_t, [I, I]),
    'ttsmi_dense_chain_pack': (I, [P, P, P, P, I, P, c_size_t, S]),
    'ttsmi_dense_chain_supported': (I, [I, I, I]),
    'ttsmi_dense_chain_fwd': (I, [P, P, P, c_size_t, I, I, P, P, P, P, P, P, P, P, P, F, c_uint64, P, c_uint32, c_uint32, F,
                                  P, P, P, P, P, I, P, P, P, P, P, S]),
    'ttsmi_dense_chain_bwd_pack_bytes': (c_size_t, [I]),
    'ttsmi_dense_chain_bwd_supported': (I, [I, I, I]),
    'ttsmi_dense_chain_bwd_nparts': (I, [I]),
    'ttsmi_dense_block_bwd_chained': (I, [P]),
    'ttsmi_dense_chain_bwd_pack': (I, [P, P, P, I, P, c_size_t, S]),
    'ttsmi_dense_chain_pack_batched': (I, [P, I, S]),
    'ttsmi_dense_chain_bwd': (I, [P, P, P, P, P, P, P, P, c_size_t, I, I, F, c_uint64, P, c_uint32, P, P, P, I, P, P, c_size_t, S]),
    'ttsmi_dense_block_fwd': (I, [P, P, P]),
    'ttsmi_dense_block_bwd': (I, [P, P, P, P]),
    'ttsmi_dense_stack_fwd': (I, [P, I, P, P]),
    'ttsmi_dense_stack_bwd': (I, [P, I, P, P, P]),
    'ttsmi_ft_train_step': (I, [P, I]),
    'ttsmi_add2_f32': (I, [P, P, P, L, S]),
    'ttsmi_pad_cols_f32': (I, [P, I, P, I, I, S]),
}


def _dense_block_fields():
    i32, u32, u64, f, p = ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float, ctypes.c_void_p
    ptrs = ('step_dev', 'pad', 'klen', 'dropmask',
            'bqkv', 'bo', 'ln1_g', 'ln1_b', 'b1', 'b2', 'ln2_g', 'ln2_b',
            'wqkv_t', 'wo_t', 'w1_t', 'w2_t', 'wqkv_b', 'wo_b', 'w1_b', 'w2_b',
            'g_wqkv', 'g_bqkv', 'g_wo', 'g_bo', 'g_ln1_g', 'g_ln1_b', 'g_w1', 'g_b1', 'g_w2', 'g_b2', 'g_ln2_g', 'g_ln2_b',
            'qkv', 'cx', 'a_bf', 'h1', 'out_bf', 'lse', 'o', 'a', 'f', 'out', 'mean1', 'rstd1', 'mean2', 'rstd2')
    ptrs2 = ('df', 'dh1', 'd_o', 'dctx', 'dqkv', 'da', 'dh', 'attn_ws', 'ln_ws1', 'ln_ws2', 'wgrad_ws')
    return ([(n, i32) for n in ('B', 'H', 'T', 'd', 'F')] + [('rate', f)] +
            [(n, u32) for n in ('site_attn', 'site_ln1', 'site_ln2')] + [('seed', u64)] +
            [(n, p) for n in ptrs] + [('fuse_ln', i32), ('attn_split', i32)] +
            [(n, p) for n in ('xhat1', 'xhat2', 'lnp_ws1', 'lnp_ws2')] + [('lnp_ws1_bytes', u64), ('lnp_ws2_bytes', u64)] +
            [(n, p) for n in ptrs2] + [(n, u64) for n in ('attn_ws_bytes', 'ln_ws_bytes', 'wgrad_ws_bytes')] +
            [('main_stream', p), ('side_stream', p), ('ev', p * 4), ('below', p), ('ln2_done', i32), ('res16', i32), ('relu_bits', p),
             ('chain_w', p), ('chain_w_bytes', u64), ('above', p), ('qkv_done', i32), ('chain_pad_', i32),
             ('chain_bw', p), ('chain_bw_bytes', u64)])


class DenseBlockDesc(ctypes.Structure):
    """ctypes mirror of `ttsmi_dense_block` (include/ttsmi.h) - field order and types must match
    (tests/test_abi.py compares the size and a few offsets with the compiled header)."""
    _fields_ = _dense_block_fields()

FT_MAX_PRED_LAYERS, FT_MAX_BLOCKS, FT_EVENTS = 8, 32, 12


class FtPredLayer(ctypes.Structure):
    """ctypes mirror of `ttsmi_ft_pred_layer` (include/ttsmi.h; tests/test_abi.py compares every offset)."""
    _fields_ = ([(n, ctypes.c_int32) for n in ('k', 'Cin', 'Cout', 'Cout_pad')] + [('site', ctypes.c_uint32), ('pad_', ctypes.c_int32)] +
                [(n, ctypes.c_void_p) for n in ('w_t', 'w_d', 'bias', 'ln_g', 'ln_b', 'g_w', 'g_b', 'g_ln_g', 'g_ln_b', 'c', 'n', 'mean',
                                                'rstd', 'dc', 'dc_pad', 'dx', 'ln_ws')] + [('ln_ws_bytes', ctypes.c_uint64)] +
                [(n, ctypes.c_void_p) for n in ('xT', 'dyT', 'wg_ws')] + [('wg_ws_bytes', ctypes.c_uint64)])


class FtPredictor(ctypes.Structure):
    """ctypes mirror of `ttsmi_ft_predictor`."""
    _fields_ = ([('n_layers', ctypes.c_int32), ('relu_head', ctypes.c_int32), ('layer', FtPredLayer * FT_MAX_PRED_LAYERS)] +
                [(n, ctypes.c_void_p) for n in ('lin_w', 'lin_b', 'g_lin_w', 'g_lin_b', 'hm', 'y', 'dn', 'dbranch', 'rd_ws')] +
                [('rd_ws_bytes', ctypes.c_uint64)])


def _ft_step_fields():
    i32, u32, u64, i64, f, p = ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    P = lambda *names: [(n, p) for n in names]
    return ([(n, i32) for n in ('B', 'Tp', 'Tm', 'd', 'V', 'n_mel', 'n_enc', 'n_dec')] + [('rate', f), ('prate', f), ('seed', u64)] +
            P('step_dev') + [('site_enc_ln', u32), ('site_dec_ln', u32)] + P('main_stream', 'side_stream', 'wgrad_stream') +
            [('ev', p * FT_EVENTS)] + P('tokens', 'tgt_mel', 'tgt_dur', 'tgt_pitch', 'emb', 'g_emb',
                                        'enc_ln_g', 'enc_ln_b', 'enc_ps', 'g_enc_ln_g', 'g_enc_ln_b', 'g_enc_ps',
                                        'dec_ln_g', 'dec_ln_b', 'dec_ps', 'g_dec_ln_g', 'g_dec_ln_b', 'g_dec_ps',
                                        'pe_enc', 'pe_dec', 'pit_w', 'pit_b', 'g_pit_w', 'g_pit_b',
                                        'out_wt', 'out_wb', 'out_b', 'g_out_w', 'g_out_b') +
            [('enc', p * FT_MAX_BLOCKS), ('dec', p * FT_MAX_BLOCKS), ('dur', FtPredictor), ('pit', FtPredictor)] +
            P('pad_e', 'klen_e', 'pad_d', 'klen_d', 'x_emb', 'h0', 'h0_bf', 'mean0', 'rstd0', 'hp', 'idx', 'cum', 'lens',
              'x_dec', 'h1', 'h1_bf', 'mean1', 'rstd1', 'mel', 'loss_out', 'g_mel', 'g_dur', 'g_pit', 'loss_ws') +
            [('loss_ws_bytes', u64), ('loss_w', f * 3), ('pad2_', i32), ('loss_denom', i64 * 3)] +
            P('d_dec_out', 'd_x_dec', 'd_hp', 'd_branch', 'd_enc_out', 'd_x_emb', 'ln_ws0', 'ln_ws1') +
            [('ln_ws0_bytes', u64), ('ln_ws1_bytes', u64)] + P('pit_ws') + [('pit_ws_bytes', u64)] + P('wgrad_ws') +
            [('wgrad_ws_bytes', u64)] + P('p_flat', 'g_flat', 'm_flat', 'v_flat') + [('n_flat', i64)] + P('lr_dev', 'step_rw') +
            [('beta1', f), ('beta2', f), ('eps', f), ('pack_now', i32)] + P('flat_bf16', 'tr_desc') +
            [('tr_n', i32), ('tr_tiles', i32), ('n_conv_wd', i32), ('pack_ahead', i32),
             ('conv_w', p * (2 * FT_MAX_PRED_LAYERS)), ('conv_wd', p * (2 * FT_MAX_PRED_LAYERS)),
             ('conv_k', i32 * (2 * FT_MAX_PRED_LAYERS)), ('conv_cin', i32 * (2 * FT_MAX_PRED_LAYERS)),
             ('conv_cout', i32 * (2 * FT_MAX_PRED_LAYERS))])


class FtStep(ctypes.Structure):
    """ctypes mirror of `ttsmi_ft_step`."""
    _fields_ = _ft_step_fields()


TTSMI_F32, TTSMI_BF16, TTSMI_BF16_IO, TTSMI_BF16X3 = 0, 1, 2, 3
LAUNCH_OBSERVER = ctypes.CFUNCTYPE(None, c_int, c_char_p, ctypes.c_double, ctypes.c_double, c_void_p)

_lib = None


class TtsmiError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the bound library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TtsmiError(f'{LIB_PATH} is missing - run `python -c "import __graft_entry__ as g; '
                         f'g.build()"` (hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    l = ctypes.CDLL(LIB_PATH)
    try:
        l.ttsmi_version.restype = I
        got = int(l.ttsmi_version())
    except AttributeError as e:
        raise TtsmiError(f'{LIB_PATH} does not export ttsmi_version; rebuild it') from e
    if got != EXPECTED_VERSION:
        raise TtsmiError(f'{LIB_PATH} reports ABI version {got}, this binding needs {EXPECTED_VERSION}: a stale or variant '
                         f'build would be called with shifted arguments - rebuild it (python -m transformertts_amd.build)')
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise TtsmiError(f'libttsmi.so does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    _lib = _Traced(l)
    return _lib


_trace = None


def set_trace(callback) -> None:
    """Measurement hook (bench.py / profiling only): when set, every C-ABI call is made as
    callback(name, args, fn) -> rc, so a caller can bracket it with HIP events on the launch stream
    and attribute algorithmic FLOPs/bytes from the arguments.  None restores direct calls."""
    global _trace
    _trace = callback


class _Traced:
    """Attribute proxy over the ctypes library that routes calls through the trace hook if set."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._fns = {name: getattr(cdll, name) for name in SIGNATURES}

    def __getattr__(self, name):
        fn = self._fns[name]
        if _trace is None:
            return fn
        return lambda *args: _trace(name, args, fn)


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = lib().ttsmi_last_error()
        raise TtsmiError(f'{what or "libttsmi"} failed (code {rc}): '
                         f'{msg.decode() if msg else "unknown error"}')
