"""bench.py's instrumented step must know every C-ABI entry point the product calls WITHOUT a stream as its last
argument: the hook treats a trailing integer as the launch stream (round 4: the new stack launchers ended in a device
pointer and the default bench line crashed in its roofline leg - caught by the profile session, not by a test)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_every_streamless_entry_point_is_passed_through_by_the_hook():
    src = open(os.path.join(ROOT, 'include', 'ttsmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    hook = bench[bench.index('def hook(name, args, fn):'):bench.index('# the launch stream is the entry point')]
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(ttsmi_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        name, params = m.group(1), m.group(2)
        if 'ttsmi_stream_t' in params.split(',')[-1]:
            continue                                      # launches on an explicit stream: bracketed with events
        passed = (name.endswith('_bytes') or name.endswith('_nparts') or name.endswith('_supported') or 'comm' in name
                  or f"'{name}'" in hook)
        assert passed, f'{name} has no stream argument and bench.py:instrumented_step does not pass it through'


def test_attention_valu_roof_arithmetic():
    """bench.py's vector-ALU roof of the attention families: scores = FLOPs / (4 or 8 x dh); one wave64 instruction row
    covers 64 scores; 1 024 SIMDs at 2.3 GHz.  A kernel that took exactly the roof time has fraction 1."""
    import bench
    B, H, T, dh = 32, 4, 900, 64
    fwd_flops = 4.0 * B * H * T * T * dh
    r = bench.attention_valu_roof(fwd_flops, 1.0, dh, 'fwd')
    scores = B * H * T * T
    want_ms = scores / 64 * bench.ATTN_VALU_CYCLES_PER_SCORE_ROW['fwd'] / bench.SIMDS / (bench.SIMD_CLOCK_GHZ * 1e9) * 1e3
    assert abs(r['valu_roof_ms'] - want_ms) < 1e-12 and abs(r['valu_roof_frac'] - want_ms) < 1e-12
    assert abs(bench.attention_valu_roof(fwd_flops, want_ms, dh, 'fwd')['valu_roof_frac'] - 1.0) < 1e-12
    b = bench.attention_valu_roof(2 * fwd_flops, 1.0, dh, 'bwd')                 # backward: 8 T^2 dh for the same scores
    assert abs(b['valu_roof_ms'] / r['valu_roof_ms'] -
               bench.ATTN_VALU_CYCLES_PER_SCORE_ROW['bwd'] / bench.ATTN_VALU_CYCLES_PER_SCORE_ROW['fwd']) < 1e-12
    assert 0.017 < r["valu_roof_ms"] < 0.020                                     # ~18.5 us for one decoder layer (measured: 56-59 us)


def test_attention_valu_roofs_are_added_to_a_per_kernel_table_and_never_raise():
    import bench
    cfg, _ = bench.workload_config('configs[1]')
    table = {bench.HATTN_FWD: {'launches': 12, 'gflop': 170.0, 'algorithmic_mb': 1.0, 'ms': 0.44, 'tflops': 386.0, 'gbs': 2.0},
             bench.HATTN_BWD: {'launches': 12, 'gflop': 340.0, 'algorithmic_mb': 1.0, 'ms': 1.36, 'tflops': 250.0, 'gbs': 2.0},
             bench.RIDERS: {'launches': 3, 'gflop': 0.0, 'algorithmic_mb': 1.0, 'ms': 0.1, 'tflops': None, 'gbs': 10.0}}
    bench.add_attention_valu_roofs(table, cfg)
    assert 0.1 < table[bench.HATTN_FWD]['valu_roof_frac'] < 1.0 and 0.1 < table[bench.HATTN_BWD]['valu_roof_frac'] < 1.0
    assert 'valu_roof_frac' not in table[bench.RIDERS]
    bench.add_attention_valu_roofs({}, cfg)                                  # nothing to add
    bench.add_attention_valu_roofs({bench.HATTN_FWD: {'ms': 0.0}}, cfg)      # incomplete entry
    bench.add_attention_valu_roofs(table, {})                                # no architecture keys
    ref_cfg, _ = bench.workload_config('ref-default')
    t2 = {bench.HATTN_FWD: {'gflop': 1.0, 'ms': 1.0}}
    bench.add_attention_valu_roofs(t2, ref_cfg)                              # dh = 192: the counts are not its kernels'
    assert 'valu_roof_frac' not in t2[bench.HATTN_FWD]


def test_every_launch_group_the_block_launcher_announces_has_a_family_in_the_bench_table():
    """The C++ block launcher names its launches to the profiling observer (OBS("...") in csrc/dense_block.hip); bench.py sorts
    them into kernel families by that name (OBSERVED) and everything it does not know lands in the riders' bucket.  A new
    launch group (round 5: the two chain entry points) must be sorted on purpose: either a family, or the explicit list of
    HBM-bound riders here.  The PMC table must know every family that has kernels of its own."""
    import bench
    src = open(os.path.join(ROOT, 'transformertts_amd', 'csrc', 'dense_block.hip')).read()
    announced = set(re.findall(r'OBS\("([a-z0-9_]+)"', src))
    assert {'ttsmi_dense_chain_fwd', 'ttsmi_dense_chain_bwd', 'ttsmi_attention_bwd', 'ttsmi_hgemm_ln_bwd'} <= announced
    riders_on_purpose = {'ttsmi_layernorm_bwd_xhat'}
    unsorted = announced - set(bench.OBSERVED) - riders_on_purpose
    assert not unsorted, f'launch groups without a kernel family in bench.OBSERVED: {sorted(unsorted)}'
    assert bench.OBSERVED['ttsmi_dense_chain_fwd'] == bench.OBSERVED['ttsmi_dense_chain_bwd'] == bench.CHAIN
    for fam in set(bench.OBSERVED.values()):
        assert fam in bench.PMC_KERNELS, f'no rocprof kernel names for the family {fam[:40]}...'
    names, helpers = bench.PMC_KERNELS[bench.CHAIN]
    assert 'dense_chain16_bwd_kernel' in names and 'dense_chain16_pack_kernel' in helpers
