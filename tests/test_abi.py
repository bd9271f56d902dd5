"""The drop-in boundary: every entry point include/tnt_hip.h declares is exported by the built library
(masters-thesis_amd/csrc/libtnt_hip.so), bound in masters_thesis_amd/_lib.py with the declared arity and
argument kinds, and nothing is bound that the header does not declare.  No compute call is made (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tnt_hip.h")


def declared():
    """name -> list of parameter type strings, parsed from the header's prototypes."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint32_t\s+(tnt_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        ps = [] if params in ("", "void") else [" ".join(p.split()) for p in params.split(",")]
        out[name] = [re.sub(r"\s*\w+$", "", p).strip() for p in ps]      # drop the parameter name
    return out


def kind(ctype):
    from masters_thesis_amd import _lib
    return {_lib.P: "ptr", _lib.I32: "int32_t", _lib.I64: "int64_t", _lib.U32: "uint32_t", _lib.U64: "uint64_t",
            _lib.F32: "float"}[ctype]


def test_header_is_plain_c_abi():
    src = open(HEADER).read()
    assert 'extern "C"' in src
    for bad in ("torch", "at::", "std::", "hipStream_t stream"):       # plain pointers and sizes; streams travel as void*
        assert bad not in re.sub(r"/\*.*?\*/", " ", src, flags=re.S), bad
    d = declared()
    assert len(d) >= 35
    for name, params in d.items():
        if params:
            assert params[-1] == "void*" or name in ("tnt_bn_nchunk", "tnt_lstm_seq_supported", "tnt_lstm_seq_bwd_work_floats", "tnt_gemm_fused_cfg", "tnt_embedding_bwd_parts", "tnt_attention_front_bwd_parts", "tnt_attention_metric_parts", "tnt_lc_seq_bwd_work_floats", "tnt_lc_seq_fwd_work_floats", "tnt_gemm3_work_floats", "tnt_gemm3_sync_words", "tnt_gemm3_plan", "tnt_gemm3_pair_supported"), (name, params[-1])    # trailing stream argument


def test_library_exports_and_binding_match_the_header():
    from masters_thesis_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the kernel library first: __graft_entry__.build()"
    lib = _lib.load()
    d = declared()
    assert set(_lib.SIGNATURES) == set(d), (sorted(set(d) - set(_lib.SIGNATURES)), sorted(set(_lib.SIGNATURES) - set(d)))
    raw = C.CDLL(_lib.LIB_PATH)
    for name, params in d.items():
        assert hasattr(raw, name), f"{name} is declared in the header but not exported"
        sig = _lib.SIGNATURES[name]
        assert len(sig) == len(params), (name, len(sig), len(params))
        for i, (ct, ps) in enumerate(zip(sig, params)):
            want = "ptr" if ps.endswith("*") else ps.replace("const ", "")
            assert kind(ct) == want, (name, i, ps, kind(ct))
    assert lib.tnt_version() >= 104


def test_product_has_no_cpu_fallback():
    """ops.backend() must refuse to run without the HIP extension / a GPU instead of degrading silently."""
    import torch
    import masters_thesis_amd.ops as ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    old = ops._backend
    ops._backend = None
    try:
        with pytest.raises(Exception):
            ops.backend()
    finally:
        ops._backend = old


def test_gemm3_plan_is_host_code_and_picks_a_one_round_grid():
    """tnt_gemm3_plan (the cost model of csrc/gemm3.hip) runs on the host: for a spread of shapes it returns a known tile,
    a split that leaves no workgroup without K stages, and never a split whose workgroups could not all be resident at
    once (they wait for each other inside the launch)."""
    import ctypes
    from masters_thesis_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    tiles = {1: (160, 128, 32), 2: (128, 160, 32), 3: (128, 128, 32), 4: (128, 80, 32), 5: (64, 128, 32), 6: (128, 64, 32), 7: (64, 64, 32),
             8: (256, 80, 32), 9: (64, 64, 64), 10: (64, 128, 64), 11: (128, 64, 64)}
    for M in (5, 64, 256, 512, 960, 1024, 4096):
        for N in (7, 256, 512, 2048, 5001):
            for K in (3, 64, 544, 960, 2048, 5001, 20000):
                for tA, tB in ((0, 0), (1, 0), (0, 1)):
                    for batch in (1, 2):
                        for allow in (0, 1):
                            t, s = ctypes.c_int32(0), ctypes.c_int32(0)
                            rc = lib.tnt_gemm3_plan(M, N, K, tA, tB, batch, allow, ctypes.byref(t), ctypes.byref(s))
                            assert rc == 0 and t.value in tiles and s.value >= 1, (M, N, K, tA, tB, batch, allow, rc)
                            bm, bn, bk = tiles[t.value]
                            if not allow:
                                assert s.value == 1
                            if s.value > 1:
                                units = -(-M // bm) * -(-N // bn) * s.value * batch
                                nst = -(-K // bk)
                                per = -(-nst // s.value)
                                assert units <= 256 and -(-nst // per) == s.value and per >= 4, (M, N, K, t.value, s.value)
                                assert lib.tnt_gemm3_work_floats(M, N, t.value, s.value, batch) == units * bm * bn
    t, s = ctypes.c_int32(0), ctypes.c_int32(0)
    assert lib.tnt_gemm3_plan(64, 64, 64, 1, 1, 1, 1, ctypes.byref(t), ctypes.byref(s)) != 0        # both transposed: refused
