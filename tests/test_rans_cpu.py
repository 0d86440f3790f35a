"""Integer path on the CPU: CDF quantiser and rANS coder. Three independent implementations — pure Python oracle
(oracle/rans_py.py), plain-C oracle (oracle/rans_oracle.c) and the product's C++ coder (clc_amd/csrc/rans_host.cpp,
through the C ABI) — must agree bit for bit with each other and with the known-answer vectors in tests/golden/."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tables():
    from oracle import leaves

    gc = leaves.GaussianConditional(None)
    gc.update_scale_table(leaves.get_scale_table())
    return (np.ascontiguousarray(gc.quantized_cdf.numpy().astype(np.int32)), gc.cdf_length.numpy().astype(np.int32),
            gc.offset.numpy().astype(np.int32), gc)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLD, "rans_kat.json")) as f:
        return json.load(f)


def test_gaussian_tables_known_answers(tables, kat):
    cdf, ln, off, gc = tables
    assert list(cdf.shape) == kat["gaussian_cdf_shape"] == [64, 3133]
    assert hashlib.sha256(cdf.tobytes()).hexdigest() == kat["gaussian_cdf_sha256"]
    assert cdf[0, :6].tolist() == kat["cdf_row0"] == [0, 1, 65534, 65535, 65536, 0]  # SURVEY.md A.5 self-check
    assert [int(ln[0]), int(ln[-1])] == kat["cdf_len_first_last"] == [5, 3133]
    assert [int(off[0]), int(off[-1])] == kat["offset_first_last"] == [-1, -1565]


def test_product_tables_equal_oracle(tables):
    from clc_amd.entropy_models import GaussianConditional
    from clc_amd.models.clc import get_scale_table

    cdf, ln, off, _ = tables
    gc = GaussianConditional(None)
    gc.update_scale_table(get_scale_table())
    c2, l2, o2 = gc.host_tables()
    assert np.array_equal(c2, cdf) and np.array_equal(l2, ln) and np.array_equal(o2, off)


def test_pmf_to_quantized_cdf_three_ways(kat):
    from clc_amd.entropy_models import pmf_to_quantized_cdf
    from oracle import rans_c, rans_py

    for case in kat["pmf_cases"]:
        p = np.asarray(case["pmf"], dtype=np.float32)
        assert rans_py.pmf_to_quantized_cdf(p.tolist()) == case["cdf"]
        assert rans_c.pmf_to_quantized_cdf(p) == case["cdf"]
        assert pmf_to_quantized_cdf(p).tolist() == case["cdf"]
        assert case["cdf"][0] == 0 and case["cdf"][-1] == 65536 and all(b > a for a, b in zip(case["cdf"], case["cdf"][1:]))
    rng = np.random.default_rng(5)
    for n in (2, 5, 33, 500, 3131):
        p = rng.random(n).astype(np.float32) ** 6
        p /= p.sum()
        assert rans_c.pmf_to_quantized_cdf(p) == rans_py.pmf_to_quantized_cdf(p.tolist()) == pmf_to_quantized_cdf(p).tolist()


def test_pmf_errors():
    from clc_amd import lib
    from clc_amd.entropy_models import pmf_to_quantized_cdf

    with pytest.raises(lib.ClcError):
        pmf_to_quantized_cdf(np.array([0.5, -0.1, 0.6], dtype=np.float32))
    with pytest.raises(lib.ClcError):
        pmf_to_quantized_cdf(np.array([0.0, 0.0], dtype=np.float32))
    with pytest.raises(lib.ClcError):
        pmf_to_quantized_cdf(np.array([0.5, np.nan], dtype=np.float32))


def test_known_answer_streams(tables, kat):
    from clc_amd import ans
    from oracle import rans_c, rans_py

    cdf, ln, off, _ = tables
    for case in kat["cases"]:
        sym, idx, want = case["symbols"], case["indexes"], bytes.fromhex(case["stream_hex"])
        assert rans_py.RansEncoder().encode_with_indexes(sym, idx, cdf.tolist(), ln.tolist(), off.tolist()) == want, case["name"]
        assert rans_c.encode(sym, idx, cdf, ln, off) == want, case["name"]
        assert ans.encode(sym, idx, cdf, ln, off) == want, case["name"]
        assert ans.decode(want, idx, cdf, ln, off).tolist() == sym, case["name"]
        out, words = rans_c.decode(want, idx, cdf, ln, off)
        assert out.tolist() == sym and words * 4 == len(want), case["name"]
    assert bytes.fromhex(kat["cases"][0]["stream_hex"]) == bytes.fromhex("0000008000000000")  # empty stream = state L only


def test_random_roundtrip_and_cross_decode(tables):
    from clc_amd import ans
    from oracle import rans_c, rans_py

    cdf, ln, off, gc = tables
    rng = np.random.default_rng(42)
    for n in (1, 7, 1000, 81920):
        idx = rng.integers(0, 64, n).astype(np.int32)
        sym = np.round(rng.normal(0, gc.scale_table.numpy()[idx])).astype(np.int32)
        esc = rng.random(n) < 0.01
        sym[esc] = rng.integers(-(2 ** 24), 2 ** 24, int(esc.sum())).astype(np.int32)
        s_prod = ans.encode(sym, idx, cdf, ln, off)
        assert s_prod == rans_c.encode(sym, idx, cdf, ln, off)
        if n <= 1000:
            assert s_prod == rans_py.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist())
            assert rans_py.RansDecoder().decode_with_indexes(s_prod, idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist()) == sym.tolist()
        assert np.array_equal(ans.decode(s_prod, idx, cdf, ln, off), sym)
        assert len(s_prod) % 4 == 0 and len(s_prod) >= 8


def test_reference_api_surface(tables):
    """BufferedRansEncoder / RansDecoder used the way /root/reference/models/CLC_run.py:658,712-713,762-763,793 does (python lists, slice by slice)."""
    from clc_amd import ans

    cdf, ln, off, gc = tables
    rng = np.random.default_rng(3)
    chunks = []
    for _ in range(5):
        idx = rng.integers(0, 64, 300).astype(np.int32)
        chunks.append((np.round(rng.normal(0, gc.scale_table.numpy()[idx])).astype(np.int32), idx))
    enc = ans.BufferedRansEncoder()
    sym_all, idx_all = [], []
    for s, i in chunks:
        sym_all.extend(s.tolist())
        idx_all.extend(i.tolist())
    enc.encode_with_indexes(sym_all, idx_all, cdf.tolist(), ln.tolist(), off.tolist())
    stream = enc.flush()
    dec = ans.RansDecoder()
    dec.set_stream(stream)
    for s, i in chunks:  # decoded incrementally, one slice at a time
        assert dec.decode_stream(i.tolist(), cdf.tolist(), ln.tolist(), off.tolist()) == s.tolist()
    assert ans.BufferedRansEncoder().flush() == bytes.fromhex("0000008000000000")
    with pytest.raises(ValueError):
        ans.encode([1, 2], [0], cdf, ln, off)
    with pytest.raises(ValueError):
        ans.encode([1], [64], cdf, ln, off)
    with pytest.raises(Exception):
        ans.RansDecoder().set_stream(b"\x00\x01\x02")
