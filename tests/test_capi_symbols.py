"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/atsc_hip.h
declares, the host-side helpers agree with the reference's known answers, and the compressor
refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest

from tests.golden import kat as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def A():
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    return atsc_amd


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "atsc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(atsc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(A):
    names = _declared_functions()
    assert len(names) >= 20
    lib = A.capi.lib()
    for n in names:
        assert hasattr(lib, n), "libatsc_hip.so does not export %s" % n
        assert n in A.capi.SIGNATURES, "capi.py does not bind %s" % n
    assert set(A.capi.SIGNATURES) <= set(names)


def test_product_does_not_touch_the_oracle():
    # the oracle is test infrastructure: nothing under atsc_amd/ may import, link or call it
    for dp, _, files in os.walk(os.path.join(ROOT, "atsc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "libatsc_oracle" not in txt and "orc_" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_chunk_sizes_and_next_size(A):
    for n, want in K.CHUNK_SIZES:
        assert A.chunk_sizes(n) == want
    for n, want in K.NEXT_SIZE:
        assert A.capi.lib().atsc_next_size(n) == want


def test_clean_data(A):
    x = np.array([1.0, np.nan, 2.0, np.inf, -np.inf, 3.0])
    assert list(A.clean_data(x)) == [1.0, 2.0, 3.0]


def test_bro_prefix_and_open(A):
    assert A.bro_prefix(1) == bytes(K.STREAM_CONSTANT_1024[:10])
    p = A.bro_prefix(300)  # u8 frame counter wraps (header.rs:52-54), varint count does not
    assert p[8] == 300 % 256 and p[9:] == bytes([251, 44, 1])
    off, nf = A.bro_open(bytes(K.STREAM_CONSTANT_1024))
    assert (off, nf) == (10, 1)
    bad = bytearray(K.STREAM_CONSTANT_1024)
    bad[4] = 9
    with pytest.raises(A.AtscError) as ei:
        A.bro_open(bytes(bad))
    assert ei.value.rc == A.capi.E_VERSION
    bad = bytearray(K.STREAM_CONSTANT_1024)
    bad[1] = 0
    with pytest.raises(A.AtscError) as ei:
        A.bro_open(bytes(bad))
    assert ei.value.rc == A.capi.E_FORMAT


def test_no_gpu_means_loud_failure(A):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(A.AtscError) as ei:
        A.Context(0)
    assert ei.value.rc == A.capi.E_NO_DEVICE


def test_payload_bound(A):
    # RLE with all-distinct F64 values is the largest payload: 3 + n*(8+1+varint(idx))
    for n in (1, 256, 4096):
        worst = 3 + 2 + n * (8 + 1 + 3)
        assert A.capi.lib().atsc_payload_bound_bytes(n) >= worst
