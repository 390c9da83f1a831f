"""The host logic of the range decoder (data_loader/gz_shard.py, gz.py) without a GPU: range cuts, the record-boundary rule, maps applied in
rank order, CRC combination, the verdict every rank computes from the gathered facts."""
import zlib

import numpy as np

from ribodetector_amd import gz
from ribodetector_amd.data_loader import gz_shard as gs


def test_range_bounds_are_section_multiples_and_cover_the_file():
    for size in (0, 1, 16384, 16385, 10 ** 6 + 3, (1 << 30) + 12345):
        for w in (1, 2, 3, 8):
            b = gs.range_bounds(size, w)
            assert b[0] == 0 and b[-1] == size and len(b) == w + 1 and all(x <= y for x, y in zip(b, b[1:]))
            assert all(x % 16384 == 0 for x in b[:-1])


def test_record_cut_rule():
    f = gs.find_record_cut
    assert f(b"@r1\nAC\n+\nII\n", None, False) == 0 and f(b"@r1\nAC\n+\nII\n", 10, False) == 0
    assert f(b"GT\n+\nII\n@r2\nAC\n+\nII\n", ord("A"), False) == 8
    assert f(b"@II\n@r2\nACGT\n+\n@III\n", 10, False) == 4                 # a quality line that starts with '@' is not a header
    assert f(b"II\n@r2\nAC\n", ord("I"), False) is None                     # not enough text to see the '+' line
    assert f(b"no newline at all", ord("x"), False) is None
    assert f(b"ACGT\n>r2 x\nAC\n", ord("A"), True) == 5 and f(b">r\nAC\n", None, True) == 0 and f(b"ACGT\nAC\n", 10, True) is None


def test_maps_compose_to_the_window_zlib_would_have():
    """three 'ranges' of a text, each described by its map (bytes it produced, markers where it copied from the window in front of it):
    applying the maps in order gives the last 32 KiB of the text"""
    rng = np.random.default_rng(3)
    text = rng.integers(32, 127, 100000, dtype=np.uint8)
    cuts = [0, 20000, 70000, 100000]
    window, valid = np.zeros(32768, dtype=np.uint8), 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        n = b - a
        m = np.empty(32768, dtype=np.uint16)
        if n >= 32768:
            m[:] = text[b - 32768:b]
        else:
            m[:32768 - n] = 0x8000 | np.arange(n, 32768, dtype=np.uint16)      # shifted entries of the window in front
            m[32768 - n:] = text[a:b]
        window = gz.apply_map(m, window)
        valid = min(32768, valid + n)
        assert np.array_equal(window[32768 - valid:], text[b - valid:b])


def test_crc32_combine_is_zlibs():
    rng = np.random.default_rng(5)
    for la, lb in ((0, 0), (1, 0), (0, 5), (1000, 1), (77777, 12345), (3, 1 << 20)):
        a, b = rng.integers(0, 256, la, dtype=np.uint8).tobytes(), rng.integers(0, 256, lb, dtype=np.uint8).tobytes()
        assert gz.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)


def _meta(first, nxt, ended=False, status=0, segs=1, n=100000, fresh=False):
    return {"status": status, "why": "decode error" if status else None, "first_abs": first, "next_abs": nxt, "ended": ended, "fresh_after": fresh,
            "n_text": n, "map": np.arange(32768, dtype=np.uint16) | 0x8000, "segs": [{"n_text": n // segs, "final": False, "trailer": None}] * segs}


def test_verdict_is_a_function_of_the_gathered_facts():
    ok = [[_meta(80, 5000)], [_meta(5000, 9000)], [_meta(9000, None, ended=True)]]
    assert gs._verdict_x1(ok, 3, [1 << 20]) is None
    bad = [[_meta(80, 5000)], [_meta(5008, 9000)], [_meta(9000, None, ended=True)]]
    assert "does not start where" in gs._verdict_x1(bad, 3, [1 << 20])
    assert gs._verdict_x1([[_meta(80, 5000)], [_meta(None, 9000)], [_meta(9000, None, ended=True)]], 3, [1 << 20]) is not None
    assert "no block start" in gs._verdict_x1([[_meta(None, 5000)], [_meta(5000, None, ended=True)]], 2, [1 << 20])
    assert "ended before" in gs._verdict_x1([[_meta(80, 5000)], [_meta(5000, 9000)], [_meta(9000, 12000)]], 3, [1 << 20])
    assert "decode error" in gs._verdict_x1([[_meta(80, 5000)], [_meta(5000, 9000, status=1)], [_meta(9000, None, ended=True)]], 3, [1 << 20])
    # the window in front of a rank: a rank whose share holds a member's end hands on only what it decoded behind it
    w, v = gs.start_window(ok, 0, 2)
    assert v == 32768
    two = [[_meta(80, 5000, n=1000)], [_meta(5000, 9000, segs=2, n=600)], [_meta(9000, None, ended=True)]]
    assert gs.start_window(two, 0, 1)[1] == 1000 and gs.start_window(two, 0, 2)[1] == 300
    fresh = [[_meta(80, 5000, fresh=True)], [_meta(5000, None, ended=True)]]
    assert gs.start_window(fresh, 0, 1)[1] == 0
