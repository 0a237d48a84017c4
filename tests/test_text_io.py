"""Row f1: the reference's text interchange format (SaveArrayToFile FSAC.cpp:492-505, LoadFloatArray
FSAC.cpp:454-490).  Host-side utilities of libfrequensee.so against the oracle's restatement and against
hand-derived strings of FString::SanitizeFloat."""
import ctypes as C

import numpy as np
import pytest


def sanitize(oracle_mod, v):
    buf = C.create_string_buffer(512)
    n = oracle_mod.load().fso_sanitize_float(float(v), buf, 512)
    return buf.value.decode() if n >= 0 else None


def test_sanitize_float_known_strings(oracle_mod):
    # "%f" then trailing zeros trimmed, at least one fractional digit (FString::SanitizeFloat)
    for v, s in ((0.0, "0.0"), (-0.0, "0.0"), (1.0, "1.0"), (0.5, "0.5"), (0.10622519254684448, "0.106225"),
                 (1e-7, "0.0"), (123456.789, "123456.789"), (-2.5, "-2.5"), (1e10, "10000000000.0"),
                 (0.1234565, "0.123457" if "%f" % 0.1234565 == "0.123457" else "0.123456")):
        assert sanitize(oracle_mod, v) == s


def test_save_load_round_trip_matches_oracle(pkg, oracle_mod, tmp_path):
    rng = np.random.default_rng(5)
    ir = np.concatenate([rng.normal(0, 0.2, 2000), [0.0, -0.0, 1.0, -1.0, 1e-9, 3.5e4, 7.25], np.zeros(100)]).astype(np.float32)
    a, b = tmp_path / "saved_ir.txt", tmp_path / "oracle_ir.txt"
    pkg._capi.save_array_to_file(ir, a)
    assert oracle_mod.load().fso_save_array_to_file(ir.ctypes.data, ir.shape[0], str(b).encode()) == 0
    assert a.read_bytes() == b.read_bytes()                      # byte-identical files
    txt = a.read_text()
    assert not txt.endswith("\n") and txt.count("\n") == ir.shape[0] - 1   # FString::Join: no trailing newline
    back = pkg._capi.load_float_array(a)
    ref = np.zeros(ir.shape[0] + 5, np.float32)
    n = oracle_mod.load().fso_load_float_array(str(a).encode(), ref.ctypes.data, ref.shape[0])
    assert n == ir.shape[0] == back.shape[0] and np.array_equal(back, ref[:n])
    assert np.abs(back - ir).max() <= 5.1e-7 * max(1.0, np.abs(ir).max())   # "%f": six decimals
    # LoadFloatArray: empty lines are culled, garbage parses as 0 (Atof), CRLF tolerated
    c = tmp_path / "in.txt"
    c.write_bytes(b"1.5\n\n-2\r\nabc\n  3e2\n\n")
    got = pkg._capi.load_float_array(c)
    assert list(got) == [1.5, -2.0, 0.0, 300.0]
    with pytest.raises(pkg.FrequenSeeError):
        pkg._capi.load_float_array(tmp_path / "missing.txt")
