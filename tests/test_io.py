"""Trajectory file format (reference src/io.cpp:126-218): Python writer/reader (mav_trajectory_generation_amd/io.py),
C++ veneer writer/reader (include/compat/mav_trajectory_generation/io.h) and PyYAML as the independent judge of what
is valid YAML.  Host-only; also runs the C++ selftest of the veneer's Trajectory analysis helpers."""
import os
import subprocess

import numpy as np
import pytest
import yaml

from mav_trajectory_generation_amd import io as mio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_io")


@pytest.fixture(scope="module")
def exe():
    src = os.path.join(ROOT, "tests", "cpp", "test_io.cpp")
    hdr_dir = os.path.join(ROOT, "include", "compat", "mav_trajectory_generation")
    deps = [src] + [os.path.join(hdr_dir, f) for f in os.listdir(hdr_dir)]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include", "compat"),
                               "-I" + os.path.join(ROOT, "include"), "-o", EXE, src])
    return EXE


def parse_dump(text):
    lines = text.strip().splitlines()
    k, n, d = (int(x) for x in lines[0].split())
    times_ns, coeffs, i = [], np.zeros((k, d, n)), 1
    for s in range(k):
        times_ns.append(int(lines[i])); i += 1
        for dd in range(d):
            coeffs[s, dd] = [float(x) for x in lines[i].split()]; i += 1
    return times_ns, coeffs


def test_cpp_selftest(exe, tmp_path):
    r = subprocess.run([exe, "selftest", str(tmp_path / "t.yaml")], capture_output=True, text=True)
    assert r.returncode == 0 and "IO TESTS PASSED" in r.stdout, r.stdout + r.stderr


def test_python_round_trip_and_time_truncation(tmp_path):
    rng = np.random.default_rng(3)
    coeffs = rng.normal(size=(5, 3, 10)) * 10.0 ** rng.integers(-8, 8, size=(5, 3, 10))
    times = np.array([3.970847833173347, 0.1, 2.5e-10, 17.0, 1.0 / 3.0])
    f = str(tmp_path / "traj.yaml")
    assert mio.segments_to_file(f, coeffs, times)
    c2, t2 = mio.segments_from_file(f)
    assert np.array_equal(c2, coeffs)                         # 17 significant digits: exact
    assert [mio.time_to_nsec(t) for t in times] == [3970847833, 100000000, 0, 17000000000, 333333333]
    assert np.allclose(t2, np.floor(times * 1e9) * 1e-9, rtol=0, atol=1e-18)      # truncated nanoseconds
    doc = yaml.safe_load(open(f).read())                      # the reference's schema, key for key (io.cpp:27-31)
    assert list(doc.keys()) == ["segments"] and set(doc["segments"][0].keys()) == {"N", "D", "time", "coefficients"}
    assert doc["segments"][0]["N"] == 10 and doc["segments"][0]["D"] == 3 and len(doc["segments"][0]["coefficients"]) == 3
    assert not mio.segments_to_file(str(tmp_path / "no_such_dir" / "x.yaml"), coeffs, times)


def test_python_reader_rejects_what_the_reference_rejects():
    with pytest.raises(ValueError):
        mio.segments_from_yaml("foo: 1\n")
    with pytest.raises(ValueError):
        mio.segments_from_yaml("segments:\n  - N: 3\n    D: 1\n    coefficients:\n      - [1, 2, 3]\n")
    with pytest.raises(ValueError):
        mio.segments_from_yaml("segments:\n  - N: 3\n    D: 2\n    time: 5\n    coefficients:\n      - [1, 2, 3]\n")
    with pytest.raises(ValueError):
        mio.segments_from_yaml("segments:\n  - N: 4\n    D: 1\n    time: 5\n    coefficients:\n      - [1, 2, 3]\n")


def test_cpp_and_python_interoperate(exe, tmp_path):
    # C++ writes -> PyYAML + Python reader
    f = str(tmp_path / "cpp.yaml")
    assert subprocess.run([exe, "write", f]).returncode == 0
    c, t = mio.segments_from_file(f)
    times_ns, want = parse_dump(subprocess.run([exe, "read", f], capture_output=True, text=True).stdout)
    assert c.shape == (2, 2, 6) and np.array_equal(c, want)
    assert [int(round(x * 1e9)) for x in t] == times_ns
    # Python writes -> C++ reader (incl. negative, tiny, huge values)
    rng = np.random.default_rng(9)
    coeffs = rng.normal(size=(3, 4, 12)) * 10.0 ** rng.integers(-200, 200, size=(3, 4, 12))
    times = np.array([0.5, 1.25, 123.456789012])
    g = str(tmp_path / "py.yaml")
    assert mio.segments_to_file(g, coeffs, times)
    r = subprocess.run([exe, "read", g], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    times_ns, got = parse_dump(r.stdout)
    assert np.array_equal(got, coeffs) and times_ns == [mio.time_to_nsec(x) for x in times]
