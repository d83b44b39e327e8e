"""CPU test of the Eigen stand-in that oracle/_ref (the compiled reference) is built against: its inverse() and SparseQR follow
Eigen's algorithm classes; here: they are correct, the fill-reducing column order on the
block-tridiagonal R_PP is the natural order (so natural-order results stand), and a rank-deficient system gets a basic solution
the way Eigen's rank-revealing SparseQR (LIN:365-378) returns one."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stand_in_algorithms(tmp_path):
    exe = str(tmp_path / "mini_eigen_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "oracle", "ref_shim"),
                           os.path.join(ROOT, "tests", "cpp", "mini_eigen_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    inv = re.findall(r"inverse n=(\d) residual=(\S+)", out)
    assert [int(n) for n, _ in inv] == [1, 2, 3, 4, 5, 6] and all(float(r) < 1e-13 for _, r in inv)
    assert float(re.search(r"inverse zero-pivot residual=(\S+)", out).group(1)) < 1e-12
    for f, rank, order, res in re.findall(r"qr f=(\d) rank=(\d+) order=(\S+) residual=(\S+)", out):
        f, rank = int(f), int(rank)
        order = [int(x) for x in order.strip(",").split(",")]
        assert rank == 7 * f and sorted(order) == list(range(7 * f)) and float(res) < 1e-12
        # minimum degree with ties to the lowest index walks the chain from its first block; only inside the last two
        # blocks (one clique by then) may the order differ from the natural one -- and it does not
        assert order == list(range(7 * f))
    rank, zeros, res = re.search(r"rankdef rank=(\d) zeros=(\d) residual=(\S+)", out).groups()
    assert int(rank) == 3 and int(zeros) == 1 and float(res) < 1e-12
