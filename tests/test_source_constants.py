"""Constants that a GPU test restates in Python must be the kernels' own (CPU: reads the sources, no GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _src(name):
    with open(os.path.join(ROOT, "flock_amd", "csrc", name)) as f:
        return f.read()


def test_q8_bucket_sizing_in_the_parity_test_is_the_kernels():
    """tests/test_gpu_parity.py::_q8_part_log2 / _q8_part_bucket build inputs that land in ONE hash bucket of q8's grouped hash path:
    they must size and hash exactly as q8.hip does."""
    q8, tab = _src("q8.hip"), _src("hashtab.hpp")
    m = re.search(r"kPartPersonsPerBucket = (\d+), kPartAuctionsPerBucket = (\d+);", q8)
    assert m, "q8.hip no longer states the bucket sizing in the form the test reads"
    persons, auctions = int(m.group(1)), int(m.group(2))
    max_log2 = int(re.search(r"constexpr int kPartMaxLog2 = (\d+);", q8).group(1))
    fib = int(re.search(r"constexpr uint32_t kFibHash = (0x[0-9A-Fa-f]+)u;", tab).group(1), 16)
    with open(os.path.join(ROOT, "tests", "test_gpu_parity.py")) as f:
        test = f.read()
    assert f"while l < {max_log2} and ((max_p >> l) > {persons} or (max_a >> l) > {auctions}):" in test
    assert f"np.uint32(0x{fib:08X})" in test
    assert "return log2nb ? (k * kFibHash) >> (32 - log2nb) : 0u;" in q8          # the bucket = the hash's top bits
