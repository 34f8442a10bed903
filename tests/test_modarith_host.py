"""Host compile (-DFHE_EMU) of the device arithmetic header: the 8-term column sums with a single 64-bit Barrett reduction
(sum8, csrc/modarith.h) against 128-bit integers for every admissible modulus size (tests/sum8_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sum8_against_128_bit_arithmetic(tmp_path):
    exe = str(tmp_path / "sum8_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-DFHE_EMU", f"-I{ROOT}/tests/emu", f"-I{ROOT}/openfhe-development_amd/csrc",
                           os.path.join(ROOT, "tests", "sum8_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sum8_check OK" in out.stdout, out.stdout + out.stderr
