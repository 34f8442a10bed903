"""The gfx950 butterfly / reduction code of ntt_bfly_pinned.h is generated; the generator simulates every block it emits
against the arithmetic it replaces (python integers) before writing.  This test re-runs generator + simulation and
checks that the committed header is its output."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_header_is_current_and_simulates():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ntt_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_generated_row8_header_is_current_and_simulates():
    """ntt_bfly8_pinned.h (the 8-residues-per-lane row pass): same generator discipline, plus the lazy-inverse plan tables the lane
    emulator's C++ butterflies follow"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ntt8_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
