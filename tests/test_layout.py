"""Structure rules: the product never touches the oracle; required files exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_does_not_import_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|/root/reference", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "videogpa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f"{f} references the oracle / reference tree"
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read()


def test_required_layout():
    for p in ("include/videogpa_hip.h", "oracle/__init__.py", "tests/golden/make_golden.py", "bench.py", "__graft_entry__.py",
              "DESIGN.md", "INTEGRATION.md", "profiles"):
        assert os.path.exists(os.path.join(ROOT, p)), p


def test_generated_attention_loops_are_in_sync_with_their_generator():
    """csrc/w1_*_loop.inc / *_clobbers.inc are the checked-in output of tools/gen_w1_asm.py (default knobs)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("W1_ABLATE", "W1_KNOBS")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_w1_asm.py"), "--check"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
