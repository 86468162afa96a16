#!/usr/bin/env python3
"""Run bench.py with module attributes of heal_swin_amd.ops overridden (policy thresholds, feature flags) -- for same-box A/B runs
of settings that have no environment switch.
usage: python tools/policy_ab.py OWN_BIAS_MAX_K=768 MLP_KEEP_ACT=False [@hs_gemm_nt_set_tile=2] -- --workload T256 --steps 8 --no-companions ..."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

argv = sys.argv[1:]
split = argv.index("--") if "--" in argv else len(argv)
sets, rest = argv[:split], argv[split + 1:]

import bench  # noqa: E402

_build = bench.build_model


def build_model(*a, **kw):  # (the overrides are applied where bench.py itself first imports the package: same initialisation order)
    from heal_swin_amd import ops

    for s in sets:
        name, val = s.split("=", 1)
        if name.startswith("@"):  # @hs_function=int: call a C entry point of the library with one integer argument
            from heal_swin_amd import _lib
            rc = getattr(_lib.lib, name[1:])(int(val))
            print(f"[policy_ab] {name[1:]}({val}) -> {rc}", file=sys.stderr)
            continue
        if not hasattr(ops, name):
            raise SystemExit(f"heal_swin_amd.ops has no attribute {name}")
        try:
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        if getattr(ops, name) != val:
            setattr(ops, name, val)
            print(f"[policy_ab] ops.{name} = {val!r}", file=sys.stderr)
    return _build(*a, **kw)


bench.build_model = build_model
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
bench.main()
