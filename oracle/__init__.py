"""CPU oracle for the HEAL-SWIN hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the integer/index work, plain
torch-CPU fp32 for the floating-point work) of the algorithms on the reference's
hot path (`/root/reference/heal_swin/models_torch/{hp_windowing,hp_shifting,
swin_hp_transformer}.py`).  Every function cites the reference file:line it follows.

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from here -- as the checker, never as the thing shipped;
  * nothing under `heal_swin_amd/` imports `oracle`.

Pinning status ("Oracle"):
  * index tables, masks, module/whole-model forward + gradients, losses: PINNED against
    golden vectors produced by importing the reference itself in the build container
    (`tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`);
  * `healpix.nest2ring/ring2nest`: the reference delegates these to healpy==1.15.2
    (`hp_shifting.py:329,333`), which is NOT in `/root/reference` and not installed:
    PARITY UNPINNED beyond healpy's published docstring known-answers and the
    bijection/inverse properties (tests/test_oracle_tables.py).
"""
