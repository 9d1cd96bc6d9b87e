"""CPU oracle for the Dropout-Decoding hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 + numpy, our own words) of the
reference algorithm in kigb/DropoutDecoding `models/{llava,llavanext,instructblip}.py`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(`dropoutdecoding_amd/`) never imports it and never falls back to it.

Parity status: PINNED.  Every function here is checked against golden vectors that
were produced by importing and running the reference itself in the build container
(`oracle/gen_golden.py`, fixtures under `tests/golden/`; transformers 5.15.0,
torch 2.10.0 CPU fp32 — the reference pins transformers 4.44.0 / torch 2.4.0, whose
sources are not available here, see DESIGN.md "Oracle").  The reference ships no
tests or golden vectors of its own.
"""
