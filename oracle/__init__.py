"""CPU oracle for the quasimap path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package. The product package ``gramtools_amd`` never does.

Parity status: pinned against the reference's own known-answer tests
(``tests/golden/*.json``; see ``tests/test_oracle_golden.py``).
"""
from .oracle import Oracle, build_oracle, RNG_LEMIRE, RNG_DIVISION  # noqa: F401
from .prg_text import (  # noqa: F401
    encode_prg,
    prg_string_to_ints,
    encode_dna_bases,
    ints_to_prg_bytes,
)
