"""CPU oracle for the tensorcircuit-ng statevector / expectation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``tensorcircuit-ng_amd/``) imports this directory.  The only legal users are
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the reported CPU number, never
as the thing shipped.

Contents
--------
``gates``     gate matrices restated from ``tensorcircuit/gates.py``
``tn``        tensor-network restatement of the reference algorithm
              (node graph -> single-gate merge -> greedy pairwise path ->
              ``np.tensordot`` chain -> final transpose), following
              ``tensorcircuit/cons.py`` and ``tensorcircuit/basecircuit.py``
``dense``     an independent gate-by-gate dense state-vector simulator
              (different algorithm, complex128) used to cross-check ``tn``
``workloads`` the HEA-A / HEA-B / TFIM workload builders of SURVEY.md section 8

Parity pin: the reference itself cannot be imported in the build container
(``opt_einsum`` / ``tensornetwork`` / ``graphviz`` are absent), so
the oracle is pinned against the known-answer constants of the reference's own
tests (``tests/test_oracle_kat.py`` cites each ``file:line``) and against the
independent dense simulator.  At BASELINE.json's sizes (n >= 24) parity with the
third-party arithmetic (``tensornetwork``/``opt_einsum``) is unpinned by any
reference test; see DESIGN.md.
"""
