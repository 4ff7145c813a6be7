"""CPU oracle for the FDM diffusion-sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 CPU restatement (torch CPU ops / numpy / a few lines of C)
of the reference algorithm, function by function, each citing the reference file:line
it follows.  It exists to *check* the HIP path and to provide the `cpu_baseline` leg of
bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  The product package (face-diffusion-model_amd/fdm_amd) never imports
anything from here and fails loudly if its HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported in the build
container through tests/golden/refshim.py; the committed fixtures live in
tests/golden/*.npz and were produced by tests/golden/make_golden.py.
"""
