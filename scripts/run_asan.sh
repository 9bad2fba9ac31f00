#!/bin/bash
# CPU test-suite against the AddressSanitizer + UndefinedBehaviorSanitizer build of the oracle (oracle/Makefile `asan`)
# and of the C++ host-layer test's CPU half.  GPU AddressSanitizer is not available on the pool: sanitizers run here only.
set -eu
cd "$(dirname "$0")/.."
make -C oracle -s asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
export SIPP_ORACLE_ASAN=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-8}
# python itself is not instrumented: preload the runtime so that the instrumented .so finds it
LD_PRELOAD="$ASAN_LIB" python -m pytest tests -q -m "not gpu" -x \
    --deselect tests/test_host_cpp.py --deselect tests/test_abi.py "$@"
# the C++ host layer (include/sipp_host.hpp): flat-proof parsing / record layouts under the sanitizers
make -C tests/host -s asan
env -u SIPP_ORACLE_ASAN python - <<'PY'
import numpy as np
from tests import _oracle
_oracle.stark_prove(0, np.load("tests/golden/sipp_n4_ios.npz")["g1"]).tofile("/tmp/sipp_asan_proof.bin")
PY
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 tests/host/test_sipp_circuit_asan layout /tmp/sipp_asan_proof.bin
# ... and from_flat on 20,000 damaged copies of that buffer: an Error or a round trip, never undefined behaviour
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 tests/host/test_sipp_circuit_asan fuzz /tmp/sipp_asan_proof.bin 20000
# the library's verifier (sipp_amd/csrc/verify.cpp, compiled into the test binary with the sanitizers) on 20,000 damaged proofs per kind
make -C tests/host -s verify_fuzz_asan
env -u SIPP_ORACLE_ASAN python - <<'PY'
import numpy as np
from tests import _oracle
g = np.load("tests/golden/sipp_n4_ios.npz")
_oracle.stark_prove(4, g["g1"]).tofile("/tmp/sipp_asan_g1h.bin")
_oracle.stark_prove(2, g["fq12"]).tofile("/tmp/sipp_asan_fq12.bin")
PY
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 tests/host/verify_fuzz_asan /tmp/sipp_asan_g1h.bin 20000
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 tests/host/verify_fuzz_asan /tmp/sipp_asan_fq12.bin 20000 5
# ... and the generic verifiers (opening proofs, outer proofs) on serialized cases: tests/test_product_verifier_generic.py writes and runs them
env -u SIPP_ORACLE_ASAN python -m pytest tests/test_product_verifier_generic.py -q -k sanitizers
