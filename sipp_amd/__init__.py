"""sipp_amd -- MI355X-native STARK sub-prover for the SIPP hot path (reference
src/verifier_circuit.rs:133-135).  The product is the C-ABI library libsipp_hip.so
(include/sipp_hip.h); this package is the thin Python harness tests and bench.py use
to drive it.  There is no CPU fallback: if the HIP library is missing or no GPU is
present the calls fail loudly."""
from ._lib import lib, Ctx, Instance, InstanceQueue, SippError, default_config, StarkConfig, FriParams, Challenger, PlonkParams, PlonkGate, PlonkCircuit, PlonkGenerator, PlonkSchedule, PlonkScheduleHost, CircuitData, io_shard, shard_ios, stark_verify  # noqa: F401
from . import proof_cost  # noqa: F401

__all__ = ["lib", "Ctx", "Instance", "InstanceQueue", "SippError", "default_config", "StarkConfig", "FriParams", "Challenger", "PlonkParams", "PlonkGate", "PlonkCircuit", "PlonkGenerator", "PlonkSchedule", "PlonkScheduleHost", "CircuitData", "proof_cost", "io_shard", "shard_ios", "stark_verify"]
