"""The batched HIP propagator that stands where the reference calls the Basilisk C++ engine."""
from .config import default_config  # noqa: F401
from .propagator import BatchedPropagator, pack_ic  # noqa: F401
