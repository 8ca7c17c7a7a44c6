"""MI355X-native ReSTIR DI hot path. Host-side Python mirror of the C-ABI."""
from . import types, scenes  # noqa: F401
