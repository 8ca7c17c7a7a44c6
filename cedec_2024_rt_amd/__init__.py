"""MI355X-native ReSTIR DI hot path. Host-side Python mirror of the C-ABI."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the frame wants five streams side by side (main,
# pipelined stage 0, tail, second lane, RCCL). Read by the runtime when it initialises: set before anything touches the GPU
# (a value the caller has set is respected). DESIGN.md section 7, profiles/r03_hw_queue_mapping.txt.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import types, scenes  # noqa: E402,F401
