"""remo3d_amd: ReMo3D's per-measurement-point FEM path on MI355X."""
import os as _os

# Hardware queues of the HIP runtime in this process (its default: 4; read once, at the first HIP call - so it is set HERE, when
# the package is first imported, and only if the caller has not chosen a value).  Every context (model.Model opens
# DEFAULT_CONTEXTS per GPU) has its own HIP stream; streams beyond the runtime's queue count share a queue, i.e. their kernels
# wait for each other.  bench.py at size L, one box, alternating runs (profiles/r04_bl_hw_queues_and_contexts.json):
# 4 queues / 3 contexts 159.9-160.0 points/s, 4 / 5: 158.7-159.0, 8 / 5: 168.1-168.3, 8 / 6: 165.7-166.5, 2 / 3: 142.1.
# A process that initialises HIP before importing this package (or drives the C ABI without it) exports the variable itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
