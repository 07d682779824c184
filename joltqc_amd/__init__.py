"""MI355X-native J/K Fock-build and DFT grid backend behind the JoltQC ``apply(mf)`` interface."""
import os as _os

# ROCm maps HIP streams onto 4 hardware queues by default; the class kernels of one J/K build are independent and are
# spread over 8 streams (joltqc_amd/pyscf/jk.py N_STREAMS).  Must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
