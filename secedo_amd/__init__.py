"""secedo_amd -- MI355X-native similarity-matrix path of SECEDO (computeSimilarityMatrix).

Only what the path needs: the HIP kernels + C-ABI (csrc/, include/secedo_simmat.h) and the
host-side mirror of the reference interface (similarity_matrix.py).
"""
from .pileup import FlatPileup, PosData, flatten  # noqa: F401
from .similarity_matrix import (InvalidNormalization, NORMALIZATIONS, SecedoError,  # noqa: F401
                                SimilarityMatrixPlan, compute_similarity_matrix, get_devices, llr, llr_closed_form,
                                set_devices, to_enum)
from .filter import Filter, NO_POS, filter_resident  # noqa: F401,E402
from .pileup_reader import get_grouping, read_pileup  # noqa: F401,E402
from .spectral import laplacian, smallest_eigenpairs  # noqa: F401,E402
from .em import expectation_maximization  # noqa: F401,E402
