"""eigenexa_amd -- MI355X-native implementation of EigenExa's eigen_sx / eigen_s hot path.

Host-side mirror of the reference's Fortran module ``eigen_libs_mod`` (src/eigen_libs.F:14-218): same
entry-point names, argument meaning and defaults, over the C-ABI of ``libeigenexa_amd.so``
(include/eigenexa_amd.h).  All numerical work happens in hand-written HIP kernels; there is no CPU path.
"""
from .api import (  # noqa: F401
    eigen_init,
    eigen_free,
    eigen_comm_info,
    eigen_get_matdims,
    eigen_get_procs,
    eigen_get_id,
    eigen_get_version,
    eigen_get_errinfo,
    eigen_memory_internal,
    eigen_loop_start,
    eigen_loop_end,
    eigen_translate_l2g,
    eigen_translate_g2l,
    eigen_owner_node,
    eigen_owner_index,
    eigen_sx,
    eigen_s,
    eigen_h,
    eigen_sx_bc,
    eigen_s_bc,
    numroc,
    KMATH_EIGEN_GEV,
    eigen_NB_f,
    eigen_NB_b,
)
from . import layout  # noqa: F401
