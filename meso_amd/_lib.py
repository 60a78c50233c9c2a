"""ctypes loader for libmeso_hip.so (the C ABI of include/meso_hip.h).

There is no CPU fallback: if the shared library is missing the import of any compute entry
point fails loudly, and ``meso_init`` fails when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MESO_LIB: another build of the same library inside the tree (A/B timing of kernel variants in one GPU session)
LIB_PATH = os.environ.get("MESO_LIB") or os.path.join(_HERE, "libmeso_hip.so")

_vp, _i, _d, _f = C.c_void_p, C.c_int, C.c_double, C.c_float
_i64, _u32, _sz, _cp = C.c_int64, C.c_uint32, C.c_size_t, C.c_char_p

HOST_EXCHANGE_FN = C.CFUNCTYPE(_i, _vp, _i, C.POINTER(_i), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp),
                               C.POINTER(_sz))

# name -> (restype, argtypes); mirrors include/meso_hip.h one to one
SIGNATURES = {
    "meso_last_error": (_cp, []),
    "meso_version": (_i, []),
    "meso_init": (_i, [_i, C.POINTER(_vp)]),
    "meso_finalize": (_i, [_vp]),
    "meso_device_sync": (_i, [_vp]),
    "meso_set_option": (_i, [_vp, _cp, _d]),
    "meso_set_box": (_i, [_vp, _vp, _vp, _vp]),
    "meso_comm_init": (_i, [_vp, _i, _i, _vp, _i, _vp, _sz]),
    "meso_comm_get_unique_id": (_i, [_vp, _sz]),
    "meso_decomp_procgrid": (_i, [_i, _vp, _vp]),
    "meso_decomp_plan": (_i, [_vp, _vp, _vp, _vp, _i, _d] + [_vp] * 8),
    "meso_comm_count": (_i, [_vp, C.POINTER(_i)]),
    "meso_membw_probe": (_i, [_vp, _sz, _i, C.POINTER(_d)]),
    "meso_pair_floor": (_i, [_vp, _i, _i, C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "meso_pair_kernel_name": (_i, [_vp, C.c_char_p, _i]),
    "meso_tally_ev": (_i, [_vp]),
    "meso_xchg_stats": (_i, [_vp, C.c_char_p, _i]),
    "meso_comm_set_host_exchange": (_i, [_vp, HOST_EXCHANGE_FN, _vp]),
    "meso_set_mass": (_i, [_vp, _i, _vp]),
    "meso_atoms_upload": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "meso_atoms_count": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "meso_atoms_download": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "meso_neighbor": (_i, [_vp, _d, _i, _i, _i]),
    "meso_pair_dpd_settings": (_i, [_vp, _i, _d, _i]),
    "meso_pair_dpd_coeff": (_i, [_vp, _i, _i, _d, _d, _d, _d, _d]),
    "meso_pair_dpd_polyforce_coeff": (_i, [_vp, _i, _i, _d, _d, _i, _vp]),
    "meso_pair_dpd_tableforce_coeff": (_i, [_vp, _i, _i, _d, _d, _i, _vp]),
    "meso_special_bonds": (_i, [_vp, _d, _d, _d]),
    "meso_bonds_upload": (_i, [_vp, _i, _vp, _vp, _vp]),
    "meso_bond_style_harmonic": (_i, [_vp, _i]),
    "meso_bond_coeff": (_i, [_vp, _i, _d, _d]),
    "meso_bond_style_fene": (_i, [_vp, _i]),
    "meso_bond_coeff_fene": (_i, [_vp, _i, _d, _d, _d, _d]),
    "meso_bond_compute": (_i, [_vp, _i]),
    "meso_compute_ebond": (_i, [_vp, C.POINTER(_d)]),
    "meso_angles_upload": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "meso_angle_style_harmonic": (_i, [_vp, _i]),
    "meso_angle_coeff": (_i, [_vp, _i, _d, _d]),
    "meso_angle_compute": (_i, [_vp, _i]),
    "meso_compute_eangle": (_i, [_vp, C.POINTER(_d)]),
    "meso_timestep": (_i, [_vp, _d]),
    "meso_setup": (_i, [_vp]),
    "meso_run": (_i, [_vp, _i]),
    "meso_nve_initial": (_i, [_vp]),
    "meso_nve_final": (_i, [_vp]),
    "meso_neighbor_decide": (_i, [_vp, C.POINTER(_i)]),
    "meso_reneighbor": (_i, [_vp]),
    "meso_halo_forward": (_i, [_vp]),
    "meso_force_clear": (_i, [_vp, _i]),
    "meso_pair_compute": (_i, [_vp, _i, _i, _i]),
    "meso_step_advance": (_i, [_vp, _i64]),
    "meso_compute_temp": (_i, [_vp, C.POINTER(_d)]),
    "meso_compute_pe": (_i, [_vp, C.POINTER(_d)]),
    "meso_compute_pressure": (_i, [_vp, C.POINTER(_d)]),
    "meso_neigh_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_d), C.POINTER(_i64)]),
    "meso_neigh_download": (_i, [_vp, _vp, _vp, _i]),
    "meso_neigh_parts": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), _vp, _vp, _vp, _vp, _i]),
    "meso_merged_download": (_i, [_vp, _vp, _vp, _i]),
    "meso_timer_reset": (_i, [_vp]),
    "meso_timer_get": (_i, [_vp, _cp, C.POINTER(_d), C.POINTER(_i64)]),
    "meso_ntimestep": (_i64, [_vp]),
    "meso_test_tea": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "meso_test_gaussian": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "meso_test_logistic": (_i, [_vp, _i, _vp, _vp, _vp]),
    "meso_write_restart": (_i, [_vp, C.c_char_p]),
    "meso_read_restart": (_i, [_vp, C.c_char_p]),
    "meso_profile_window": (_i, [_vp, _i, C.c_int64, C.c_int64]),
    "meso_seed_now": (_u32, [_i, _i64]),
    "meso_script_run": (_i, [_vp, _cp, _cp, _cp, _cp, _sz]),
}

_lib = None


def load():
    """Load libmeso_hip.so and attach the prototypes; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmeso_hip.so is missing (%s): build it with `python -m meso_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("MESO_LIB") and not hasattr(lib, name):
                continue              # A/B timing against an OLDER build (tools/ab_boxes.sh): entry points added since are simply absent
            fn = getattr(lib, name)   # AttributeError if the .so does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
