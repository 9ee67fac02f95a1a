"""ctypes binding of libmmmusig_hip.so (include/mmmusig.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible, loading / context
creation raises.  `build()` compiles the library in-tree with hipcc (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMM_LIB_PATH") or os.path.join(_HERE, "lib", "libmmmusig_hip.so")
_LIB = None

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
vp = C.c_void_p


class MmmError(RuntimeError):
    pass


class SolverOpts(C.Structure):
    _fields_ = [("xtol_rel", C.c_double), ("xtol_abs", C.c_double), ("nu_lower", C.c_double),
                ("xtol_rule", C.c_int), ("max_eval", C.c_int)]


def build(force=False, jobs=4):
    """hipcc --offload-arch=gfx950 build of every HIP translation unit into lib/libmmmusig_hip.so."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-s", "-C", csrc, "-j%d" % jobs, "all"]
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


# symbol -> (restype, argtypes); every symbol include/mmmusig.h declares is listed here
_SIGS = {
    "mmm_version": (C.c_int, []),
    "mmm_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "mmm_ctx_destroy": (C.c_int, [vp]),
    "mmm_last_error": (C.c_char_p, [vp]),
    "mmm_ctx_synchronize": (C.c_int, [vp]),
    "mmm_ctx_stream": (vp, [vp]),
    "mmm_ctx_device_name": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "mmm_ctx_profile_begin": (C.c_int, [vp]),
    "mmm_ctx_profile_repeat": (C.c_int, [vp, C.c_int]),
    "mmm_ctx_profile_select": (C.c_int, [vp, C.c_int]),
    "mmm_ctx_profile_end_phases": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mmm_ctx_profile_end": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mmm_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mmm_comm_init_rank": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p]),
    "mmm_comm_transport": (C.c_char_p, [vp]),
    "mmm_p2p_local_handle": (C.c_int, [vp, C.c_int, C.c_char_p]),
    "mmm_p2p_attach": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p]),
    "mmm_p2p_selftest": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "mmm_p2p_enable": (C.c_int, [vp, C.c_int]),
    "mmm_comm_nranks": (C.c_int, [vp]),
    "mmm_lda_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i64p, vp, vp, f64p, C.POINTER(vp)]),
    "mmm_ilda_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, i32p, f64p, i32p, i64p, vp, vp, f64p, C.POINTER(vp)]),
    "mmm_lda_destroy": (C.c_int, [vp]),
    "mmm_lda_get": (C.c_int, [vp, C.c_int, f64p, C.c_size_t]),
    "mmm_lda_set": (C.c_int, [vp, C.c_int, f64p, C.c_size_t]),
    "mmm_lda_set_hyper": (C.c_int, [vp, C.c_double, f64p, C.c_int]),
    "mmm_lda_update_gamma": (C.c_int, [vp]),
    "mmm_lda_update_phi": (C.c_int, [vp]),
    "mmm_lda_update_lambda": (C.c_int, [vp]),
    "mmm_lda_update_beta": (C.c_int, [vp]),
    "mmm_lda_update_theta": (C.c_int, [vp]),
    "mmm_lda_loglik": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "mmm_lda_elbo": (C.c_int, [vp, C.POINTER(C.c_double), vp]),
    "mmm_lda_iterate": (C.c_int, [vp, C.c_int]),
    "mmm_lda_ll_history": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "mmm_lda_geometry": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "mmm_lda_row_bytes": (C.c_int, [vp]),
    "mmm_lda_prologue_moved": (C.c_int, [vp]),
    "mmm_lda_fit": (C.c_int, [vp, C.c_int, C.c_double, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mmm_solver_opts_default": (None, [C.POINTER(SolverOpts)]),
    "mmm_lda_infer": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmm_ctm_create": (C.c_int, [vp, C.c_int, C.c_int, i32p, i32p, f64p, i64p, vp, vp, vp, vp, vp, f64p, C.POINTER(SolverOpts), C.POINTER(vp)]),
    "mmm_ctm_destroy": (C.c_int, [vp]),
    "mmm_ctm_get": (C.c_int, [vp, C.c_int, f64p, C.c_size_t]),
    "mmm_ctm_set": (C.c_int, [vp, C.c_int, f64p, C.c_size_t]),
    "mmm_ctm_update_zeta": (C.c_int, [vp]),
    "mmm_ctm_update_theta": (C.c_int, [vp]),
    "mmm_ctm_update_nu": (C.c_int, [vp]),
    "mmm_ctm_update_lambda": (C.c_int, [vp]),
    "mmm_ctm_update_mu": (C.c_int, [vp]),
    "mmm_ctm_update_Sigma": (C.c_int, [vp]),
    "mmm_ctm_update_gamma": (C.c_int, [vp]),
    "mmm_ctm_update_Elnphi": (C.c_int, [vp]),
    "mmm_ctm_update_props": (C.c_int, [vp]),
    "mmm_ctm_update_alpha": (C.c_int, [vp]),
    "mmm_ctm_update_phi": (C.c_int, [vp]),
    "mmm_ctm_loglik": (C.c_int, [vp, f64p]),
    "mmm_ctm_elbo": (C.c_int, [vp, C.POINTER(C.c_double), vp]),
    "mmm_ctm_objectives": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), f64p, C.POINTER(C.c_double), f64p]),
    "mmm_ctm_geometry": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "mmm_debug_math": (C.c_int, [vp, C.c_int, C.c_size_t, f64p, vp, f64p]),
    "mmm_ctm_solver_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp, vp]),
    "mmm_ctm_events": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "mmm_lda_events": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "mmm_ctm_iterate": (C.c_int, [vp, C.c_int, C.c_int]),
    "mmm_ctm_ll_history": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "mmm_ctm_fit": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mmm_ctm_infer": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmm_ctm_create_batch": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i32p, i32p, f64p, i64p, vp, vp, vp, vp, vp, f64p, C.POINTER(SolverOpts), C.POINTER(vp)]),
    "mmm_ctm_replicas": (C.c_int, [vp]),
    "mmm_ctm_select": (C.c_int, [vp, C.c_int]),
    "mmm_ctm_fit_batch": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]),
    "mmm_tuning_opts_default": (None, [vp]),
    "mmm_ctx_set_tuning": (C.c_int, [vp, vp]),
    "mmm_ctx_get_tuning": (C.c_int, [vp, vp]),
    "mmm_lda_update_Elntheta": (C.c_int, [vp]),
    "mmm_lda_update_Elnbeta": (C.c_int, [vp]),
    "mmm_ctm_update_doc": (C.c_int, [vp, C.c_int, C.c_int]),
    "mmm_ctm_doc_sums": (C.c_int, [vp, C.c_int, vp, vp]),
    "mmm_lambda_objective": (C.c_int, [vp, C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, C.POINTER(C.c_double), vp]),
    "mmm_nu_objective": (C.c_int, [vp, C.c_int, f64p, f64p, f64p, vp, f64p, C.POINTER(C.c_double), vp]),
    "mmm_alpha_objective": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mmm_mixture_loglik": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64p, vp, vp, f64p, f64p, C.POINTER(C.c_double)]),
    "mmm_mixture_loglik_features": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i64p, vp, vp, f64p, C.c_int, f64p, C.POINTER(C.c_double)]),
}


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load the shared library (raises MmmError if it has not been built)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise MmmError("libmmmusig_hip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C multimodalmusig.jl_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)      # AttributeError here = the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(rc, ctx_handle=None, what=""):
    if rc != 0:
        msg = lib().mmm_last_error(ctx_handle)
        raise MmmError("%s failed with status %d: %s" % (what or "libmmmusig_hip call", rc, (msg or b"").decode(errors="replace")))


_ALL_CTX = []


class TuningOpts(C.Structure):
    """mmm_tuning_opts (include/mmmusig.h)"""
    _fields_ = [("lda_build", C.c_int), ("ctm_build", C.c_int), ("geometry_cus", C.c_int), ("grid_blocks", C.c_int), ("waves_per_block", C.c_int),
                ("moment_blocks", C.c_int), ("side_stream", C.c_int), ("resident_cap", C.c_int), ("disable", C.c_uint), ("solve_lanes", C.c_int), ("solve_waves", C.c_int), ("reserved", C.c_int * 5)]


BUILDS = {"auto": 0, "sparse": 1, "dense": 2, "wide": 3}
OFF = {"lda_padded_rows": 1 << 0, "lda_count_rows": 1 << 1, "lda_rows16": 1 << 2, "lda_ll_join": 1 << 3, "lda_merged": 1 << 4, "p2p_folded": 1 << 5,
       "ctm_packed": 1 << 6, "ctm_cpl": 1 << 7, "ctm_kfit": 1 << 8, "ctm_fused_gauss": 1 << 9, "ctm_ll_rows": 1 << 10, "lda_early_prologue": 1 << 11, "ctm_pipe_gauss": 1 << 12, "ctm_solve_order": 1 << 13}


class Context:
    """One GPU + one HIP stream (+ one RCCL rank).  Mirrors `mmm_ctx`."""

    def __init__(self, device=0):
        self.h = vp()
        rc = lib().mmm_ctx_create(int(device), C.byref(self.h))
        if rc != 0:
            msg = lib().mmm_last_error(None)
            raise MmmError("mmm_ctx_create(device=%d) failed with status %d: %s -- the HIP path is the only path; "
                           "no CPU fallback exists" % (device, rc, (msg or b"").decode(errors="replace")))
        self.device = int(device)
        self.nranks, self.rank = 1, 0
        _ALL_CTX.append(self)

    def synchronize(self):
        check(lib().mmm_ctx_synchronize(self.h), self.h, "mmm_ctx_synchronize")

    def set_tuning(self, lda_build="auto", ctm_build="auto", geometry_cus=0, grid_blocks=0, waves_per_block=0, moment_blocks=0, side_stream=0,
                   resident_cap=0, disable=(), solve_lanes=0, solve_waves=0):
        """mmm_ctx_set_tuning: the caller's choices for the handles created on this context FROM NOW ON (no arguments: the defaults).
        lda_build / ctm_build: "auto" | "sparse" | "dense" | "wide"; geometry_cus: size the launches as if the device had that many CUs
        (pins the association of the cross-document sums, and so the bits of a fit, across devices); disable: names of OFF."""
        t = TuningOpts()
        t.lda_build, t.ctm_build = BUILDS[lda_build], BUILDS[ctm_build]
        t.geometry_cus, t.grid_blocks, t.waves_per_block, t.moment_blocks = int(geometry_cus), int(grid_blocks), int(waves_per_block), int(moment_blocks)
        t.side_stream, t.resident_cap = int(side_stream), int(resident_cap)
        t.solve_lanes, t.solve_waves = int(solve_lanes), int(solve_waves)
        d = 0
        for name in ([disable] if isinstance(disable, str) else disable):
            d |= OFF[name]
        t.disable = d
        check(lib().mmm_ctx_set_tuning(self.h, C.byref(t)), self.h, "mmm_ctx_set_tuning")
        return self

    def get_tuning(self):
        t = TuningOpts()
        check(lib().mmm_ctx_get_tuning(self.h, C.byref(t)), self.h, "mmm_ctx_get_tuning")
        return t

    @property
    def stream(self):
        return lib().mmm_ctx_stream(self.h)

    def device_name(self):
        b = C.create_string_buffer(64)
        check(lib().mmm_ctx_device_name(self.h, b, 64), self.h)
        return b.value.decode()

    def profile_begin(self, repeat=1, phase=0):
        check(lib().mmm_ctx_profile_repeat(self.h, int(repeat)), self.h, "mmm_ctx_profile_repeat")
        check(lib().mmm_ctx_profile_select(self.h, int(phase)), self.h, "mmm_ctx_profile_select")
        check(lib().mmm_ctx_profile_begin(self.h), self.h, "mmm_ctx_profile_begin")

    def profile_end(self):
        """-> (number of dominant-kernel launches, sum of their HIP-event durations in ms)"""
        n = C.c_int(); ms = C.c_double()
        check(lib().mmm_ctx_profile_end(self.h, C.byref(n), C.byref(ms)), self.h, "mmm_ctx_profile_end")
        return n.value, ms.value

    def profile_end_phases(self):
        """after profile_begin(phase=8): {phase: (number of spans, summed milliseconds)} of the phases that were launched"""
        n = (C.c_int * 8)(); ms = (C.c_double * 8)()
        check(lib().mmm_ctx_profile_end_phases(self.h, n, ms), self.h, "mmm_ctx_profile_end_phases")
        return {i: (n[i], ms[i]) for i in range(8) if n[i]}

    def init_comm(self, nranks, rank, unique_id):
        check(lib().mmm_comm_init_rank(self.h, int(nranks), int(rank), unique_id), self.h, "mmm_comm_init_rank")
        self.nranks, self.rank = int(nranks), int(rank)

    @property
    def transport(self):
        """'p2p' (xGMI mailboxes), 'rccl' or 'none': what the per-iteration all-reduce uses."""
        return lib().mmm_comm_transport(self.h).decode()

    def p2p_enable(self, on):
        """Switch the per-iteration all-reduce between the xGMI mailboxes (True) and ncclAllReduce (False).  Collective in effect: every rank
        must make the same call before its next model call (mmm_p2p_enable)."""
        check(lib().mmm_p2p_enable(self.h, 1 if on else 0), self.h, "mmm_p2p_enable")

    def init_p2p(self, nranks, rank, allgather, allmin):
        """Mailbox all-reduce without an RCCL communicator.  allgather(bytes) -> list of every rank's bytes (rank order),
        allmin(int) -> minimum over ranks; both are collective calls of the host's own transport (gloo, MPI, ...)."""
        nranks, rank = int(nranks), int(rank)
        mine = C.create_string_buffer(64)
        check(lib().mmm_p2p_local_handle(self.h, nranks, mine), self.h, "mmm_p2p_local_handle")
        handles = b"".join(allgather(mine.raw))
        ok = 1
        try:
            check(lib().mmm_p2p_attach(self.h, nranks, rank, handles), self.h, "mmm_p2p_attach")
        except MmmError:
            ok = 0
        if not allmin(ok):
            if ok:
                lib().mmm_p2p_enable(self.h, 0)
            raise MmmError("p2p mailboxes could not be mapped on every rank")
        good = C.c_int(0)
        check(lib().mmm_p2p_selftest(self.h, C.byref(good)), self.h, "mmm_p2p_selftest")
        if not allmin(good.value):
            lib().mmm_p2p_enable(self.h, 0)
            raise MmmError("p2p all-reduce rehearsal failed on some rank")
        self.nranks, self.rank = nranks, rank

    def close(self):
        if self.h:
            lib().mmm_ctx_destroy(self.h)
            self.h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    b = C.create_string_buffer(128)
    check(lib().mmm_comm_unique_id(b), None, "mmm_comm_unique_id")
    return b.raw


_DEFAULT_CTX = {}
_LIVE = None


def track(model):
    """Remember a live model handle so that it is destroyed (before its context) at interpreter exit, while the
    HIP runtime is still up."""
    global _LIVE
    import atexit
    import weakref
    if _LIVE is None:
        _LIVE = weakref.WeakSet()
        atexit.register(_shutdown)
    _LIVE.add(model)


def _shutdown():
    for m in list(_LIVE or ()):
        try:
            m.close()
        except Exception:
            pass
    for c in list(_ALL_CTX):
        try:
            c.close()
        except Exception:
            pass


def default_context(device=0):
    c = _DEFAULT_CTX.get(device)
    if c is None or not c.h:
        c = _DEFAULT_CTX[device] = Context(device)
    return c
