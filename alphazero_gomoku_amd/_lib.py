"""ctypes binding of libazgomoku.so (include/azgomoku.h).  There is no CPU fallback:
if the HIP library is missing or a call fails, an exception is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AZG_LIBRARY selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("AZG_LIBRARY") or os.path.join(_HERE, "libazgomoku.so")

PRIOR_NET, PRIOR_UNIFORM32, PRIOR_HASH, PRIOR_PEAKED = 0, 1, 2, 3
LAYOUT_TORCH, LAYOUT_THEANO = 0, 1
CNN_F16X3, CNN_F32 = 0, 1

ERROR_NAMES = {-1: "AZG_EINVAL", -2: "AZG_EHIP", -3: "AZG_ENOWEIGHTS", -4: "AZG_EARENA",
               -5: "AZG_ESTATE", -6: "AZG_EMOVE", -7: "AZG_ERANGE"}

# every symbol include/azgomoku.h declares
SYMBOLS = ["azg_create", "azg_destroy", "azg_last_error", "azg_set_weights", "azg_reset_games",
           "azg_set_position", "azg_search", "azg_root_stats", "azg_sample_and_advance",
           "azg_advance", "azg_update_with_move", "azg_status", "azg_states",
           "azg_drain_samples", "azg_eval_batch", "azg_board_step_batch", "azg_profile", "azg_search_parts", "azg_set_search_parts",
           "azg_stream", "azg_selfplay_begin", "azg_selfplay_step", "azg_selfplay_fetch",
           "azg_selfplay_end", "azg_equi_data", "azg_pure_search", "azg_host_timing", "azg_arena_stats", "azg_reroot_stats", "azg_selfplay_fetch_packed", "azg_unpack_states", "azg_playout_begin", "azg_playout_end",
           "azg_board_step_batch_static", "azg_gate_stats", "azg_gate_poke",
           "azg_train_param_count", "azg_train_forward", "azg_train_backward", "azg_train_adam", "azg_train_step", "azg_train_epoch"]


class AzgError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "AZG_E?"), code, message))
        self.code = code


class Config(C.Structure):
    _fields_ = [("board_n", C.c_int32), ("n_in_row", C.c_int32), ("n_playout", C.c_int32),
                ("n_games", C.c_int32), ("c_puct", C.c_double), ("device", C.c_int32),
                ("prior_mode", C.c_int32), ("arena_nodes", C.c_int32), ("cnn_mode", C.c_int32),
                ("big_arenas", C.c_int32)]


_lib = None


def _load_hip_runtime():
    """Make exactly one HIP runtime visible before libazgomoku.so resolves its hip* symbols.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so; a second
    runtime from /opt/rocm in the same process would own separate device contexts (device
    pointers of one are unknown to the other, and whichever initialises second may see no
    GPU).  So: use PyTorch's copy when PyTorch is installed, else the system one."""
    import importlib.util
    candidates = []
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so"]
    errors = []
    for c in candidates:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:          # keep looking, report all at the end
            errors.append("%s: %s" % (c, e))
    raise ImportError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errors))


def load():
    """Load the shared library (raises if it was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libazgomoku.so not found at %s: build it with alphazero_gomoku_amd/csrc/build.sh "
            "(hipcc --offload-arch=gfx950); this package has no CPU fallback" % LIB_PATH)
    _load_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32p, f32p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
    u8p, i8p, u64p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    L.azg_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.azg_destroy.argtypes = [vp]
    L.azg_destroy.restype = None
    L.azg_last_error.argtypes = [vp]
    L.azg_last_error.restype = C.c_char_p
    L.azg_set_weights.argtypes = [vp, C.POINTER(vp), C.c_int]
    L.azg_reset_games.argtypes = [vp, u8p, C.POINTER(C.c_uint32), i64p]
    L.azg_set_position.argtypes = [vp, C.c_int, u64p, u64p, C.c_int, C.c_int]
    L.azg_search.argtypes = [vp]
    L.azg_root_stats.argtypes = [vp, vp, vp, vp, vp]
    L.azg_sample_and_advance.argtypes = [vp, C.c_double, i32p, f64p]
    L.azg_advance.argtypes = [vp, i32p, C.c_int, C.c_double]
    L.azg_update_with_move.argtypes = [vp, i32p]
    L.azg_status.argtypes = [vp, u8p, i8p, i32p]
    L.azg_states.argtypes = [vp, vp]
    L.azg_selfplay_begin.argtypes = [vp, C.c_double, C.c_uint32, C.c_int64, C.c_int64, C.c_int64]
    L.azg_selfplay_step.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.azg_selfplay_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    L.azg_selfplay_fetch_packed.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.azg_unpack_states.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    ip = C.POINTER(C.c_int)
    L.azg_playout_begin.argtypes = [vp, C.c_int, u64p, u64p, ip, ip, ip, ip]
    L.azg_playout_end.argtypes = [vp, C.c_int, vp, C.c_double]
    L.azg_selfplay_end.argtypes = [vp, C.POINTER(C.c_int)]
    L.azg_equi_data.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.azg_drain_samples.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
    L.azg_eval_batch.argtypes = [vp, vp, C.c_int, vp, vp]
    L.azg_board_step_batch.argtypes = [vp, C.c_int, u64p, i32p, i32p, i32p, i32p, u8p, i8p, f32p]
    L.azg_board_step_batch_static.argtypes = [vp, C.c_int, u64p, i32p, i32p, i32p, i32p, u8p, i8p, f32p]
    L.azg_host_timing.argtypes = [vp, f64p, i64p]
    L.azg_arena_stats.argtypes = [vp, i64p, i64p]
    L.azg_reroot_stats.argtypes = [vp, i64p, i64p]
    L.azg_gate_stats.argtypes = [vp, i64p, i64p]
    L.azg_gate_poke.argtypes = [vp, C.c_int]
    L.azg_train_param_count.argtypes = [vp]
    L.azg_train_forward.argtypes = [vp, C.c_int, vp, C.POINTER(vp), vp, vp, vp, C.POINTER(C.c_double)]
    L.azg_train_backward.argtypes = [vp, C.c_int, vp, vp, vp, C.POINTER(vp), vp, f32p, f32p]
    L.azg_train_adam.argtypes = [vp, C.POINTER(vp), vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    L.azg_train_step.argtypes = [vp, C.c_int, vp, vp, vp, C.POINTER(vp), vp, vp, vp, C.c_int64, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_double, f32p, f32p]
    L.azg_train_epoch.argtypes = [vp, C.c_int, vp, vp, vp, C.POINTER(vp), vp, vp, vp, C.c_int64, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_double, f32p, f32p, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_double)]
    L.azg_pure_search.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32), i32p]
    L.azg_profile.argtypes = [vp, i64p, f64p, i64p]
    L.azg_search_parts.argtypes = [vp]
    L.azg_set_search_parts.argtypes = [vp, C.c_int]
    L.azg_stream.argtypes = [vp]
    L.azg_stream.restype = vp
    for name in SYMBOLS:
        if name not in ("azg_destroy", "azg_last_error", "azg_stream"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(handle, rc):
    if rc != 0:
        msg = load().azg_last_error(handle)
        raise AzgError(rc, msg.decode() if msg else "")
