"""ctypes binding of libjlm_hip.so (C ABI declared in include/jlm_hip.h).

The product path has no CPU fallback: :func:`lib` raises if the shared library
has not been built (``python -c 'import __graft_entry__ as g; g.build()'``) and
:func:`require_gpu` raises if no MI355X is visible.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# JLM_HIP_LIB: developer override used by tools/ab_lib.sh to A/B two builds of the same ABI
LIB_PATH = os.environ.get("JLM_HIP_LIB") or os.path.join(_HERE, "csrc", "libjlm_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "csrc", "libjlm_host.so")

JLM_MAX_SEGMENTS = 8


class Segment(Structure):
    _fields_ = [("v_start", c_int), ("v_end", c_int), ("k", c_int), ("t_off", c_int),
                ("B", c_void_p), ("ldb", c_int)]


class Lattice(Structure):
    _fields_ = [("n_sent", c_int), ("beam", c_int), ("n_frames", c_int),
                ("sent_len", c_void_p), ("end_off", c_void_p),
                ("node_start", c_void_p), ("node_word", c_void_p)]


class BeamState(Structure):
    _fields_ = [("score", c_void_p), ("lse", c_void_p), ("ysum", c_void_p),
                ("bp", c_void_p), ("node", c_void_p), ("word", c_void_p),
                ("cnt", c_void_p), ("live", c_void_p), ("n_live", c_void_p),
                ("edge", c_void_p),
                ("live_base", c_void_p), ("lse_part", c_void_p), ("ld_part", c_int), ("n_parts", c_int), ("flags", c_void_p)]


class DecodeModel(Structure):
    """jlm_decode_model (include/jlm_hip.h)."""
    _fields_ = [("segs", POINTER(Segment)), ("n_segs", c_int), ("b2", c_void_p), ("H", c_int), ("ldt", c_int),
                ("untied", c_int), ("self_norm", c_int), ("split_lstm", c_int),
                ("emb", c_void_p), ("ld_emb", c_int), ("wt", c_void_p), ("gate_bias", c_void_p), ("kpad", c_int), ("E", c_int),
                ("gate_descale", c_float), ("h_scale", c_float),
                ("wt8", c_void_p), ("xgate8", c_void_p), ("untied_split", c_void_p), ("untied_descale", c_float), ("lse_fixed_ref", c_int),
                ("pmt", c_void_p), ("pmt_split", c_void_p), ("n_t", c_int), ("t_descale", c_float),
                ("split_segs", POINTER(Segment)), ("split_t_scale", POINTER(c_float)), ("split_descale", POINTER(c_float)),
                ("split_bias_col", POINTER(c_int)),
                ("mixed_segs", POINTER(Segment)), ("mixed_t_scale", POINTER(c_float)), ("mixed_descale", POINTER(c_float)),
                ("mixed_s8", POINTER(c_float)), ("mixed_bias2", c_void_p), ("mixed_head_split", POINTER(c_int))]


class DecodePlan(Structure):
    """jlm_decode_plan (include/jlm_hip.h)."""
    _fields_ = [("kind", c_int), ("max_cands", c_int), ("h", c_void_p), ("c", c_void_p), ("T", c_void_p),
                ("g0", c_void_p), ("cidx", c_void_p), ("sidx", c_void_p),
                ("sg_word", c_void_p), ("sg_off", c_void_p), ("sg_node", c_void_p), ("edge", c_void_p),
                ("vs_words", c_void_p), ("vs_off", c_void_p), ("vs_max", c_int),
                ("di_words", c_void_p), ("di_off", c_void_p), ("di_idx", c_void_p), ("di_max", c_int),
                ("dd_words", c_void_p), ("dd_off", c_void_p), ("dd_max", c_int),
                ("run_max", c_void_p), ("run_sum", c_void_p), ("part", c_void_p), ("max_parts", c_int), ("lse_cu_share_pct", c_int),
                ("out_nodes", c_void_p), ("out_len", c_void_p), ("out_score", c_void_p), ("stride", c_int),
                ("di_wwords", c_void_p), ("sg_wword", c_void_p), ("Tm", c_void_p), ("ld_tm", c_int)]


P = c_void_p
_SIGS = {
    "jlm_abi_version": ([], c_int),
    "jlm_device_arch": ([c_int, c_char_p, c_int], c_int),
    "jlm_lstm_step": ([P, P, c_int, P, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P, P], c_int),
    "jlm_gemm_nt": ([P, c_int, P, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, P, P], c_int),
    "jlm_vocab_lse_partials": ([P, c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_int, c_int, P, P], c_int),
    "jlm_lse_combine": ([P, c_int, c_int, P, P, c_int, P, P], c_int),
    "jlm_vocab_lse_stationary": ([POINTER(Segment), c_int, P, P, c_int, P, P, c_int, c_int, c_int, P, P], c_int),
    "jlm_pack_split_f16": ([P, c_int, c_int, c_int, c_float, P, c_int, P], c_int),
    "jlm_lstm_step_xg": ([P, P, c_int, P, P, P, P, P, P, P, c_int, c_float, c_float, P, c_int, P, P], c_int),
    "jlm_vocab_lse_partials_split": ([P, c_int, c_int, c_int, P, c_int, P, P, c_float, P, c_int, c_int, c_int, P, P], c_int),
    "jlm_gemm_nt_split": ([P, c_int, P, P, c_int, P, P, c_int, P, P, c_float, c_int, c_int, c_int, P, P], c_int),
    "jlm_dequant_u8": ([P, c_int, c_int, c_int, P, c_int, P, c_int, P], c_int),
    "jlm_pack_split_f16_col": ([P, c_int, c_float, P, c_int, c_int, P], c_int),
    "jlm_vocab_lse_split": ([POINTER(Segment), POINTER(c_float), POINTER(c_float), POINTER(c_int), c_int, P, P, c_int, P, P,
                             c_int, c_int, c_int, P, P], c_int),
    "jlm_edge_logits": ([POINTER(Segment), c_int, P, P, c_int, P, P, P, P, P, P, c_int, P, P, c_int, c_int, P], c_int),
    "jlm_wordlist_lse": ([POINTER(Segment), c_int, P, P, c_int, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, P],
                         c_int),
    "jlm_edge_logits_perm": ([POINTER(Segment), c_int, P, P, c_int, P, P, P, P, P, P, P, c_int, P, P, c_int, c_int, P], c_int),
    "jlm_wordlist_lse_perm": ([POINTER(Segment), c_int, P, P, c_int, P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, P],
                              c_int),
    "jlm_wordlist_lse_split": ([POINTER(Segment), c_float, c_float, P, P, c_int, P, P, P, P, P, P, c_int, c_int, P, P, P,
                               c_int, c_int, c_int, P], c_int),
    "jlm_wordlist_merge_split": ([POINTER(Segment), c_float, c_float, P, P, c_int, P, c_int, c_int, c_int, P, P, c_int, c_int,
                                 P, P, P, P], c_int),
    "jlm_beam_step": ([POINTER(Lattice), POINTER(BeamState), c_int, c_int, c_int, P], c_int),
    "jlm_beam_step_max_cands": ([c_int, c_int, c_int], c_int),
    "jlm_pack_mixed": ([P, c_int, c_int, c_int, P, c_float, c_float, c_float, P, c_int, P], c_int),
    "jlm_vocab_lse_hybrid": ([POINTER(Segment), POINTER(c_float), POINTER(c_float), POINTER(c_int), POINTER(Segment), POINTER(c_float),
                              POINTER(c_float), POINTER(c_int), c_int, P, P, c_int, P, c_int, P, P, c_int, c_int, c_int, P, P], c_int),
    "jlm_mixed_t_stride": ([POINTER(Segment), c_int], c_int),
    "jlm_pack_t_mixed": ([POINTER(Segment), POINTER(c_float), c_int, P, c_int, P, c_int, P, P, c_int, P], c_int),
    "jlm_pack_t_mixed6": ([POINTER(Segment), POINTER(c_float), c_int, P, c_int, P, c_int, P, P, c_int, P], c_int),
    "jlm_vocab_lse_mixed": ([POINTER(Segment), POINTER(c_float), POINTER(c_float), P, c_int, P, c_int, P, c_int, c_int, c_int, P, P],
                            c_int),
    "jlm_backtrace": ([POINTER(Lattice), POINTER(BeamState), P, P, P, c_int, P], c_int),
    "jlm_vocab_lse_mixed_fr": ([POINTER(Segment), POINTER(c_float), POINTER(c_float), P, c_int, P, c_int, P, c_int, c_int, c_int, P, P],
                            c_int),
    "jlm_softmax_rows": ([P, P, c_int, c_int, c_int, c_int, P], c_int),
    "jlm_decode_frames": ([POINTER(DecodeModel), POINTER(DecodePlan), POINTER(Lattice), POINTER(BeamState), P, P, P], c_int),
    "jlm_lse_probe": ([POINTER(DecodeModel), P, P, P, c_int, c_int, P, P, P, P, c_int, c_int, P, c_int, P], c_int),
}
EXPORTS = sorted(_SIGS)

_lib = None


class JlmHipError(RuntimeError):
    pass


def lib():
    """The loaded shared library (argtypes set).  Raises when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JlmHipError(
                "libjlm_hip.so is not built (%s).  Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % LIB_PATH)
        # torch first: its bundled HIP runtime must be the one in the process before this
        # library (linked against libamdhip64) is mapped, or the two runtimes disagree
        # about the device (hipErrorNoDevice on the first launch).
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        if l.jlm_abi_version() != 11:
            raise JlmHipError("libjlm_hip.so ABI version mismatch")
        _lib = l
    return _lib


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise JlmHipError("no GPU visible: jlm_amd runs only on MI355X (gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def check(rc, what):
    if rc != 0:
        raise JlmHipError("%s failed with code %d" % (what, rc))
