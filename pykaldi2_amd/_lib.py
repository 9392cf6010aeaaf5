"""ctypes binding of libpk2hip.so (C ABI declared in include/pk2hip.h).

There is no CPU fallback: if the shared library is missing or a call fails the
error is raised.  torch is imported first so the library resolves
libamdhip64.so.7 to the HIP runtime PyTorch already loaded (one runtime, one
set of streams per process).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: loads the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PK2_LIB") or os.path.join(_HERE, "libpk2hip.so")   # PK2_LIB: experiment builds


class Pk2Error(RuntimeError):
    pass


class NumBatch(C.Structure):
    _fields_ = [("arc_src", C.c_void_p), ("arc_dst", C.c_void_p), ("arc_pdf", C.c_void_p),
                ("arc_weight", C.c_void_p), ("frame_off", C.c_void_p), ("state_off", C.c_void_p),
                ("final_state", C.c_void_p), ("final_weight", C.c_void_p), ("final_off", C.c_void_p),
                ("total_arcs", C.c_int64), ("max_seq_arcs", C.c_int64)]


class DecoderOpts(C.Structure):
    """pk2_decoder_opts"""
    _fields_ = [("beam", C.c_float), ("lattice_beam", C.c_float), ("beam_delta", C.c_float),
                ("acoustic_scale", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("tokens_per_frame", C.c_int32), ("links_per_frame", C.c_int32)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/pk2hip.h one to one
SIGNATURES = {
    "pk2_last_error": (C.c_char_p, []),
    "pk2_version": (C.c_int, []),
    "pk2_den_graph_create": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "pk2_den_graph_from_openfst": (C.c_int, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "pk2_den_graph_destroy": (C.c_int, [_vp]),
    "pk2_den_graph_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "pk2_den_graph_initial_probs": (C.c_int, [_vp, _vp]),
    "pk2_den_graph_arcs_per_lane": (_i32, []),
    "pk2_den_graph_path": (_i32, [_vp, _i32]),
    "pk2_den_graph_persist_form": (_i32, [_vp, _i32]),
    "pk2_den_graph_debug_ordering": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i32), _vp, _vp,
                                               _vp, _vp, _vp, _vp]),
    "pk2_den_graph_debug_virtual": (C.c_int, [_vp, C.c_int, C.POINTER(_i32), _vp, _vp, _vp, _vp, _vp,
                                              C.POINTER(_i64), _vp, _vp, _vp, C.POINTER(_i32), _vp, _vp]),
    "pk2_den_graph_debug_persist": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pk2_den_graph_debug_persist2": (C.c_int, [_vp, C.c_int] + [_vp] * 18),
    "pk2_chain_workspace_bytes": (_sz, [_vp, _i32, _i32, _i64]),
    "pk2_chain_objf_and_deriv": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, C.POINTER(NumBatch), _f32,
                                           _f32, _f32, _f32, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "pk2_chain_objf_and_deriv_op": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, C.POINTER(NumBatch), _f32,
                                              _f32, _f32, _f32, _vp, _i64, _i64, _vp, _vp, _sz, _f32, _vp, _vp]),
    "pk2_chain_den_fwd_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, _f32, _vp, _vp, _i64, _i64,
                                        _vp, _sz, _vp]),
    "pk2_chain_debug_flags": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "pk2_split_to_phones": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "pk2_sup_model_create": (_vp, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp,
                                   _vp, _vp, _i32, _vp]),
    "pk2_sup_model_destroy": (None, [_vp]),
    "pk2_sup_model_pdf": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "pk2_supervision_create": (_vp, [_vp, _vp, _vp, _i32, _i32, _i32, _i32]),
    "pk2_supervision_destroy": (None, [_vp]),
    "pk2_supervision_sizes": (None, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                     C.POINTER(_i32)]),
    "pk2_supervision_copy": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pk2_decode_graph_create": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pk2_decode_graph_create_words": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pk2_decode_graph_link_words": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pk2_decode_graph_from_openfst": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pk2_decode_graph_destroy": (C.c_int, [_vp]),
    "pk2_decode_graph_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32)]),
    "pk2_lattice_batch_create": (C.c_int, [_vp, _vp, _i32, C.POINTER(DecoderOpts), C.POINTER(_vp)]),
    "pk2_lattice_batch_bytes": (_sz, [_vp]),
    "pk2_lattice_batch_destroy": (C.c_int, [_vp]),
    "pk2_lattice_decode": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp]),
    "pk2_lattice_summary": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pk2_lattice_persist_status": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "pk2_lattice_mmi": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_double, C.c_double, _i32, _vp, _i64, _i64, _vp, _vp]),
    "pk2_lattice_mpe": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, C.c_double, C.c_double, _vp, _i64, _i64,
                                  _vp, _vp]),
    "pk2_lattice_export": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp]),
    "pk2_comm_unique_id_bytes": (_i32, []),
    "pk2_comm_unique_id": (C.c_int, [_vp]),
    "pk2_comm_init": (C.c_int, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "pk2_allreduce_bucket": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pk2_allreduce_guarded": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "pk2_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32]),
    "pk2_comm_destroy": (C.c_int, [_vp]),
    "pk2_sim_apply_rir": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _vp, _vp]),
    "pk2_sim_power": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pk2_sim_add_noise": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _f32, _vp, _vp, _vp]),
    "pk2_sim_gain_norm": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pk2_fbank_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pk2_fbank_destroy": (C.c_int, [_vp]),
    "pk2_fbank_num_frames": (_i32, [_i64]),
    "pk2_fbank_compute": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "pk2_pad_roll_subsample": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "pk2_mvn_apply": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "pk2_softmax_ce_fwd_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "pk2_softmax_ce_fwd_bwd_mean": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "pk2_scale_inplace_ratio": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "pk2_scale_by_count": (C.c_int, [_vp, _i64, _f32, _vp, _vp]),
    "pk2_scale_by_scalars": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "pk2_gemm_f32": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _f32, _vp, _i64,
                               _vp, _vp]),
    "pk2_gemm_f32_act": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _i32, _vp, _i64,
                                   _vp]),
    "pk2_gemm_f32_seg": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _i64, _vp, _i64, _i64, _f32, _vp, _i64, _vp,
                                   _i32, _vp, _i64, _vp]),
    "pk2_gemm_f32_batched": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64,
                                       _f32, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "pk2_gemm_f32_tn_colsum": (C.c_int, [_i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _vp]),
    "pk2_gemm_set_arith": (C.c_int, [_i32]),
    "pk2_gemm_get_arith": (C.c_int, []),
    "pk2_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "pk2_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "pk2_softmax_mask_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pk2_softmax_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "pk2_attention_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _f32, C.c_uint64, _vp, _vp, _vp]),
    "pk2_attention_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _f32, C.c_uint64, _vp,
                                    _vp, _vp]),
    "pk2_relu_fwd": (C.c_int, [_vp, _i64, _vp]),
    "pk2_relu_bwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pk2_add_inplace": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pk2_colsum_f32": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "pk2_lstm_fwd_workspace_floats": (_sz, [_i32, _i32, _i32]),
    "pk2_lstm_layer_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pk2_lstm_persist_status": (C.c_int, [C.POINTER(C.c_uint32)]),
    "pk2_lattice_determinize": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i64, C.POINTER(_vp)]),
    "pk2_det_lattice_sizes": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "pk2_det_lattice_export": (C.c_int, [_vp] * 12),
    "pk2_det_lattice_destroy": (None, [_vp]),
    "pk2_persist_guard_status": (C.c_int, [C.POINTER(C.c_uint32)]),
    "pk2_persist_guard_clear": (C.c_int, []),
    "pk2_persist_guard_raise": (C.c_int, [_vp]),
    "pk2_persist_guard_export": (C.c_int, [_vp, _vp]),
    "pk2_persist_guard_import": (C.c_int, [_vp, C.c_uint32, _vp]),
    "pk2_persist_guard_verdict": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "pk2_lstm_bwd_scratch_floats": (_sz, [_i32, _i32, _i32]),
    "pk2_lstm_layer_bwd_bias": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp]),
    "pk2_lstm_layer_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pk2_dropout_f32": (C.c_int, [_vp, _vp, _i64, _f32, C.c_uint64, _vp]),
    "pk2_grad_norm": (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "pk2_grad_norm_workspace_bytes": (_sz, [_i64]),
    "pk2_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32,
                                _vp, _f32, _vp]),
    "pk2_sgd_step": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _f32, _vp, _f32, _vp]),
    "pk2_stream_create_cu_mask": (C.c_int, [_i32, C.POINTER(_vp)]),
    "pk2_stream_destroy": (C.c_int, [_vp]),
    "pk2_debug_where": (C.c_int, [_vp, _i32, _vp]),
    "pk2_debug_peer_reduce": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
}

_lib = None


def lib():
    """Loads libpk2hip.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Pk2Error("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def h2d(host, device):
    """Host array / tensor -> device without synchronising the stream: over pinned memory and non-blocking.  A pageable
    copy is hipMemcpyAsync + a wait for everything the stream holds, i.e. a training loop that copies its labels or its
    waveforms that way never has the next step's launches queued while the device works on this one (measured on the
    TransformerAM step: DESIGN.md 4.3).  The pinned block goes back to torch's host allocator behind the copy's event."""
    import numpy as np
    t = torch.from_numpy(host) if isinstance(host, np.ndarray) else host
    if torch.device(device).type != "cuda" or t.is_cuda or t.numel() == 0:
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def check(status):
    if status != 0:
        msg = lib().pk2_last_error()
        raise Pk2Error("libpk2hip status %d: %s" % (status, msg.decode() if msg else ""))


def persist_guard_raised():
    """True once a persistent kernel of this process has given up on the current device (include/pk2hip.h: guard of the
    persistent kernels).  Reads a host-mapped word: no device synchronisation."""
    flag = C.c_uint32(0)
    check(lib().pk2_persist_guard_status(C.byref(flag)))
    return flag.value != 0


def check_persist_guard(where, raised=None):
    if persist_guard_raised() if raised is None else raised:
        raise Pk2Error("%s: a persistent kernel (one-launch LSTM recurrence / denominator / lattice decoder) timed out on "
                       "this device; its output was poisoned with NaN and the optimiser kernels have been leaving the "
                       "weights untouched since.  Typical causes: the GPU is shared with another process, or fewer than "
                       "256 CUs are available to the launch.  PK2_LSTM_SEQ=0 PK2_LSTM_PERSIST=0 PK2_DEN_PERSIST=0 "
                       "PK2_LAT_DECODER=frames select the launch-per-step kernels." % where)


def ptr(t):
    """Raw address of a torch tensor / numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise Pk2Error("pykaldi2_amd needs an MI355X (torch.cuda.is_available() is False); "
                       "there is no CPU fallback")
