"""ctypes binding of libldmseg_hip.so (the C ABI declared in include/ldmseg_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build it with ``python -m ldmseg_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDMSEG_HIP_LIB: developer override used for same-box A/B runs of two builds; the shipped path is in-tree
LIB_PATH = os.environ.get("LDMSEG_HIP_LIB") or os.path.join(_HERE, "libldmseg_hip.so")

F32, BF16, BF16X3 = 0, 1, 2
PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("cross_attention", C.c_int32),
                ("compute_dtype", C.c_int32), ("device", C.c_int32)]


class VAECfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("int_channels", C.c_int32), ("out_channels", C.c_int32),
                ("latent_channels", C.c_int32), ("num_latents", C.c_int32), ("num_upscalers", C.c_int32),
                ("upscale_channels", C.c_int32), ("norm_num_groups", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("compute_dtype", C.c_int32), ("device", C.c_int32)]


class VAEImageCfg(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("device", C.c_int32)]


class SampleCfg(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("timesteps", C.POINTER(C.c_int64)), ("coef", C.POINTER(C.c_float)),
                ("prediction_type", C.c_int32), ("clip_sample", C.c_int32), ("self_condition", C.c_int32),
                ("clip_sample_range", C.c_float),
                ("known_dev", C.c_void_p), ("z0_dev", C.c_void_p), ("noise_dev", C.c_void_p),
                ("paste_coef", C.POINTER(C.c_float))]


_vp, _i, _i64, _f, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double

# name -> (restype, argtypes); mirrors include/ldmseg_hip.h and include/ldmseg_hip_ops.h
SIGNATURES = {
    "ldmseg_unet_create": (_i, [C.POINTER(UNetCfg), _i, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp)]),
    "ldmseg_unet_destroy": (None, [_vp]),
    "ldmseg_unet_forward": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "ldmseg_unet_forward_parts": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "ldmseg_unet_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ldmseg_unet_num_params": (_i64, [_vp]),
    "ldmseg_vae_create": (_i, [C.POINTER(VAECfg), _i, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp)]),
    "ldmseg_vae_destroy": (None, [_vp]),
    "ldmseg_vae_decode": (_i, [_vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    "ldmseg_vae_decode_argmax": (_i, [_vp, _vp, _f, _i, _i, _f, _i64, _vp, _vp, _vp]),
    "ldmseg_vae_encode": (_i, [_vp, _vp, _f, _f, _i, _i, _vp, _vp]),
    "ldmseg_vae_posterior": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp]),
    "ldmseg_vae_num_params": (_i64, [_vp]),
    "ldmseg_ddim_step": (_i, [_vp, _vp, _f, _f, _f, _f, _i, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "ldmseg_add_noise": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _i, _sz, _vp]),
    "ldmseg_remove_noise": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _i, _sz, _vp]),
    "ldmseg_unet_reserve": (_i, [_vp, _i, _i]),
    "ldmseg_unet_set_attention_fp8": (_i, [_vp, _i]),
    "ldmseg_unet_gn_fallbacks": (_i, [_vp, _vp]),
    "ldmseg_unet_gn_backoff": (_i, [_vp, _vp]),
    "ldmseg_unet_cf_fallbacks": (_i, [_vp, _vp]),
    "ldmseg_sample_loop": (_i, [_vp, C.POINTER(SampleCfg), _vp, _vp, _i, _i, _vp, _vp]),
    "ldmseg_vae_image_create": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ldmseg_vae_image_destroy": (None, [_vp]),
    "ldmseg_vae_image_num_params": (_i64, [_vp]),
    "ldmseg_vae_image_encode": (_i, [_vp, _vp, _f, _f, _i, _i, _i, _vp, _vp]),
    "ldmseg_vae_decode_panoptic": (_i, [_vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _i, _d, _i64, _vp, _vp, _vp, _vp, _vp,
                                        _vp]),
    "ldmseg_panoptic_postprocess": (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _i, _d, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ldmseg_bit_encode": (_i, [_vp, _i, _i, _i, _i64, _f, _f, _f, _vp, _vp, _vp]),
    "ldmseg_bit_decode": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ldmseg_last_error": (C.c_char_p, []),
    "ldmseg_version": (C.c_char_p, []),
    "ldmseg_profile_enable": (_i, [_i]),
    "ldmseg_profile_read": (_i, [_i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ldmseg_profile_reset": (_i, []),
    "ldmseg_profile_dump": (_i, [C.c_char_p]),
    "ldmseg_debug_set": (_i, [_i, _i]),
    "ldmseg_debug_get": (_i, [_i]),
    # include/ldmseg_hip_ops.h
    "ldmseg_op_conv2d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_groupnorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "ldmseg_op_layernorm": (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp]),
    "ldmseg_op_attention": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_attention_fp8": (_i, [_vp, _i, _i, _i, _i, _vp, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_op_convt2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_bilinear2x": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_chained_ff_out": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_igemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ldmseg_op_conv3x3_plus_1x1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_op_ln_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "ldmseg_op_conv_out_tail": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, C.POINTER(C.c_float), _i, _i, _f, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _f, _f, _vp, _vp]),
    "ldmseg_op_transformer_ff": (_i, [_vp] * 10 + [_i, _i, _f, _i, _i, _vp, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_op_transformer_in": (_i, [_vp] * 8 + [_i, _i, _f, _i, _i, _vp, _vp, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_op_gn_transformer_in": (_i, [_vp] * 3 + [_f, _i, _i] + [_vp] * 7 + [_i, _i, _f, _i, _i, _vp, _vp, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_bench_igemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                C.POINTER(C.c_float), _vp]),
    "ldmseg_bench_attention": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_op_panoptic_from_decoder": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _i, _d, _i64, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp]),
    "ldmseg_op_conv_groupnorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp]),
    "ldmseg_bench_groupnorm": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), _vp]),
    "ldmseg_igemm_last_kernel": (_i, [C.c_char_p, _i]),
    "ldmseg_op_fastdiv": (_i, [_vp, _i, _i, _vp, _vp]),
    "ldmseg_igemm_log": (_i, [_i]),
    "ldmseg_igemm_log_read": (_i, [C.c_char_p, _i]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Run `python __graft_entry__.py build` or `python -m ldmseg_amd.build`.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("LDMSEG_HIP_LIB") and not hasattr(h, name):
                continue                   # an older build loaded for an A/B measurement may lack newer test hooks
            fn = getattr(h, name)          # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().ldmseg_last_error()
        raise RuntimeError(f"libldmseg_hip: {what} failed with code {code}: {msg.decode() if msg else ''}")


def igemm_last_kernel():
    """Instantiation + plan of the most recent igemm launch, e.g. 'igemm<bf16,256,160,4,2,3,1,0> splits=1 grid=256'."""
    buf = C.create_string_buffer(128)
    lib().ldmseg_igemm_last_kernel(buf, 128)
    return buf.value.decode()


def igemm_log(enable):
    lib().ldmseg_igemm_log(1 if enable else 0)


def igemm_log_read():
    """Set of distinct igemm instantiations launched since igemm_log(True)."""
    buf = C.create_string_buffer(1 << 16)
    check(lib().ldmseg_igemm_log_read(buf, 1 << 16), "ldmseg_igemm_log_read")
    return {ln for ln in buf.value.decode().split("\n") if ln}


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (kernels run where torch ops would)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_cuda_f32(t, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on the MI355X (got {type(t).__name__}"
                           f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}): "
                           "the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def weight_arrays(state_dict, device):
    """names/pointers/numels arrays for *_create from a torch state dict (moved to `device` fp32)."""
    import torch
    keep = []
    names = []
    for k, v in state_dict.items():
        t = v.detach().to(device=device, dtype=torch.float32).contiguous()
        keep.append(t)
        names.append(k.encode())
    n = len(keep)
    c_names = (C.c_char_p * n)(*names)
    c_ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in keep])
    c_numels = (C.c_int64 * n)(*[t.numel() for t in keep])
    return n, c_names, c_ptrs, c_numels, keep
