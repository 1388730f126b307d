"""Build and bind libvdamd.so (the C ABI declared in include/vd_amd.h).

The product path has no CPU fallback: if the HIP library cannot be built or
loaded, every compute entry point raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libvdamd.so")
SOURCES = ["igemm.hip", "conv_wino.hip", "conv_wino_r64.hip", "conv_wino_z128.hip", "gemm_frag.hip", "gemm_split.hip", "split_pack.hip", "norm.hip", "backward.hip", "attn_spatial.hip", "attn_temporal.hip", "misc.hip", "engine.hip"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vd_amd.h")
# per-source flags on top of the common ones.  conv_wino_z128.hip: its main loop is ONE fully unrolled body of 288 MFMA slots; past
# LLVM's default size limit for `#pragma unroll` (16 k IR instructions) hipcc silently keeps the loops and indexes the register
# arrays through scratch
SOURCE_FLAGS = {"conv_wino_z128.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}


def _toolchain():
    """(hipcc, flags) of this process' build."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + \
        os.environ.get("VD_HIPCC_FLAGS", "").split()
    return hipcc, flags


def _sha(*parts):
    import hashlib
    h = hashlib.sha1()
    for p in parts:
        h.update(p if isinstance(p, bytes) else str(p).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def source_sha():
    """Identity of the library the sources beside this file would build: SHA-1 over the compiler flags, every csrc/*.hip in
    SOURCES, vd_common.h and include/vd_amd.h -- by CONTENT, so a checkout / rsync / snapshot with arbitrary mtimes can
    neither hide a stale binary nor force a rebuild of a fresh one.  The built library carries it (vd_source_sha())."""
    hipcc, flags = _toolchain()
    return _sha("\0".join(flags), repr(sorted(SOURCE_FLAGS.items())), *[_read(os.path.join(_CSRC, s)) for s in SOURCES], _read(os.path.join(_CSRC, "vd_common.h")),
                _read(HEADER))[:16]


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    try:
        with open(SO_PATH + ".flags") as f:
            return f.read().strip() != source_sha()
    except OSError:
        return True


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> in-tree libvdamd.so (cross-compiles without a GPU): one object per source
    (csrc/.obj/, compiled in parallel, rebuilt only when the hash of the source + headers + flags changed), then one link.
    One process per GPU may get here at once (torchrun): the build is serialised by a lock file and the library is
    written under a per-process name, so a rank either builds or waits and then finds the library fresh."""
    if not force and not _stale():
        return SO_PATH
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(SO_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return SO_PATH
            hipcc, flags = _toolchain()
            sha = source_sha()
            objdir = os.path.join(_CSRC, ".obj")
            os.makedirs(objdir, exist_ok=True)
            hdr_bytes = [_read(os.path.join(_CSRC, "vd_common.h")), _read(HEADER)]
            # the library's identity as a translation unit of its own (generated: not part of the hash it states)
            idsrc = os.path.join(objdir, "build_id.cpp")
            with open(idsrc, "w") as f:
                f.write(f'extern "C" const char* vd_source_sha(void) {{ return "{sha}"; }}\n')

            def compile_one(src):
                path = idsrc if src is None else os.path.join(_CSRC, src)
                obj = os.path.join(objdir, os.path.basename(path).rsplit(".", 1)[0] + ".o")
                extra = SOURCE_FLAGS.get(src, [])
                want = _sha("\0".join(flags + extra), _read(path), *hdr_bytes)
                try:
                    have = open(obj + ".sha").read().strip()
                except OSError:
                    have = ""
                if force or have != want or not os.path.exists(obj):
                    cmd = [hipcc, *flags, *extra, "-c", path, "-o", obj] if src is not None else [hipcc, "-O2", "-fPIC", "-c", path, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    r = subprocess.run(cmd, capture_output=True, text=True)
                    if r.returncode != 0:
                        raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
                    with open(obj + ".sha", "w") as f:
                        f.write(want + "\n")
                return obj

            with ThreadPoolExecutor(max_workers=min(8, len(SOURCES), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_one, SOURCES + [None]))
            tmp = f"{SO_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, SO_PATH)
            with open(SO_PATH + ".flags", "w") as f:
                f.write(sha + "\n")
            return SO_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


class VdConfig(ctypes.Structure):
    _fields_ = [("image_size", ctypes.c_int), ("num_channels", ctypes.c_int), ("num_res_blocks", ctypes.c_int),
                ("num_heads", ctypes.c_int), ("T", ctypes.c_int), ("n_attention_ds", ctypes.c_int),
                ("attention_ds", ctypes.c_int * 8), ("use_scale_shift_norm", ctypes.c_int),
                ("use_spatial_encoding", ctypes.c_int), ("use_frame_encoding", ctypes.c_int),
                ("enforce_position_invariance", ctypes.c_int), ("use_rpe_net", ctypes.c_int),
                ("allow_interactions_between_padding", ctypes.c_int), ("rp_alpha", ctypes.c_float),
                ("rp_beta", ctypes.c_float), ("rp_gamma", ctypes.c_float), ("time_embed_mult", ctypes.c_int),
                ("cond_emb_type", ctypes.c_int), ("learn_sigma", ctypes.c_int)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_longlong
_U = ctypes.c_ulonglong
_F = ctypes.c_float

# name -> (restype, argtypes); every symbol declared in include/vd_amd.h
SIGNATURES = {
    "vd_last_error": (ctypes.c_char_p, []),
    "vd_version": (ctypes.c_char_p, []),
    "vd_source_sha": (ctypes.c_char_p, []),
    "vd_create": (_I, [ctypes.POINTER(VdConfig), ctypes.POINTER(_P)]),
    "vd_destroy": (None, [_P]),
    "vd_param_count": (_I, [_P]),
    "vd_param_info": (_I, [_P, _I, ctypes.c_char_p, _I, ctypes.POINTER(_I), ctypes.POINTER(_L)]),
    "vd_weights_bytes": (_L, [_P]),
    "vd_set_weight_storage": (_I, [_P, _P, _L]),
    "vd_set_weight_storage_host": (_I, [_P, _P, _L]),
    "vd_load_weight": (_I, [_P, ctypes.c_char_p, _P, _L]),
    "vd_weights_missing": (_I, [_P]),
    "vd_mark_weights_loaded": (_I, [_P]),
    "vd_weights_layout_id": (_U, [_P]),
    "vd_device_errors": (_I, [_P, ctypes.POINTER(_I)]),
    "vd_pos_channels": (_I, [_P]),
    "vd_pos_resolution": (_I, [_P]),
    "vd_set_freqs": (_I, [_P, _P, _I, _P, _I]),
    "vd_set_schedule": (_I, [_P, _I, _P, _P, _F]),
    "vd_workspace_bytes": (_I, [_P, _I, _I, ctypes.POINTER(_L)]),
    "vd_unet_forward": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "vd_p_sample": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _U, _U, _P, _P, _P, _P]),
    "vd_ddim_sample": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _U, _U, _P, _P, _P, _P]),
    "vd_p_mean_variance": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "vd_vb_terms": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "vd_prior_bpd": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "vd_window_generation": (_U, [_P]),
    "vd_set_window_prefix_cache": (_I, [_P, _I]),
    "vd_window_prefix_frames": (_I, [_P]),
    "vd_set_window_suffix_skip": (_I, [_P, _I]),
    "vd_window_suffix_frames": (_I, [_P]),
    "vd_window_begin": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U, _U, _L, _P]),
    "vd_window_run": (_I, [_P, _I, _P]),
    "vd_window_graphs": (_I, [_P]),
    "vd_posterior_update": (_I, [_P, _I, _I, _L, _P, _P, _P, _I, _F, _P, _U, _U, _P, _P, _P]),
    "vd_posterior_from_xstart": (_I, [_P, _I, _I, _L, _P, _P, _P, _I, _F, _P, _U, _U, _P, _P, _P, _P]),
    "vd_attn_blocks": (_I, [_P]),
    "vd_attn_block_info": (_I, [_P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "vd_set_attn_capture": (_I, [_P, _P, _P, _I]),
    "vd_bwd_weights_bytes": (_L, [_P]),
    "vd_set_bwd_weight_storage": (_I, [_P, _P, _L, _I]),
    "vd_load_weight_bwd": (_I, [_P, ctypes.c_char_p, _P, _L]),
    "vd_guided_step": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vd_q_sample": (_I, [_P, _I, _L, _P, _P, _P, _P, _P]),
    "vd_randn": (_I, [_P, _L, _U, _U, _P]),
    "vd_profile_begin": (_I, []),
    "vd_profile_end": (_I, [_P, _I]),
    "vd_profile_classes": (_I, []),
    "vd_profile_class_name": (ctypes.c_char_p, [_I]),
    "vd_op_conv": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P]),
    "vd_conv_stats_split": (_I, [_I]),
    "vd_conv_wino_block_couts": (_I, [_I, _I, _I, _I]),
    "vd_op_conv_stats": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "vd_op_gn_affine": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "vd_pack_conv3_wino": (_I, [_P, _P, _I, _I]),
    "vd_pack_linear_frag": (_I, [_P, _P, _I, _I]),
    "vd_pack_linear_split": (_I, [_P, _P, _I, _I]),
    "vd_pack_conv3_wino_split": (_I, [_P, _P, _I, _I]),
    "vd_set_model_mean_type": (_I, [_P, _I]),
    "vd_math_mode": (_I, []),
    "vd_split_image_u16": (_L, [_L, _L]),
    "vd_pack_conv3_wino_ups": (_I, [_P, _P, _I, _I]),
    "vd_conv_ups_stats_split": (_I, [_I]),
    "vd_op_conv_wino_ups": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "vd_op_conv_wino_split": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "vd_op_conv_wino_act": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "vd_conv_wino_act_ok": (_I, [_I, _I, _I, _I]),
    "vd_pack_conv3_split": (_I, [_P, _P, _I, _I]),
    "vd_op_conv_split": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "vd_op_linear_split": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    "vd_op_linear_split_stats": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P]),
    "vd_linear_stats_split": (_I, [_I, _I, _I]),
    "vd_op_gn_fold": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "vd_op_affine_apply": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "vd_op_affine_act": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _P, _P]),
    "vd_op_gn_temporal": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "vd_op_attn_spatial": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "vd_op_attn_temporal": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vd_op_out_conv": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
}

_lib = None


def lib():
    """The loaded library with argtypes set; builds it on first use if the .so is missing/stale."""
    global _lib
    if _lib is None:
        path = os.environ.get("VD_LIB")            # kernel-experiment builds (tools/); default: the in-tree library
        if not path:
            build()
            path = SO_PATH
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if not os.environ.get("VD_LIB") and L.vd_source_sha().decode() != source_sha():
            raise RuntimeError(f"libvdamd.so was built from other sources ({L.vd_source_sha().decode()}) than the ones beside it "
                               f"({source_sha()}): rebuild (video_diffusion_amd._lib.build(force=True))")
        _lib = L
    return _lib


class VdError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().vd_last_error().decode(errors="replace")
        raise VdError(f"libvdamd error {rc}: {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array, or NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
