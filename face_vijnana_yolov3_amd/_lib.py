"""ctypes binding of include/fv_hotpath.h (the same stub INTEGRATION.md shows)."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FV_LIB_PATH') or os.path.join(_PKG, 'libfv_hotpath.so')  # FV_LIB_PATH: dev A/B of two builds
_lib = None

c_void_p, c_int, c_double, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p


class FvError(RuntimeError):
    pass


def lib():
    """Load libfv_hotpath.so; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime (libamdhip64): import it FIRST so this library binds to
        # the same runtime instance -- device pointers, streams and events are shared with torch.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise FvError('libfv_hotpath.so not built: run `python -m face_vijnana_yolov3_amd.build` '
                          '(there is no CPU fallback for the hot path)')
        L = ctypes.CDLL(LIB_PATH)
        L.fv_abi_version.restype = c_int
        L.fv_create.restype = c_int
        L.fv_create.argtypes = [c_int, c_void_p, ctypes.POINTER(c_void_p)]
        L.fv_destroy.restype = None
        L.fv_destroy.argtypes = [c_void_p]
        L.fv_last_error.restype = c_char_p
        L.fv_last_error.argtypes = [c_void_p]
        L.fv_set_stream.restype = c_int
        L.fv_set_stream.argtypes = [c_void_p, c_void_p]
        L.fv_decode_nms.restype = c_int
        L.fv_decode_nms.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        _declare(L)
        _lib = L
    return _lib


class LayerDesc(ctypes.Structure):
    """fv_layer_desc of include/fv_hotpath.h."""
    _fields_ = [('darknet_index', ctypes.c_int32), ('ksize', ctypes.c_int32), ('stride', ctypes.c_int32),
                ('cin', ctypes.c_int32), ('cout', ctypes.c_int32), ('has_bn', ctypes.c_int32),
                ('role', ctypes.c_int32), ('in_div', ctypes.c_int32), ('out_div', ctypes.c_int32),
                ('w_off', ctypes.c_int64), ('gamma_off', ctypes.c_int64), ('beta_off', ctypes.c_int64),
                ('mean_off', ctypes.c_int64), ('var_off', ctypes.c_int64)]


class ProfileRec(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 64), ('launches', ctypes.c_int64), ('ms_total', ctypes.c_double),
                ('flops_total', ctypes.c_double), ('bytes_total', ctypes.c_double)]


BUCKET_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64)


def _declare(L):
    i32, i64, f32, f64, vp, sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, c_void_p, ctypes.c_size_t
    sig = {
        'fv_set_option': (i32, [vp, c_char_p, ctypes.c_longlong]),
        'fv_get_option': (i32, [vp, c_char_p, ctypes.POINTER(ctypes.c_longlong)]),
        'fv_set_bucket_on_side': (i32, [vp, i32]),
        'fv_side_stream': (vp, [vp]),
        'fv_set_conv_scratch': (i32, [vp, vp, sz]),
        'fv_set_bn_zero_debias_step': (i32, [vp, ctypes.c_longlong]),
        'fv_profile_enable': (i32, [vp, i32]),
        'fv_profile_collect': (i32, [vp, ctypes.POINTER(ProfileRec), i32, ctypes.POINTER(i32)]),
        'fv_bbox_iou_pairs': (i32, [vp, vp, vp, i64, vp]),
        'fv_num_layers': (i32, []),
        'fv_layer': (i32, [i32, ctypes.POINTER(LayerDesc)]),
        'fv_param_count': (i64, []),
        'fv_state_count': (i64, []),
        'fv_workspace_bytes': (sz, [i32, i32, i32]),
        'fv_forward_infer': (i32, [vp, vp, vp, vp, i32, i32, vp, sz, vp]),
        'fv_forward_base': (i32, [vp, vp, vp, vp, i32, i32, vp, sz, vp, vp]),
        'fv_train_step': (i32, [vp, vp, vp, vp, vp, i32, i32, vp, sz, vp, vp, f64, BUCKET_FN, vp]),
        'fv_train_workspace_tensor': (i32, [i32, i32, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(i64)]),
        'fv_adam_step': (i32, [vp, vp, vp, vp, vp, i64, i64, f64, f64, f64, f64, f64]),
        'fv_scale': (i32, [vp, vp, i64, f64]),
        'fv_conv2d_forward': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp]),
        'fv_conv2d_stat_rows': (i32, [i64]),
        'fv_conv2d_dgrad': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
        'fv_conv2d_wgrad': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
        'fv_transpose_weights': (i32, [vp, vp, i32, i32, i32, i32, vp]),
        'fv_pack_first_layer': (i32, [vp, vp, i32, i32, vp]),
        'fv_bn_finalize': (i32, [vp, vp, vp, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp]),
        'fv_bn_act': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32]),
        'fv_bn_bwd_scratch_floats': (i64, [i64, i32]),
        'fv_bn_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, vp, vp, vp]),
        'fv_bn_stat_slots': (i32, [i32]),
        'fv_conv2d_forward_slots': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32]),
        'fv_bn_act_slots': (i32, [vp, vp, vp, i32, i64, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, f32]),
        'fv_conv2d_dgrad_bnred': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, i32]),
        'fv_bn_bwd_slots': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, i32, i32, vp, vp, vp]),
        'fv_mse_loss_grad': (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
        'fv_fd_loss_grad': (i32, [vp, vp, vp, i32, i32, vp, vp]),
        'fv_letterbox': (i32, [vp, vp, i32, i32, i32, vp, ctypes.POINTER(ctypes.c_int32)]),
        'fv_letterbox_batch': (i32, [vp, vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), i32, i32, vp,
                                    ctypes.POINTER(ctypes.c_int32)]),
        'fv_yolov3_num_layers': (i32, []),
        'fv_yolov3_layer': (i32, [i32, i32, ctypes.POINTER(LayerDesc)]),
        'fv_yolov3_param_count': (i64, [i32]),
        'fv_yolov3_state_count': (i64, [i32]),
        'fv_yolov3_workspace_bytes': (sz, [i32, i32, i32]),
        'fv_yolov3_forward': (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]),
        'fv_yolov3_train_workspace_bytes': (sz, [i32, i32, i32]),
        'fv_yolov3_train_step': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp, vp, f64, BUCKET_FN, vp]),
        'fv_yolov3_train_workspace_tensor': (i32, [i32, i32, i32, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(i64)]),
        'fv_yolo_decode_nms_batch': (i32, [vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(f32), f32, f64, i32, i32, i32, i32, i32,
                                           vp, vp, vp, vp]),
        'fv_yolo_decode_nms': (i32, [vp, vp, vp, vp, i32, i32, ctypes.POINTER(f32), f32, f64, i32, i32, i32, i32, i32,
                                     vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


class Context:
    """One fv_ctx per GPU / rank, bound to a HIP stream (default: torch's current stream)."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise FvError('no MI355X visible: the hot path has no CPU fallback')
        self.device = int(device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.overlap = True            # option 'overlap' (kept (the library's default)
        self._h = c_void_p()
        rc = lib().fv_create(self.device, c_void_p(stream), ctypes.byref(self._h))
        if rc != 0:
            raise FvError('fv_create failed (%d): %s' % (rc, lib().fv_last_error(None).decode()))

    def check(self, rc, what):
        if rc != 0:
            raise FvError('%s failed (%d): %s' % (what, rc, lib().fv_last_error(self._h).decode()))

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        self.check(lib().fv_set_stream(self._h, c_void_p(stream_ptr)), 'fv_set_stream')

    def set_option(self, key, value):
        """fv_set_option: a tuning switch of include/fv_hotpath.h ('overlap', 'tail_split', 'conv_waves8', 'conv1x1_persist',
        'conv_bm64', 'conv_small', 'conv_halo', 'conv0_direct', 'wgrad_fused_taps')."""
        self.check(lib().fv_set_option(self._h, key.encode(), int(value)), 'fv_set_option(%s)' % key)

    def get_option(self, key):
        v = ctypes.c_longlong(0)
        self.check(lib().fv_get_option(self._h, key.encode(), ctypes.byref(v)), 'fv_get_option(%s)' % key)
        return int(v.value)

    def set_overlap(self, on):
        self.set_option('overlap', 1 if on else 0)
        self.overlap = bool(on)

    def set_bucket_on_side(self, on):
        self.check(lib().fv_set_bucket_on_side(self._h, 1 if on else 0), 'fv_set_bucket_on_side')

    def side_stream(self):
        """hipStream_t of the library's side stream (int), for torch.cuda.ExternalStream."""
        return int(lib().fv_side_stream(self._h) or 0)

    def scale(self, tensor, alpha):
        """tensor *= alpha on the context's stream (fv_scale: the BN-state mean over the ranks of a data-parallel step)."""
        self.check(lib().fv_scale(self._h, ptr(tensor), tensor.numel(), float(alpha)), 'fv_scale')

    def set_tail_split(self, on):
        self.set_option('tail_split', 1 if on else 0)

    def set_bn_zero_debias_step(self, step):
        self.check(lib().fv_set_bn_zero_debias_step(self._h, int(step)), 'fv_set_bn_zero_debias_step')

    def set_wgrad_fused_taps(self, on):
        self.set_option('wgrad_fused_taps', 1 if on else 0)

    def set_conv_halo(self, on):
        self.set_option('conv_halo', 1 if on else 0)

    def set_conv_waves8(self, on):
        self.set_option('conv_waves8', 1 if on else 0)

    def set_conv0_direct(self, on):
        self.set_option('conv0_direct', 1 if on else 0)

    def set_conv_scratch(self, tensor):
        """Lend device scratch (a torch tensor, kept alive here) to the per-operator conv calls."""
        self._conv_scratch = tensor
        if tensor is None:
            self.check(lib().fv_set_conv_scratch(self._h, c_void_p(None), 0), 'fv_set_conv_scratch')
        else:
            self.check(lib().fv_set_conv_scratch(self._h, c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()),
                       'fv_set_conv_scratch')

    def profile(self, on, shapes=False):
        """on: HIP-event pairs around every launch; shapes=True: the matrix kernels' records carry their problem shape in the name."""
        self.check(lib().fv_profile_enable(self._h, (2 if shapes else 1) if on else 0), 'fv_profile_enable')

    def profile_collect(self):
        """-> {kernel name: dict(launches, ms, flops, bytes)} since profiling was enabled (syncs)."""
        recs = (ProfileRec * 512)()
        n = ctypes.c_int(0)
        self.check(lib().fv_profile_collect(self._h, recs, 512, ctypes.byref(n)), 'fv_profile_collect')
        return {recs[i].name.decode(): dict(launches=int(recs[i].launches), ms=recs[i].ms_total,
                                            flops=recs[i].flops_total, bytes=recs[i].bytes_total)
                for i in range(min(n.value, 512))}

    def close(self):
        if self._h:
            lib().fv_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Raw device pointer of a contiguous torch tensor."""
    assert t.is_contiguous()
    return c_void_p(t.data_ptr())
