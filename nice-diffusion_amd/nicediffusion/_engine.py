"""Execution plan of one UNet forward on an AMD GPU: a flat list of libnd_hip.so launches over NHWC buffers.

The plan is built once per (model, batch size): weights are repacked into the kernels' layout, every intermediate
gets a slot in a reuse pool, and every launch becomes a pre-bound ctypes call.  Running the plan is a loop over those
calls on the current stream; it allocates nothing and never synchronises, so a caller can capture it into a hipGraph
(``Diffusion.denoise`` does).  Reference counterpart: the module-by-module Python dispatch of
``DiffusionModel.forward`` (model.py:451-476) and ``UsesStepsSequential.forward`` (model.py:42-48).
"""
import ctypes
import json
import math
import os
import sys
import time

import torch

from . import _hip

GN_GROUPS = 32
GN_EPS = 1e-5

# (device index, NI, H, W, Cin, N, ksize, flags, has_rowbias, has_residual) -> fastest tile variant, measured once
_TUNED = {}


def _autotune_enabled():
    return os.environ.get('ND_AUTOTUNE', '1') != '0'


def _tune_cache_path():
    return os.environ.get('ND_TUNE_CACHE')


def _tune_stamp(product=False):
    """What a file of measured choices (or any other measured artefact) is only valid for: the library's SOURCE HASH
    (nd_build_id: one changed character of any kernel source changes it -- nobody has to remember to bump a version),
    its variant flags if it is an ablation build, its version and its variant tables (a choice is a variant NUMBER)."""
    lib = _hip.load()
    names = [lib.nd_conv_winograd_variant_name(v).decode() for v in range(lib.nd_conv_winograd_num_variants())]
    names += [lib.nd_conv_winograd_f4_variant_name(v).decode() for v in range(lib.nd_conv_winograd_f4_num_variants())]
    names += [lib.nd_conv_bf16_variant_name(v).decode() for v in range(lib.nd_conv_bf16_num_variants())]
    # the direct / GEMM variants have no names: their block shapes identify them (a renumbered or retuned variant changes the stamp)
    bm, bn, nt = (ctypes.c_int() for _ in range(3))
    direct = []
    for v in range(lib.nd_conv_num_variants()):
        lib.nd_conv_variant_info(v, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nt))
        direct.append('{}x{}x{}'.format(bm.value, bn.value, nt.value))
    # product=True: the stamp the PRODUCT build of the same sources carries (the variant flags of an ablation build dropped)
    bid = _hip.build_id().split('+')[0] if product else _hip.build_id()
    return 'b{}:v{}:d{}:{}:{}'.format(bid, lib.nd_version(), lib.nd_conv_num_variants(), ','.join(direct), '|'.join(names))


def preload_tune_cache(path, device_index=None, override=False):
    """Take the measured choices of a file written by ``_save_tune_cache`` (committed ones live under profiles/): entries
    are re-keyed to ``device_index`` (default: the current device) and skipped when this process already holds a choice
    for the shape unless ``override``.  A file stamped by another library build is ignored.  Returns the number of
    choices taken (0 if the file is missing, unreadable or stale)."""
    try:
        raw = json.load(open(path))
    except (ValueError, OSError):
        return 0
    stamp = raw.pop('__stamp__', None)
    if stamp != _tune_stamp():
        # an ablation / experiment build (loaded under ND_ALLOW_ABLATION=1) is timed on the PRODUCT's plan: it takes the cache
        # of the product build of the same sources; files it writes itself carry its own flagged stamp
        # (ND_TUNE_STAMP_ANY=1, development only: A/B runs between the edit of a kernel and the regeneration of the caches; the
        # switch shows in the bench line's config.switches)
        if os.environ.get('ND_TUNE_STAMP_ANY') == '1':
            print('[nd] ND_TUNE_STAMP_ANY=1: taking {} although it was written by another library build'.format(path), file=sys.stderr)
        elif not ('+' in _hip.build_id() and stamp == _tune_stamp(product=True)):
            print('[nd] tune cache {} was written by another library build: ignored'.format(path), file=sys.stderr)
            return 0
        else:
            print('[nd] variant build {}: taking the product build\'s tune cache {}'.format(_hip.build_id(), path), file=sys.stderr)
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    n = 0
    for k, v in raw.items():
        key = (device_index,) + tuple(json.loads(k))[1:]
        if override or key not in _TUNED:
            _TUNED[key] = tuple(v)
            n += 1
    return n


def _load_tune_cache():
    """Optional on-disk cache of the measured choices (ND_TUNE_CACHE=file.json): lets a second process (e.g. a run under
    rocprofv3) start without the tuning launches."""
    path = _tune_cache_path()
    if path and os.path.exists(path) and not _TUNED:
        preload_tune_cache(path)


def _save_tune_cache(path=None):
    path = path or _tune_cache_path()
    if path:
        try:        # N ranks share the choices of rank 0 (parallel.tune_on_rank0): only that rank writes the file
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
                return
        except ImportError:
            pass
        out = {json.dumps(list(k)): list(v) for k, v in _TUNED.items()}
        out['__stamp__'] = _tune_stamp()
        json.dump(out, open(path, 'w'))


def _pad4(n):
    return (n + 3) // 4 * 4


class Act:
    """NHWC activation: flat buffer (allocated as fp32 words; holds fp32 or bf16 elements) + geometry."""
    __slots__ = ('t', 'NI', 'H', 'W', 'C', 'ld', 'cs', 'bf16')

    def __init__(self, t, NI, H, W, C, ld=None, bf16=False):
        self.t, self.NI, self.H, self.W, self.C = t, NI, H, W, C
        self.ld = C if ld is None else ld
        self.cs = None      # (('chpart', float offset), rows per image): partial output statistics left by the producing conv
        self.bf16 = bf16

    @property
    def ptr(self):
        return self.t.data_ptr()


class Normed:
    """GroupNorm output that has not been materialised: the consumer conv applies ``x * A + B`` (+SiLU) while loading.
    The conv emits nd_groupnorm_coeffs (fp32 [NI][C] A/B) first, or falls back to the explicit apply kernel."""
    __slots__ = ('src', 'src2', 'C', 'silu', 'norm', 'scale_ptr', 'shift_ptr', 'ld_ss', 'slot', 'nblk', 'pending')

    def __init__(self, **kw):
        self.pending = None       # (args, label) of a nd_groupnorm_stats_from_partials launch not emitted yet
        for k, v in kw.items():
            setattr(self, k, v)


def _epilogue_stats_mode():
    """ND_GN_EPILOGUE_STATS: which fp32 Winograd convs leave the per-channel partial statistics of their output behind for
    the GroupNorms that read it (nd_conv3x3_winograd_vstats_nhwc + nd_groupnorm_stats_from_partials) instead of a statistics
    pass over the tensor.  'auto' (default): conv_wino4_kernel only, whose epilogue pays ~90 vector instructions per
    n tile for it; '1': conv_wino16_kernel too (measured at B=64: removes 1.0 ms of statistics kernels, costs the
    MFMA-bound convs 2.2 ms -- on these fp32 kernels every vector instruction takes its cycles from the matrix pipe);
    '0': never."""
    return os.environ.get('ND_GN_EPILOGUE_STATS', 'auto')


def _gn_partials_enabled(bf16=False):
    """ND_GN_PARTIALS=0 restores one statistics pass over the (concatenated) input of every GroupNorm.  Default in fp32
    plans: every tensor's per-channel sums are computed ONCE -- by the epilogue of the conv that produces it, or by one
    nd_groupnorm_channel_partials_nhwc pass over that tensor alone -- kept with the activation and re-grouped by every
    norm that reads it, so the up path's concatenation norms (model.py:474,190) no longer re-read the skip tensors
    (statistics 2.70 -> 1.30 ms per forward at configs[1]).  bf16 plans keep the pass per norm by default: most of their
    convs already leave epilogue statistics and the extra folds measured equal (26.2 vs 26.2 ms per step at configs[3],
    same box, 44 more launches); ND_GN_PARTIALS=1 switches the route on there too."""
    return os.environ.get('ND_GN_PARTIALS', '0' if bf16 else '1') != '0'


def _gn_merge_coeffs():
    """ND_GN_MERGE_COEFFS (default 1): a norm whose statistics come from partial rows and whose consumer applies it in its
    loader gets ONE launch (nd_groupnorm_coeffs_from_partials) instead of the fold + the coefficient kernel."""
    return os.environ.get('ND_GN_MERGE_COEFFS', '1') != '0'


def _bf16_epilogue_stats():
    """ND_BF16_EPILOGUE_STATS (default 1): bf16 3x3 convs whose output feeds a GroupNorm leave that norm's partial statistics
    behind (nd_conv3x3_bf16_stats_nhwc + nd_groupnorm_stats_from_partials) where the plan builder measures that to be
    cheaper than the statistics pass over the tensor; 0: always the statistics pass; 2: the epilogue form wherever a tile
    variant offers it (tests)."""
    return int(os.environ.get('ND_BF16_EPILOGUE_STATS', '1'))


def _gn_fused_max_elems(bf16=False):
    """ND_GN_FUSED_MAX (default 2^23 elements in fp32 plans, 2^22 in bf16 plans): a GroupNorm over at most that many elements
    (images x pixels x channels) takes ONE launch (nd_groupnorm_fused_nhwc: statistics + apply, one block per (group,
    image), 16-byte loads) instead of the per-channel partials pass, the fold and the apply pass.  Measured (interleaved,
    same box): the EMNIST preset at batch 4 is launch-bound (257 -> 149 launches per forward); on
    configs[1] the 8x8 level's norms (3.1 / 6.3 M elements, three launches of 6-11 us each) gain 0.2 ms per forward at 2^23
    and the 16x16 level (9.4 M) loses 0.6 ms at 2^24; configs[3] / [4] gain 1.2 % / 0.55 % at 2^22.  0 switches the form off."""
    return int(os.environ.get('ND_GN_FUSED_MAX', str(1 << 22 if bf16 else 1 << 23)))


def _bf16_splitk():
    """ND_BF16_SPLITK (default 1): bf16 convs on small maps (<= 8192 output pixels, >= 512 input channels) are also measured
    split over K (nd_conv_bf16_splitk_nhwc, 2 and 4 splits) and run that way where it is faster; 0: never; 2: wherever a
    split form exists (tests)."""
    return int(os.environ.get('ND_BF16_SPLITK', '1'))


def _f32_splitk():
    """ND_F32_SPLITK (default 1): fp32 convolutions with few output pixels (<= 8192) and a long contraction are also measured
    split over K (nd_conv_splitk_nhwc: 2 / 4 / 8 block rows over the input-channel chunks + a deterministic reduce) and run
    that way where it is faster -- the 7x7 / 14x14 layers of the EMNIST preset at batch 4 run 16-64 blocks of a 1152-MFMA
    serial chain otherwise; 0: never; 2: wherever the form exists (tests)."""
    return int(os.environ.get('ND_F32_SPLITK', '1'))


def _winograd_f4():
    """ND_WINOGRAD_F4 (default 1): fp32 3x3 convolutions on maps that are multiples of 4 are also measured on the Winograd
    F(4x4,3x3) kernel (nd_conv3x3_winograd_f4_nhwc: a quarter of the direct multiplies, 0.5625 x the matrix instructions of
    F(2x2,3x3)) and run there where it is faster.  It is the numerically looser form -- about 5x the rounding error of
    F(2x2,3x3): 7e-6 per forward of the 64x64 preset, 1.4e-4 after its own 25-step chain against the reference's 1e-3
    (profiles/r05_f4_numerics_preset64.txt; the GPU-measured numbers are asserted in tests/test_gpu_model.py) -- so
    0 keeps F(2x2,3x3) everywhere; 2: wherever the form exists (tests)."""
    return int(os.environ.get('ND_WINOGRAD_F4', '1'))


def _gn_apply_coeffs():
    """ND_GN_APPLY_COEFFS=0: the apply pass folds its statistics itself (fold launch + nd_groupnorm_apply_nhwc; A/B switch)."""
    return os.environ.get('ND_GN_APPLY_COEFFS', '1') != '0'


def _edge_convs():
    """ND_EDGE_CONVS=0: the first / last convolutions go through the general 3x3 kernels (A/B switch)."""
    return os.environ.get('ND_EDGE_CONVS', '1') != '0'


def _fuse_gn_mode():
    """0: never fold GroupNorm into the consumer conv; 1 (default): fold the affine-only norms (attention); 2: also fold
    norm+SiLU.  Measured: a conv with N/BN output-channel blocks re-evaluates SiLU for every block and halo overlap
    (~8x for 384 channels on the Winograd tiles), which costs 20-25 % of the conv and far exceeds the 3.4 ms of the
    separate HBM-bound apply pass, so SiLU norms keep their own kernel."""
    return int(os.environ.get('ND_FUSE_GN', '1'))


class _Pool:
    """Plan-time buffer reuse: launches are stream-ordered, so a buffer can be handed out again as soon as the plan
    has emitted its last reader."""

    def __init__(self, device):
        self.device = device
        self.free = []      # list of (numel, tensor)
        self.all = []       # every buffer ever handed out: the plan keeps them alive (launches hold raw pointers)
        self.total = 0

    def take(self, numel):
        best = None
        for i, (n, t) in enumerate(self.free):
            if n >= numel and (best is None or n < self.free[best][0]):
                best = i
        if best is not None and self.free[best][0] <= 2 * numel:
            return self.free.pop(best)[1]
        t = torch.empty(numel, dtype=torch.float32, device=self.device)
        self.all.append(t)
        self.total += numel
        return t

    def give(self, t):
        self.free.append((t.numel(), t))


_CLOCK_SETTLED = [False]


class UNetPlan:
    def __init__(self, model, NI, dtype='fp32'):
        """``dtype``: 'fp32' (exact-fp32 MFMA / Winograd kernels, the reference's arithmetic) or 'bf16' (BASELINE
        configs[3], [4]: bf16 activations and weights in HBM, fp32 accumulation, fp32 GroupNorm statistics, fp32
        embedding MLP, fp32 x_t / model output at the sampler's edge)."""
        if dtype not in ('fp32', 'bf16'):
            raise ValueError('dtype must be fp32 or bf16')
        self.lib = _hip.load()
        self.model = model
        self.NI = NI
        self.dtype = dtype
        self.bf16 = dtype == 'bf16'
        self.dt = _hip.DT_BF16 if self.bf16 else _hip.DT_F32
        dev = next(model.parameters()).device
        _hip.require_device(next(model.parameters()), 'model parameters')
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        _hip.require_gfx950(dev.index)
        if torch.cuda.current_device() != dev.index:
            raise _hip.NdHipError('build the plan with its device current (torch.cuda.device({}))'.format(dev.index))
        self.device = dev
        self.ops = []           # (fn, args, label)
        self.meta = []          # per launch: label, entry point, algorithmic flops, conv tile variant
        self.keep = []          # tensors that must outlive the plan (packed weights, etc.)
        self.pool = _Pool(dev)
        self.packed_floats = 0
        self.flops = 0          # algorithmic flops of the MFMA launches (2 per MAC)
        self.conv_flops = {}    # label -> flops
        R = model.resolution
        self.R = R
        self.Cin = model.in_channels
        self.Cin_p = _pad4(model.in_channels)
        self.Cout = model.out_channels
        self.Cout_p = _pad4(model.out_channels)
        f32 = dict(dtype=torch.float32, device=dev)
        # static I/O buffers
        self.x_in = torch.zeros(NI * R * R * self.Cin_p, **f32)
        self.t_in = torch.zeros(NI, dtype=torch.int64, device=dev)
        self.y_in = torch.zeros(NI, dtype=torch.int64, device=dev) if model.conditional else None
        self.out = torch.empty(NI * R * R * self.Cout_p, **f32)
        self._gn_slots = 0
        self._gn_doubles = 0     # float64 words of GroupNorm partial statistics ([NI][blocks][32][2] per norm)
        self._cs_floats = 0     # fp32 words of partial output statistics (see conv(want_stats=True))
        self._splitk_floats = 0  # fp32 words of the split-K workspace shared by the convs that use it
        self.taps = []          # (module name, number of ops emitted when its output is complete, Act): debug hook
        _load_tune_cache()
        n_tuned = len(_TUNED)
        self._build()
        if len(_TUNED) != n_tuned:
            _save_tune_cache()
        self.weight_signature = model._weight_signature()

    # ------------------------------------------------------------------------------------------------ emit helpers
    def _emit(self, fn, args, label, flops=0, variant=None, ksize=None, shape=None):
        self.ops.append((fn, tuple(args), label))
        self.meta.append(dict(label=label, fn=fn.__name__, flops=flops, variant=variant, ksize=ksize, shape=shape))

    def _new(self, NI, H, W, C):
        n = NI * H * W * C
        return Act(self.pool.take((n + 1) // 2 if self.bf16 else n), NI, H, W, C, bf16=self.bf16)

    def _release(self, act):
        if isinstance(act, Act) and act.t is not None:
            self.pool.give(act.t)

    def _packed(self, weight, pad_c_to=None):
        """Conv2d [N,C,k,k] / Conv1d [N,C,1] / Linear [N,C] weight -> MFMA-fragment order, once, on the device.
        ``pad_c_to``: treat the weight as having that many input channels (zero columns appended)."""
        w = weight.detach()
        N, C = w.shape[0], w.shape[1]
        k = w.shape[2] if w.dim() == 4 else 1
        w = w.contiguous()
        if pad_c_to is not None and pad_c_to != C:
            wp = torch.zeros((N, pad_c_to) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
            wp[:, :C] = w
            w, C = wp, pad_c_to
        n = self.lib.nd_conv_weight_floats(N, C, k)
        assert n > 0
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _hip.check(self.lib.nd_repack_conv_weight(w.data_ptr(), out.data_ptr(), N, C, k, self._stream()),
                   'nd_repack_conv_weight')
        self.keep.append(out)
        self.packed_floats += n
        return out

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _packed_wino(self, weight, pad_c_to=None):
        """3x3 weight -> Winograd domain (U = G g G^T) in fragment order, once, on the device."""
        w = weight.detach().contiguous()
        N, C = w.shape[0], w.shape[1]
        if pad_c_to is not None and pad_c_to != C:
            wp = torch.zeros((N, pad_c_to, 3, 3), dtype=w.dtype, device=w.device)
            wp[:, :C] = w
            w, C = wp, pad_c_to
        n = self.lib.nd_conv_winograd_weight_floats(N, C)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _hip.check(self.lib.nd_repack_conv_weight_winograd(w.data_ptr(), out.data_ptr(), N, C, self._stream()),
                   'nd_repack_conv_weight_winograd')
        return out

    def _packed_wf4(self, weight, pad_c_to=None):
        """3x3 weight -> Winograd F(4x4,3x3) domain (U = G g G^T, float64 rounded once) in conv_wf4_kernel's fragment order."""
        w = weight.detach().contiguous()
        N, C = w.shape[0], w.shape[1]
        if pad_c_to is not None and pad_c_to != C:
            wp = torch.zeros((N, pad_c_to, 3, 3), dtype=w.dtype, device=w.device)
            wp[:, :C] = w
            w, C = wp, pad_c_to
        n = self.lib.nd_conv_winograd_f4_weight_floats(0, N, C)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _hip.check(self.lib.nd_repack_conv_weight_winograd_f4(w.data_ptr(), out.data_ptr(), N, C, 0, self._stream()),
                   'nd_repack_conv_weight_winograd_f4')
        return out

    def conv(self, src, weight, bias, N, ksize, out=None, src2=None, rowbias=None, ld_rowbias=0,
             residual=None, flags=0, label='conv', pad_c_to=None, want_stats=False, fuse_gn=False):
        """Emit one convolution.  ``src`` (and optional ``src2``, concatenated after it) are Acts; ``weight`` is the
        module's parameter (packed here); output spatial size is src's, doubled when CONV_IN_UP2X is set.  For 3x3
        convolutions on even sizes the Winograd F(2x2,3x3) kernel competes with the direct kernel's tile shapes and
        the fastest measured implementation is kept."""
        if self.bf16:
            return self._conv_bf16(src, weight, bias, N, ksize, out, src2, rowbias, ld_rowbias, residual, flags, label,
                                   pad_c_to, want_stats=want_stats)
        plain = src2 is None and rowbias is None and residual is None and flags == 0 and _edge_convs()
        if plain and ksize == 3 and isinstance(src, Act) and src.ld == 4 and src.C <= 4 and src.W % 16 == 0 and \
                N % 16 == 0 and N <= 256 and (out is None or out.ld % 4 == 0):
            return self._conv_first(src, weight, bias, N, out, label, want_stats)
        if plain and ksize == 3 and isinstance(src, Normed) and src.src2 is None and 9 * N <= 64 and \
                (src.src.H * src.src.W) % 256 == 0 and src.C % 32 == 0 and weight.shape[1] == src.C:
            return self._conv_last(src, weight, bias, N, out, label)
        gn = [None, None, 0]
        tmp = None
        if isinstance(src, Normed):
            nm = src
            up_ = 1 if (flags & _hip.CONV_IN_UP2X) else 0
            hw = (nm.src.H << up_) * (nm.src.W << up_)
            # the fused form needs one image per block: every tile shape is <= 256 pixels
            mode = _fuse_gn_mode()
            if (fuse_gn or mode >= 2 or (mode == 1 and not nm.silu)) and hw >= 256 and hw % 256 == 0 and rowbias is None:
                src, src2 = nm.src, nm.src2
                coefA = torch.empty(nm.src.NI * nm.C, dtype=torch.float32, device=self.device)
                coefB = torch.empty(nm.src.NI * nm.C, dtype=torch.float32, device=self.device)
                self.keep += [coefA, coefB]
                self._emit_coeffs(nm, coefA, coefB)
                gn = [coefA.data_ptr(), coefB.data_ptr(), nm.C]
                if nm.silu:
                    flags |= _hip.CONV_GN_SILU
            else:
                tmp = self._materialise(nm)
                src, src2 = tmp, None
        up = 1 if (flags & _hip.CONV_IN_UP2X) else 0
        NI, H, W = src.NI, src.H << up, src.W << up
        if out is None:
            out = self._new(NI, H, W, N)
        C1 = 0 if src2 is None else src2.C
        head = [src.ptr, src.C, src.ld, None if src2 is None else src2.ptr, C1, 0 if src2 is None else src2.ld]
        tail = [bias, rowbias, ld_rowbias, None if residual is None else residual.ptr,
                0 if residual is None else residual.ld, out.ptr, out.ld, NI, H, W, N]
        fl = 2 * NI * H * W * N * ksize * ksize * (src.C + C1)
        key = (NI, H, W, src.C + C1, N, ksize, flags, rowbias is not None, residual is not None, gn[0] is not None)
        # does a GroupNorm that reads `out` use per-channel partial statistics at all (small tensors take the one-launch norm)?
        stats_wanted = bool(want_stats and gn[0] is None and out.ld == N and _epilogue_stats_mode() != '0' and
                            NI * H * W * N > _gn_fused_max_elems(False))
        kind, var = self._pick_impl(key, fl, weight, pad_c_to, head, tail, flags, gn, single=src2 is None, out=out,
                                    stats_wanted=stats_wanted)
        if kind in ('wf4', 'wf4+splitk'):
            splits = var[1] if kind == 'wf4+splitk' else 1
            wq = self._packed_wf4(weight, pad_c_to)
            self.keep.append(wq)
            self.packed_floats += wq.numel()
            ph, ws = None, None
            if splits > 1:
                need = self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, src.C, 3, splits)
                if need <= 0:
                    raise _hip.NdHipError('nd_conv_splitk_workspace_floats: ' + _hip.last_error())
                self._splitk_floats = max(self._splitk_floats, need)
                ws = ('splitk', 0)
            if stats_wanted:
                # per-channel partial statistics of the output: from the kernel's epilogue, or -- split over K -- from the reduce pass
                rows = (self.lib.nd_conv_winograd_f4_splitk_stats_rows if splits > 1 else self.lib.nd_conv_winograd_f4_stats_rows)(0, NI, H, W)
                if rows > 0:
                    ph = ('chpart', self._cs_floats)
                    out.cs = (ph, rows)
                    self._cs_floats += (NI * rows * 2 * N + 3) // 4 * 4
            self._emit(self.lib.nd_conv3x3_winograd_f4_nhwc, head[:3] + [wq.data_ptr()] + tail + [flags, 0, ph, splits, ws], label,
                       flops=fl, variant=('wf4', 0), ksize=ksize, shape=(NI, H, W, src.C, N))
        elif kind == 'wino+splitk':
            var, splits = var
            need = self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, src.C + C1, 3, splits)
            if need <= 0:
                raise _hip.NdHipError('nd_conv_splitk_workspace_floats: ' + _hip.last_error())
            self._splitk_floats = max(self._splitk_floats, need)
            wq = self._packed_wino(weight, pad_c_to)
            self.keep.append(wq)
            self.packed_floats += wq.numel()
            self._emit(self.lib.nd_conv3x3_winograd_splitk_nhwc, head + [wq.data_ptr()] + tail + [flags, var, splits, ('splitk', 0)],
                       label, flops=fl, variant=('wino', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        elif kind == 'direct+splitk':
            var, splits = var
            need = self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, src.C + C1, ksize, splits)
            if need <= 0:
                raise _hip.NdHipError('nd_conv_splitk_workspace_floats: ' + _hip.last_error())
            self._splitk_floats = max(self._splitk_floats, need)
            wp = self._packed(weight, pad_c_to)
            self._emit(self.lib.nd_conv_splitk_nhwc, head + [wp.data_ptr()] + tail + [ksize, flags, var, splits, ('splitk', 0)],
                       label, flops=fl, variant=('direct', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        elif kind == 'wino':
            wq = self._packed_wino(weight, pad_c_to)
            self.keep.append(wq)
            self.packed_floats += wq.numel()
            mode = _epilogue_stats_mode()
            rows = 0
            if want_stats and mode != '0' and gn[0] is None and out.ld == N and (
                    mode == '1' or self.lib.nd_conv_winograd_variant_name(var) == b'nd::conv_wino4_kernel'):
                rows = self.lib.nd_conv_winograd_stats_rows(var, NI, H, W)
            if rows > 0:
                # the conv leaves the per-channel partial statistics of its output behind (no pass over `out` for any norm)
                ph = ('chpart', self._cs_floats)
                out.cs = (ph, rows)
                self._cs_floats += (NI * rows * 2 * N + 3) // 4 * 4
                self._emit(self.lib.nd_conv3x3_winograd_vstats_nhwc, head + [wq.data_ptr()] + tail + [flags, var, ph], label,
                           flops=fl, variant=('wino', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
            else:
                self._emit(self.lib.nd_conv3x3_winograd_nhwc, head + [wq.data_ptr()] + tail + [flags, var] + gn, label,
                           flops=fl, variant=('wino', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        else:
            wp = self._packed(weight, pad_c_to)
            rows = 0
            if ksize == 1 and var in (14, 15) and want_stats and gn[0] is None and rowbias is None and out.ld == N and flags == 0 and \
                    _epilogue_stats_mode() != '0':
                # gemm4_kernel's epilogue leaves the per-channel partial statistics of its output behind (the attention
                # block's output projection + residual feeds the next block's in_norm): no pass over `out`
                rows = self.lib.nd_conv1x1_stats_rows(NI, H, W, N)
            if rows > 0:
                ph = ('chpart', self._cs_floats)
                out.cs = (ph, rows)
                self._cs_floats += (NI * rows * 2 * N + 3) // 4 * 4
                self._emit(self.lib.nd_conv1x1_stats_nhwc, head + [wp.data_ptr(), tail[0]] + tail[3:] + [flags, ph], label, flops=fl,
                           variant=('direct', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
            else:
                self._emit(self.lib.nd_conv_nhwc, head + [wp.data_ptr()] + tail + [ksize, flags, var] + gn, label, flops=fl,
                           variant=('direct', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        self.flops += fl
        self.conv_flops[label] = self.conv_flops.get(label, 0) + fl
        if tmp is not None:
            self._release(tmp)
        return out

    # ------------------------------------------------------------------------------------------------ the two end convs
    def _conv_first(self, src, weight, bias, N, out, label, want_stats):
        """Conv2d(in_channels <= 4, N, 3) on the NHWC4 input (model.py:427-431): nd_conv3x3_first_nhwc, which also leaves the
        per-channel partial statistics of its output for the first residual block's in_norm."""
        lib = self.lib
        NI, H, W = src.NI, src.H, src.W
        if out is None:
            out = self._new(NI, H, W, N)
        w = weight.detach().contiguous()
        C0 = w.shape[1]
        wq = torch.empty(lib.nd_conv_first_weight_floats(N), dtype=torch.float32, device=self.device)
        _hip.check(lib.nd_repack_conv_first_weight(w.data_ptr(), wq.data_ptr(), N, C0, self._stream()),
                   'nd_repack_conv_first_weight')
        self.keep.append(wq)
        self.packed_floats += wq.numel()
        ph = None
        if want_stats and out.ld == N and _epilogue_stats_mode() != '0' and NI * H * W * N > _gn_fused_max_elems(False):
            rows = lib.nd_conv3x3_first_stats_rows(NI, H, W, N)
            if rows > 0:
                ph = ('chpart', self._cs_floats)
                out.cs = (ph, rows)
                self._cs_floats += (NI * rows * 2 * N + 3) // 4 * 4
        fl = 2 * NI * H * W * N * 9 * src.C
        self._emit(lib.nd_conv3x3_first_nhwc, [src.ptr, src.ld, wq.data_ptr(), bias, out.ptr, out.ld, NI, H, W, N, ph], label,
                   flops=fl, variant=('first', 0), ksize=3, shape=(NI, H, W, src.C, N))
        self.flops += fl
        self.conv_flops[label] = self.conv_flops.get(label, 0) + fl
        return out

    def _conv_last(self, nm, weight, bias, N, out, label):
        """GroupNorm -> SiLU -> Conv2d(C, N <= 7, 3) (model.py:446-449) with the taps in the GEMM's N dimension: one 1x1
        convolution over the RAW tensor with the norm in its loader (one n block: the fold is evaluated once, the
        normalised tensor is never written) leaves P[px][tap * N + n]; nd_conv3x3_taps_gather_nhwc adds the nine shifts."""
        NI, H, W, C = nm.src.NI, nm.src.H, nm.src.W, nm.C
        if out is None:
            out = self._new(NI, H, W, N)
        w = weight.detach()
        wt = torch.zeros(64, C, dtype=torch.float32, device=self.device)
        wt[:9 * N] = w.permute(2, 3, 0, 1).reshape(9 * N, C)
        self.keep.append(wt)
        before = self.flops
        P = self.conv(nm, wt, None, 64, 1, label=label + '.taps', fuse_gn=True)
        fl = 2 * NI * H * W * N * 9 * C
        self.flops = before + fl          # the model's flops, not the padded GEMM's
        self._emit(self.lib.nd_conv3x3_taps_gather_nhwc, [P.ptr, P.ld, bias, out.ptr, out.ld, NI, H, W, N], label + '.gather')
        self._release(P)
        return out

    def _stats_ready(self, nm):
        """Emit the deferred fold of the partial rows (the consumer reads the float64 group statistics)."""
        if nm.pending is not None:
            self._emit(self.lib.nd_groupnorm_stats_from_partials, *nm.pending)
            nm.pending = None

    def _emit_coeffs(self, nm, coefA, coefB):
        """Per-(image, channel) coefficients of a norm applied by a convolution's loader."""
        tail = [nm.norm.weight.detach().data_ptr(), nm.norm.bias.detach().data_ptr(), nm.scale_ptr, nm.shift_ptr, nm.ld_ss,
                coefA.data_ptr(), coefB.data_ptr(), nm.C]
        if nm.pending is not None:
            pa = nm.pending[0]
            self._emit(self.lib.nd_groupnorm_coeffs_from_partials,
                       pa[:6] + tail + [nm.src.NI, nm.src.H * nm.src.W, GN_GROUPS, GN_EPS], 'gn.coeffs')
            nm.pending = None
        else:
            self._emit(self.lib.nd_groupnorm_coeffs,
                       [('gnstats', nm.slot), nm.nblk] + tail + [nm.src.NI, nm.C, nm.src.H * nm.src.W, GN_GROUPS, GN_EPS],
                       'gn.coeffs')

    def _materialise(self, nm):
        """Fallback for consumers that cannot fuse the GroupNorm affine: write the normalised tensor."""
        V = 8 if self.bf16 else 4
        if nm.pending is not None and _gn_apply_coeffs() and nm.C % V == 0 and nm.src.C % V == 0 and nm.C // V <= 256 and \
                nm.src.ld % V == 0 and (nm.src2 is None or nm.src2.ld % V == 0):
            # statistics still in partial rows: ONE small launch folds them and writes the coefficients, and the apply pass
            # streams from its first instruction (no per-block fold / float64 coefficient prologue); same bits
            NI, HW = nm.src.NI, nm.src.H * nm.src.W
            coefA = torch.empty(NI * nm.C, dtype=torch.float32, device=self.device)
            coefB = torch.empty(NI * nm.C, dtype=torch.float32, device=self.device)
            self.keep += [coefA, coefB]
            self._emit_coeffs(nm, coefA, coefB)
            s2 = (None, 0, 0) if nm.src2 is None else (nm.src2.ptr, nm.src2.C, nm.src2.ld)
            out = self._new(NI, nm.src.H, nm.src.W, nm.C)
            self._emit(self.lib.nd_groupnorm_apply_coeffs_nhwc,
                       [nm.src.ptr, nm.src.C, nm.src.ld, s2[0], s2[1], s2[2], coefA.data_ptr(), coefB.data_ptr(), nm.C, out.ptr, out.ld,
                        NI, HW, _hip.GN_SILU if nm.silu else 0, self.dt], 'gn.apply')
            return out
        self._stats_ready(nm)
        s2 = (None, 0, 0) if nm.src2 is None else (nm.src2.ptr, nm.src2.C, nm.src2.ld)
        out = self._new(nm.src.NI, nm.src.H, nm.src.W, nm.C)
        args = [nm.src.ptr, nm.src.C, nm.src.ld, s2[0], s2[1], s2[2], None, 0, ('gnstats', nm.slot), nm.nblk,
                nm.norm.weight.detach().data_ptr(), nm.norm.bias.detach().data_ptr(), nm.scale_ptr, nm.shift_ptr,
                nm.ld_ss, out.ptr, out.ld, nm.src.NI, nm.src.H, nm.src.W, GN_GROUPS, GN_EPS,
                _hip.GN_SILU if nm.silu else 0, self.dt]
        self._emit(self.lib.nd_groupnorm_apply_nhwc, args, 'gn.apply')
        return out

    # ------------------------------------------------------------------------------------------------ bf16 convolution
    def _packed_bf16(self, weight, pad_c_to=None, layout=0):
        """fp32 OIHW / [N,C,1] / [N,C] weight -> bf16 MFMA-fragment order, once, on the device (the bf16 half of the
        weight ingest: the checkpoint stays fp32, the cast happens in the repack).  ``layout``: 0 = fragments of the
        32x32x16 instruction, 1 = of the 16x16x32 one (nd_conv_bf16_variant_layout of the chosen tile variant)."""
        w = weight.detach().contiguous()
        N, C = w.shape[0], w.shape[1]
        k = w.shape[2] if w.dim() == 4 else 1
        if pad_c_to is not None and pad_c_to != C:
            wp = torch.zeros((N, pad_c_to) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
            wp[:, :C] = w
            w, C = wp, pad_c_to
        n = self.lib.nd_conv_bf16_weight_elems(N, C, k)
        assert n > 0
        out = torch.empty(n, dtype=torch.bfloat16, device=self.device)
        _hip.check(self.lib.nd_repack_conv_weight_bf16(w.data_ptr(), out.data_ptr(), N, C, k, layout, self._stream()),
                   'nd_repack_conv_weight_bf16')
        return out

    def _conv_bf16(self, src, weight, bias, N, ksize, out, src2, rowbias, ld_rowbias, residual, flags, label, pad_c_to,
                   want_stats=False):
        tmp = None
        gn = [None, None, 0]
        if isinstance(src, Normed):
            nm = src
            up_ = 1 if (flags & _hip.CONV_IN_UP2X) else 0
            hw = (nm.src.H << up_) * (nm.src.W << up_)
            fl_ = 2 * nm.src.NI * hw * N * ksize * ksize * nm.C
            # fused into the conv's loader when a block never spans two images (every tile is <= 256 pixels) and the
            # launch is big enough to be tuned (ND_FUSE_GN=0 switches the fold off); else the bf16 apply pass
            # every 256-channel n block re-evaluates the affine (+SiLU) of its input tile, so the fold only pays for
            # narrow outputs (ND_FUSE_GN_MAXNB n blocks at most)
            maxnb = int(os.environ.get('ND_FUSE_GN_MAXNB', '2'))
            if _fuse_gn_mode() >= 1 and hw >= 256 and hw % 256 == 0 and rowbias is None and fl_ >= 2e8 and \
                    _autotune_enabled() and (N + 255) // 256 <= maxnb:
                src, src2 = nm.src, nm.src2
                coefA = torch.empty(nm.src.NI * nm.C, dtype=torch.float32, device=self.device)
                coefB = torch.empty(nm.src.NI * nm.C, dtype=torch.float32, device=self.device)
                self.keep += [coefA, coefB]
                self._emit_coeffs(nm, coefA, coefB)
                gn = [coefA.data_ptr(), coefB.data_ptr(), nm.C]
                if nm.silu:
                    flags |= _hip.CONV_GN_SILU
            else:
                tmp = self._materialise(nm)          # bf16 apply pass (HBM-bound, half the bytes of the fp32 one)
                src, src2 = tmp, None
        up = 1 if (flags & _hip.CONV_IN_UP2X) else 0
        NI, H, W = src.NI, src.H << up, src.W << up
        if out is None:
            out = self._new(NI, H, W, N)
        if not out.bf16:
            flags |= _hip.CONV_OUT_F32
        C1 = 0 if src2 is None else src2.C
        W_SLOT = 6                                   # position of the packed-weight pointer in the argument list
        head = [src.ptr, src.C, src.ld, None if src2 is None else src2.ptr, C1, 0 if src2 is None else src2.ld,
                None, bias, rowbias, ld_rowbias, None if residual is None else residual.ptr,
                0 if residual is None else residual.ld, out.ptr, out.ld, NI, H, W, N, ksize, flags]
        fl = 2 * NI * H * W * N * ksize * ksize * (src.C + C1)
        key = ('bf16', NI, H, W, src.C + C1, N, ksize, flags, rowbias is not None, residual is not None, gn[0] is not None)
        # (1x1: the two-blocks-per-CU GEMM form writes the rows too -- the attention block's output projection)
        stats_ok = want_stats and _bf16_epilogue_stats() and out.bf16 and out.ld == N and \
            (ksize == 3 or (rowbias is None and gn[0] is None))
        if stats_ok:
            key = key + ('stats',)
        var, mode = self._pick_bf16(key, fl, head, gn, weight, pad_c_to, W_SLOT, out if stats_ok else None)
        with_stats = mode == 'stats'
        wq = self._packed_bf16(weight, pad_c_to, self.lib.nd_conv_bf16_variant_layout(var))
        self.keep.append(wq)
        self.packed_floats += wq.numel() // 2
        head[W_SLOT] = wq.data_ptr()
        if isinstance(mode, tuple):
            # too few output tiles to fill the chip: split over K into a shared fp32 workspace + a deterministic reduce
            splits = mode[1]
            need = self.lib.nd_conv_bf16_splitk_workspace_floats(NI, H, W, N, src.C + C1, ksize, splits)
            if need <= 0:
                raise _hip.NdHipError('nd_conv_bf16_splitk_workspace_floats: ' + _hip.last_error())
            self._splitk_floats = max(self._splitk_floats, need)
            self._emit(self.lib.nd_conv_bf16_splitk_nhwc, head + [var, splits, ('splitk', 0)], label, flops=fl,
                       variant=('bf16', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        elif with_stats:
            # the conv's epilogue leaves the next GroupNorm's partial statistics behind (no pass over `out`)
            rows = self.lib.nd_conv_bf16_stats_rows(NI, H, W, N, var)
            assert rows > 0
            ph = ('chpart', self._cs_floats)
            out.cs = (ph, rows)
            self._cs_floats += (NI * rows * 2 * N + 3) // 4 * 4
            self._emit(self.lib.nd_conv3x3_bf16_stats_nhwc if ksize == 3 else self.lib.nd_conv1x1_bf16_stats_nhwc,
                       head[:-2] + [flags, var] + gn + [ph], label, flops=fl,
                       variant=('bf16', var), ksize=ksize, shape=(NI, H, W, src.C + C1, N))
        else:
            self._emit(self.lib.nd_conv_bf16_nhwc, head + [var] + gn, label, flops=fl, variant=('bf16', var), ksize=ksize,
                       shape=(NI, H, W, src.C + C1, N))
        self.flops += fl
        self.conv_flops[label] = self.conv_flops.get(label, 0) + fl
        if tmp is not None:
            self._release(tmp)
        return out

    def _pick_bf16(self, key, flops, head, gn, weight, pad_c_to, w_slot, stats_out=None):
        """(tile variant, mode) for one bf16 conv launch, mode = 'plain' | 'stats' | ('splitk', splits): -1 (the library's cost model) for tiny launches or with
        ND_AUTOTUNE=0, else measured like the fp32 path (best of two bursts of 6 launches per variant that fits), cached per
        shape.  ``stats_out`` (the output Act of a conv feeding a GroupNorm): the variants that can leave the norm's partial
        statistics behind are also measured doing so, and that form is taken when it beats the best plain variant plus the
        measured statistics pass over the output."""
        if not _autotune_enabled() or flops < 2e8:
            return -1, 'plain'
        # the knobs that decide which forms may be chosen are part of the key: a choice tuned (or loaded from an
        # ND_TUNE_CACHE file) under other settings must not override ND_BF16_SPLITK=0 / ND_BF16_EPILOGUE_STATS=0
        ck = (self.device.index,) + key + ('sk%d' % _bf16_splitk(), 'es%d' % _bf16_epilogue_stats(),
                                           'gnnb' + os.environ.get('ND_FUSE_GN_MAXNB', ''))
        if ck in _TUNED:
            c = _TUNED[ck]
            if c[0] == 'bf16+splitk':
                return c[1], ('splitk', c[2])
            return c[1], ('stats' if c[0] == 'bf16+stats' else 'plain')
        stream = self._stream()
        fn = self.lib.nd_conv_bf16_nhwc
        ksize_ = head[-2]
        stats_fn = self.lib.nd_conv3x3_bf16_stats_nhwc if ksize_ == 3 else self.lib.nd_conv1x1_bf16_stats_nhwc

        def measure(f, args):
            if not _CLOCK_SETTLED[0]:
                t0 = time.time()
                while time.time() - t0 < 1.0:
                    for _ in range(8):
                        f(*args, stream)
                    torch.cuda.synchronize()
                _CLOCK_SETTLED[0] = True
            t = None
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    f(*args, stream)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / 6
                t = ms if t is None else min(t, ms)
            return t

        best, best_ms = -1, None
        sbest, sbest_ms = -1, None
        packed = {}                                       # tuning copies of the weights, one per fragment layout
        scratch = None
        for v in range(self.lib.nd_conv_bf16_num_variants()):
            lay = self.lib.nd_conv_bf16_variant_layout(v)
            if lay not in packed:
                packed[lay] = self._packed_bf16(weight, pad_c_to, lay)
            h = list(head)
            h[w_slot] = packed[lay].data_ptr()
            args = h + [v] + gn
            if fn(*args, stream) != 0:
                continue                                  # this tile shape does not fit the problem
            t = measure(fn, args)
            if best_ms is None or t < best_ms:
                best, best_ms = v, t
            if stats_out is not None:
                o = stats_out
                rows = self.lib.nd_conv_bf16_stats_rows(o.NI, o.H, o.W, o.C, v)
                if rows > 0:
                    need = o.NI * rows * 2 * o.C
                    if scratch is None or scratch.numel() < need:
                        scratch = torch.empty(need, dtype=torch.float32, device=self.device)
                    sargs = h[:-2] + [h[-1], v] + gn + [scratch.data_ptr()]
                    if stats_fn(*sargs, stream) == 0:
                        t = measure(stats_fn, sargs)
                        if sbest_ms is None or t < sbest_ms:
                            sbest, sbest_ms = v, t
        if best_ms is None:
            # no tile variant takes this launch with these fused options: leave it to the library's own selection (which
            # reports the real reason if it cannot run it either); nothing is cached
            return -1, 'plain'
        choice = ('bf16', best)
        cost = best_ms
        pass_ms = 0.0
        if stats_out is not None:
            o = stats_out
            nblk = self.lib.nd_groupnorm_stats_blocks(o.NI, o.H * o.W, o.C, self.dt)
            part = torch.empty(o.NI * nblk * GN_GROUPS * 2, dtype=torch.float64, device=self.device)
            pargs = [o.ptr, o.C, o.ld, None, 0, 0, None, 0, part.data_ptr(), o.NI, o.H * o.W, GN_GROUPS, self.dt]
            pass_ms = measure(self.lib.nd_groupnorm_stats_nhwc, pargs)
            cost = best_ms + pass_ms
        if sbest >= 0:
            o = stats_out
            fold = torch.empty(o.NI * GN_GROUPS * 2, dtype=torch.float64, device=self.device)
            rows = self.lib.nd_conv_bf16_stats_rows(o.NI, o.H, o.W, o.C, sbest)
            fargs = [scratch.data_ptr(), o.C, rows, None, 0, 0, fold.data_ptr(), o.NI, GN_GROUPS]
            fold_ms = measure(self.lib.nd_groupnorm_stats_from_partials, fargs)
            if os.environ.get('ND_TUNE_VERBOSE', '0') == '1':
                print('[tune] %s: plain v%d %.4f ms + pass %.4f | stats v%d %.4f ms + fold %.4f' %
                      (key[1:7], best, best_ms, pass_ms, sbest, sbest_ms, fold_ms), file=sys.stderr)
            if sbest_ms + fold_ms < cost or _bf16_epilogue_stats() == 2:
                choice, cost = ('bf16+stats', sbest), sbest_ms + fold_ms
        # split over K: layers with few output tiles and a long contraction (8x8 / 16x16 maps)
        NI_, H_, W_, N_, flags_ = head[14], head[15], head[16], head[17], head[19]
        M_ = NI_ * H_ * W_
        Cin = head[1] + head[4]
        if _bf16_splitk() and _bf16_epilogue_stats() != 2 and M_ <= 8192 and Cin >= 512 and gn[0] is None and N_ % 4 == 0 and \
                not (flags_ & (_hip.CONV_IN_UP2X | _hip.CONV_RES_UP2X | _hip.CONV_OUT_F32)):
            splits = (2, 4)           # 8 splits on the 1024-pixel layers measured equal to 4
            ws = torch.empty(max(self.lib.nd_conv_bf16_splitk_workspace_floats(NI_, H_, W_, N_, Cin, head[18], S) for S in splits),
                             dtype=torch.float32, device=self.device)
            cands = [best] + [v for v in (19, 7, 5, 11) if v != best and v < self.lib.nd_conv_bf16_num_variants()]
            for v in cands:
                lay = self.lib.nd_conv_bf16_variant_layout(v)
                h = list(head)
                h[w_slot] = packed[lay].data_ptr()
                for S in splits:
                    kargs = h + [v, S, ws.data_ptr()]
                    if self.lib.nd_conv_bf16_splitk_nhwc(*kargs, stream) != 0:
                        continue
                    t = measure(self.lib.nd_conv_bf16_splitk_nhwc, kargs)
                    if os.environ.get('ND_TUNE_VERBOSE', '0') == '1':
                        print('[tune] %s: split-K v%d x%d %.4f ms (+ pass %.4f) vs %.4f' % (key[1:7], v, S, t, pass_ms, cost),
                              file=sys.stderr)
                    if t + pass_ms < cost or (_bf16_splitk() == 2 and choice[0] != 'bf16+splitk'):
                        choice, cost = ('bf16+splitk', v, S), t + pass_ms
        _TUNED[ck] = choice
        if choice[0] == 'bf16+splitk':
            return choice[1], ('splitk', choice[2])
        return choice[1], ('stats' if choice[0] == 'bf16+stats' else 'plain')

    def _pick_impl(self, key, flops, weight, pad_c_to, head, tail, flags, gn, single=True, out=None, stats_wanted=False):
        """(kind, variant) for one conv launch.  Measured on the device: two bursts of 6 launches per candidate -- every direct
        tile shape that fits and, for 3x3 on even sizes, the Winograd variants -- best kept and cached per shape.
        ND_AUTOTUNE=0 falls back to the library's cost model (direct kernel); ND_WINOGRAD=0 excludes Winograd."""
        NI, H, W, C, N, ksize, _, has_rb, _, _ = key
        heur = ('direct', self.lib.nd_conv_select_variant(NI, H, W, N, ksize, flags, 1 if has_rb else 0))
        # layers with few output pixels and a long contraction may run split over K: those are worth measuring from 2e7 flops
        splitk_ok = _f32_splitk() and NI * H * W <= 8192 and C >= 128 and C % 32 == 0 and N % 4 == 0 and gn[0] is None and \
            not (flags & _hip.CONV_RES_UP2X) and (pad_c_to is None or pad_c_to % 32 == 0)
        if not _autotune_enabled() or flops < (2e7 if splitk_ok else 2e8):
            return heur
        f4_ok = ksize == 3 and _winograd_f4() and single and gn[0] is None and C % 32 == 0 and \
            (pad_c_to is None or pad_c_to % 32 == 0) and self.lib.nd_conv_winograd_f4_stats_rows(0, NI, H, W) > 0
        # 1x1 whose output feeds a norm that reads per-channel partial rows: gemm4_kernel's epilogue can leave them
        st1 = ksize == 1 and stats_wanted and not has_rb and flags == 0 and out is not None and \
            self.lib.nd_conv1x1_stats_rows(NI, H, W, N) > 0
        ck = (self.device.index,) + key + (('sk%d' % _f32_splitk(),) if splitk_ok else ()) + \
            (('f4%d%s' % (_winograd_f4(), 's' if stats_wanted else ''),) if f4_ok else ()) + (('s1',) if st1 else ())
        if ck in _TUNED:
            c = _TUNED[ck]
            return (c[0], (c[1], c[2])) if c[0].endswith('+splitk') else c
        stream = self._stream()

        def time_it(fn, args):
            if fn(*args, stream) != 0:
                return None                           # this tile shape does not fit the problem
            if not _CLOCK_SETTLED[0]:
                # the shader clock needs about a second of load to settle; candidates timed before that look 10 % slow
                t0 = time.time()
                while time.time() - t0 < 1.0:
                    for _ in range(8):
                        fn(*args, stream)
                    torch.cuda.synchronize()
                _CLOCK_SETTLED[0] = True
            fn(*args, stream)
            best_t = None
            for _ in range(2):                        # min over 2 bursts of 6 back-to-back launches: the loop being
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # tuned for runs
                e0.record()                           # the matrix pipe continuously, where the sustained clock
                for _ in range(6):                    # (2.0-2.15 GHz) differs from that of an isolated launch
                    fn(*args, stream)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / 6
                best_t = t if best_t is None else min(best_t, t)
            return best_t

        best, best_ms = heur, None
        # a candidate that leaves no per-channel statistics behind costs the norms that read its output one pass over it
        # (nd_groupnorm_channel_partials_nhwc); only conv_wino4_kernel and conv_wf4_kernel (one pass, not split) write them
        pass_ms = 0.0
        if stats_wanted and (f4_ok or st1) and out is not None:
            nb = self.lib.nd_groupnorm_stats_blocks(NI, H * W, N, self.dt)
            if nb > 0:
                rows_t = torch.empty(NI * nb * 2 * N, dtype=torch.float32, device=self.device)
                pass_ms = time_it(self.lib.nd_groupnorm_channel_partials_nhwc, [out.ptr, N, out.ld, rows_t.data_ptr(), NI, H * W, self.dt]) or 0.0
                del rows_t
        wp = self._packed(weight, pad_c_to)
        self.keep.pop()                               # tuning copy; the chosen kind is packed again by the caller
        self.packed_floats -= wp.numel()
        for v in range(self.lib.nd_conv_num_variants()):
            if st1 and v in (14, 15):
                continue                              # timed below WITH their statistics epilogue
            ms = time_it(self.lib.nd_conv_nhwc, head + [wp.data_ptr()] + tail + [ksize, flags, v] + gn)
            if ms is not None and (best_ms is None or ms + pass_ms < best_ms):
                best, best_ms = ('direct', v), ms + pass_ms
        if st1:
            # nd_conv1x1_stats_nhwc picks the 256- or 128-pixel form itself; no pass over the output to add
            rows1 = self.lib.nd_conv1x1_stats_rows(NI, H, W, N)
            sbuf1 = torch.empty(NI * rows1 * 2 * N, dtype=torch.float32, device=self.device)
            ms = time_it(self.lib.nd_conv1x1_stats_nhwc, head + [wp.data_ptr(), tail[0]] + tail[3:] + [flags, sbuf1.data_ptr()])
            del sbuf1
            if ms is not None and (best_ms is None or ms < best_ms):
                best, best_ms = ('direct', 14), ms
        if ksize == 3 and H % 2 == 0 and W % 2 == 0 and os.environ.get('ND_WINOGRAD', '1') != '0':
            wq = self._packed_wino(weight, pad_c_to)
            for v in range(self.lib.nd_conv_winograd_num_variants()):      # (retired variant numbers refuse the launch)
                # every candidate is priced as the plan would run it: a kernel that can leave the output's statistics behind
                # (conv_wino4_kernel) is timed WITH that epilogue when they are wanted (it used to be timed without,
                # i.e. under-priced against the F(4x4) and 1x1 candidates, which are timed with theirs); the others pay the pass
                rows_v = self.lib.nd_conv_winograd_stats_rows(v, NI, H, W) if (stats_wanted and pass_ms > 0 and not gn[0]) else 0
                if rows_v > 0 and self.lib.nd_conv_winograd_variant_name(v) == b'nd::conv_wino4_kernel':
                    sbuf_v = torch.empty(NI * rows_v * 2 * N, dtype=torch.float32, device=self.device)
                    ms = time_it(self.lib.nd_conv3x3_winograd_vstats_nhwc, head + [wq.data_ptr()] + tail + [flags, v, sbuf_v.data_ptr()])
                    del sbuf_v
                else:
                    ms = time_it(self.lib.nd_conv3x3_winograd_nhwc, head + [wq.data_ptr()] + tail + [flags, v] + gn)
                    if ms is not None and self.lib.nd_conv_winograd_variant_name(v) != b'nd::conv_wino4_kernel':
                        ms += pass_ms
                if ms is not None and (best_ms is None or ms < best_ms):
                    best, best_ms = ('wino', v), ms
            if splitk_ok:
                # conv_wino4_kernel split over K: 8x8 maps at batch 64 give 384 blocks of 24 chunks for 512 slots (half the CUs
                # run two blocks, half one); two block rows of 12 chunks are three blocks per CU
                names = [self.lib.nd_conv_winograd_variant_name(v) for v in range(self.lib.nd_conv_winograd_num_variants())]
                v4 = names.index(b'nd::conv_wino4_kernel')
                ws = torch.empty(max(max(self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, C, 3, S), 4) for S in (2, 4)),
                                 dtype=torch.float32, device=self.device)
                for S in (2, 4):
                    ms = time_it(self.lib.nd_conv3x3_winograd_splitk_nhwc, head + [wq.data_ptr()] + tail + [flags, v4, S, ws.data_ptr()])
                    if ms is not None:
                        ms += pass_ms
                    if ms is not None and (best_ms is None or ms < best_ms or (_f32_splitk() == 2 and not best[0].endswith('+splitk'))):
                        best, best_ms = ('wino+splitk', v4, S), ms
                del ws
            del wq
        if splitk_ok:
            splits = (2, 4, 8)
            ws = torch.empty(max(max(self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, C, ksize, S), 4) for S in splits),
                             dtype=torch.float32, device=self.device)
            cands = [7, 8, 6, 5] + ([best[1]] if best[0] == 'direct' and best[1] < 9 and best[1] not in (7, 8, 6, 5) else [])
            for v in cands:
                for S in splits:
                    ms = time_it(self.lib.nd_conv_splitk_nhwc, head + [wp.data_ptr()] + tail + [ksize, flags, v, S, ws.data_ptr()])
                    if ms is not None:
                        ms += pass_ms
                    if ms is not None and (best_ms is None or ms < best_ms or (_f32_splitk() == 2 and not best[0].endswith('+splitk'))):
                        best, best_ms = ('direct+splitk', v, S), ms
            del ws
        if f4_ok:
            # Winograd F(4x4,3x3): one pass, and -- where the m tiles do not fill the chip (16x16 and 8x8 maps at batch 64:
            # 384 / 128 workgroups for 256 CUs) -- split over K
            wq = self._packed_wf4(weight, pad_c_to)
            f4_args = head[:3] + [wq.data_ptr()] + tail + [flags, 0]
            force = _winograd_f4() == 2 and not best[0].startswith('wf4')
            rows1 = self.lib.nd_conv_winograd_f4_stats_rows(0, NI, H, W) if stats_wanted else 0
            sbuf1 = torch.empty(NI * rows1 * 2 * N, dtype=torch.float32, device=self.device) if rows1 > 0 else None
            ms = time_it(self.lib.nd_conv3x3_winograd_f4_nhwc, f4_args + [None if sbuf1 is None else sbuf1.data_ptr(), 1, None])
            del sbuf1
            if ms is not None and (best_ms is None or ms < best_ms or force):
                best, best_ms = ('wf4', 0), ms
            if _f32_splitk() and NI * H * W <= 32768 and C >= 256 and N % 4 == 0 and not (flags & _hip.CONV_RES_UP2X):
                ws = torch.empty(max(max(self.lib.nd_conv_splitk_workspace_floats(NI, H, W, N, C, 3, S), 4) for S in (2, 4)),
                                 dtype=torch.float32, device=self.device)
                # (with statistics wanted the reduce pass writes them where the map allows: timed that way, no pass to add)
                srows = self.lib.nd_conv_winograd_f4_splitk_stats_rows(0, NI, H, W) if stats_wanted else 0
                sbuf = torch.empty(NI * srows * 2 * N, dtype=torch.float32, device=self.device) if srows > 0 else None
                extra = 0.0 if srows > 0 else pass_ms
                for S in (2, 4):
                    ms = time_it(self.lib.nd_conv3x3_winograd_f4_nhwc, f4_args + [None if sbuf is None else sbuf.data_ptr(), S, ws.data_ptr()])
                    if ms is not None and (best_ms is None or ms + extra < best_ms):
                        best, best_ms = ('wf4+splitk', 0, S), ms + extra
                del ws, sbuf
            del wq
        del wp
        _TUNED[ck] = best
        return (best[0], (best[1], best[2])) if best[0].endswith('+splitk') else best

    def linear(self, src_ptr, M, K, weight, bias, out_ptr, N, flags=0, label='linear'):
        assert K % 4 == 0
        args = [src_ptr, K, K, None, 0, 0, self._packed(weight).data_ptr(),
                None if bias is None else bias.detach().data_ptr(),
                None, 0, None, 0, out_ptr, N, 1, 1, M, N, 1, flags, -1, None, None, 0]
        var = self.lib.nd_conv_select_variant(1, 1, M, N, 1, flags, 0)
        self._emit(self.lib.nd_conv_nhwc, args, label, flops=2 * M * N * K, variant=('direct', var), ksize=1)
        self.flops += 2 * M * N * K

    def groupnorm(self, src, norm, src2=None, scale_ptr=None, shift_ptr=None, ld_ss=0, silu=True, pool=False,
                  label='gn'):
        """GroupNorm(32) of src (concatenated with src2).  Emits the statistics pass and the coefficient kernel and
        returns a ``Normed`` (applied by the consuming conv's loader); with ``pool`` the activated tensor is average
        pooled, which needs the explicit apply kernel, and an Act is returned."""
        C = src.C + (0 if src2 is None else src2.C)
        NI, H, W = src.NI, src.H, src.W
        s2 = (None, 0, 0) if src2 is None else (src2.ptr, src2.C, src2.ld)
        if NI * H * W * C <= _gn_fused_max_elems(self.bf16) and C // GN_GROUPS <= 64 and (not pool or (H % 2 == 0 and W % 2 == 0)):
            # small tensor: statistics and apply in one launch; the normalised tensor is materialised
            out = self._new(NI, H // 2 if pool else H, W // 2 if pool else W, C)
            flags = (_hip.GN_SILU if silu else 0) | (_hip.GN_POOL2 if pool else 0)
            self._emit(self.lib.nd_groupnorm_fused_nhwc,
                       [src.ptr, src.C, src.ld, s2[0], s2[1], s2[2], norm.weight.detach().data_ptr(),
                        norm.bias.detach().data_ptr(), scale_ptr, shift_ptr, ld_ss, out.ptr, out.ld, NI, H, W, GN_GROUPS, GN_EPS,
                        flags, self.dt], label + '.fused')
            return out
        # a concatenation of which ONE half carries epilogue statistics (the skip tensor of a Downsample layer has none): the
        # other half gets its rows from a pass over that half alone, so the rows the producing conv paid for are used and the
        # statistics pass does not re-read the half that has them (the tuner credits the '+stats' conv with that pass)
        mixed = src2 is not None and (src.cs is None) != (src2.cs is None)
        if _gn_partials_enabled(self.bf16) or mixed:
            # every source that does not carry per-channel partial sums yet gets them from ONE pass over that tensor alone;
            # they stay with the activation (skip tensors are normalised again in the up path, concatenated: no re-read)
            for a in (src, src2):
                if a is not None and a.cs is None and a.C % (8 if self.bf16 else 4) == 0 and a.ld % (8 if self.bf16 else 4) == 0:
                    nb = self.lib.nd_groupnorm_stats_blocks(NI, H * W, a.C, self.dt)
                    if nb <= 0:
                        continue
                    ph = ('chpart', self._cs_floats)
                    self._cs_floats += (NI * nb * 2 * a.C + 3) // 4 * 4
                    self._emit(self.lib.nd_groupnorm_channel_partials_nhwc, [a.ptr, a.C, a.ld, ph, NI, H * W, self.dt],
                               label + '.channel_partials')
                    a.cs = (ph, nb)
        from_conv = src.cs is not None and (src2 is None or src2.cs is not None)
        nblk = 1 if from_conv else self.lib.nd_groupnorm_stats_blocks(NI, H * W, C, self.dt)
        assert nblk > 0
        slot = self._gn_doubles                       # float64 offset of this norm's partials [NI][nblk][32][2]
        self._gn_doubles += NI * nblk * GN_GROUPS * 2
        self._gn_slots += 1
        pending = None
        if from_conv:
            # the fold of the partial rows is emitted by the consumer: a convolution that applies the norm in its loader
            # takes it together with the coefficients (nd_groupnorm_coeffs_from_partials, one launch instead of two)
            pending = ([src.cs[0], src.C, src.cs[1], None if src2 is None else src2.cs[0], 0 if src2 is None else src2.C,
                        0 if src2 is None else src2.cs[1], ('gnstats', slot), NI, GN_GROUPS], label + '.stats_from_partials')
            if pool or not _gn_merge_coeffs():
                self._emit(self.lib.nd_groupnorm_stats_from_partials, *pending)
                pending = None
        else:
            stats_args = [src.ptr, src.C, src.ld, s2[0], s2[1], s2[2], None, 0, ('gnstats', slot), NI, H * W, GN_GROUPS,
                          self.dt]
            self._emit(self.lib.nd_groupnorm_stats_nhwc, stats_args, label + '.stats')
        if pool:
            out = self._new(NI, H // 2, W // 2, C)
            flags = (_hip.GN_SILU if silu else 0) | _hip.GN_POOL2
            apply_args = [src.ptr, src.C, src.ld, s2[0], s2[1], s2[2], None, 0, ('gnstats', slot), nblk,
                          norm.weight.detach().data_ptr(), norm.bias.detach().data_ptr(), scale_ptr, shift_ptr, ld_ss,
                          out.ptr, out.ld, NI, H, W, GN_GROUPS, GN_EPS, flags, self.dt]
            self._emit(self.lib.nd_groupnorm_apply_nhwc, apply_args, label + '.apply')
            return out
        return Normed(src=src, src2=src2, C=C, silu=silu, norm=norm, scale_ptr=scale_ptr, shift_ptr=shift_ptr,
                      ld_ss=ld_ss, slot=slot, nblk=nblk, pending=pending)

    # ------------------------------------------------------------------------------------------------ K1/K2
    def _emit_embedding(self, NR, t_ptr, y_ptr, e_ptr):
        """Emit K1/K2 for NR rows: t[NR] (int64) and y[NR] -> e[NR][e_ld], the embedding projections of every residual
        block (model.py:346-352 and the ``step_embedding`` Linear of each block, model.py:155-161,197).  Used with NR = NI
        inside the forward and with NR = steps * NI by ``embed_table``; returns the scratch buffers (kept alive by the
        caller)."""
        m, lib = self.model, self.lib
        f32 = dict(dtype=torch.float32, device=self.device)
        mc = m.model_channels
        ed = 4 * mc
        temb = torch.zeros(NR * mc, **f32)
        h1 = torch.empty(NR * ed, **f32)
        emb = torch.empty(NR * ed, **f32)
        semb = torch.empty(NR * ed, **f32)

        def packed(key, w):
            if key not in self._embed_packed:
                self._embed_packed[key] = self._packed(w)
            return self._embed_packed[key]

        def gemm(src, K, wkey, w, b, out, N, flags, label):
            # the tile variant (and with it the order of the K sum) is the one the forward's own NR = NI launch selects, also
            # when embed_table runs the same GEMM on steps * NI rows: a table row is then bit for bit what the forward leaves
            # in e_all, and a chain resumed in parts equals the chain run in one piece
            var = lib.nd_conv_select_variant(1, 1, self.NI, N, 1, flags, 0)
            args = [src.data_ptr(), K, K, None, 0, 0, packed(wkey, w).data_ptr(), b.detach().data_ptr(),
                    None, 0, None, 0, out, N, 1, 1, NR, N, 1, flags, var, None, None, 0]
            self._emit(lib.nd_conv_nhwc, args, label, flops=2 * NR * N * K, variant=('direct', var), ksize=1)
            self.flops += 2 * NR * N * K

        self._emit(lib.nd_timestep_embed, [t_ptr, self.freqs.data_ptr(), NR, mc, temb.data_ptr(), mc], 'temb')
        l0, l2 = m.step_embed[0], m.step_embed[2]
        gemm(temb, mc, 'l0', l0.weight, l0.bias, h1.data_ptr(), ed, _hip.CONV_SILU_OUT, 'step_embed.0')
        gemm(h1, ed, 'l2', l2.weight, l2.bias, emb.data_ptr(), ed, 0, 'step_embed.2')
        if m.conditional:
            tab = m.class_embedding.weight.detach()
            self._emit(lib.nd_embedding_add_silu, [emb.data_ptr(), tab.data_ptr(), y_ptr, tab.shape[0], NR, ed,
                                                   semb.data_ptr()], 'class_emb')
        else:
            self._emit(lib.nd_embedding_add_silu, [emb.data_ptr(), None, None, 0, NR, ed, semb.data_ptr()], 'emb_silu')
        if e_ptr is not None:
            gemm(semb, ed, 'e', self.e_w, self.e_b, e_ptr, self.e_ld, 0, 'step_embedding.all')
        return [temb, h1, emb, semb]

    def embed_table_bytes(self, rows):
        """Table + the K1/K2 scratch rows of ``embed_table`` for a chain of ``rows`` steps."""
        mc = self.model.model_channels
        return 4 * rows * self.NI * (self.e_ld + 13 * mc) + 16 * rows * self.NI

    def drop_embed_table(self):
        """Release the chain table (and its scratch); a graph captured on it must not be replayed afterwards."""
        self._etab = None

    def embed_table(self, t_rows):
        """K1/K2 of a whole chain at once: ``t_rows`` (int64 [S], the model timestep of every step index the chain will
        visit) and the labels in ``y_in`` -> fp32 table [S][NI * e_ld] whose row r is bit for bit what the forward's own K1/K2
        launches would leave in ``e_all`` at t_rows[r] (the same kernels AND tile variants on S * NI rows).  The step body then copies one row
        (nd_copy_row_by_step) and calls run(skip_embed=True).  Launched on the current stream; storage and launch list
        are cached per S, so a captured graph that reads the table stays valid across calls."""
        self._require_current_device()
        assert self.e_all is not None
        S = int(t_rows.numel())
        NI = self.NI
        et = self._etab
        if et is None or et['S'] != S:
            dev = self.device
            et = dict(S=S, t=torch.zeros(S * NI, dtype=torch.int64, device=dev),
                      y=None if self.y_in is None else torch.zeros(S * NI, dtype=torch.int64, device=dev),
                      table=torch.empty(S * NI * self.e_ld, dtype=torch.float32, device=dev))
            saved = (self.ops, self.meta, self.flops)
            self.ops, self.meta = [], []
            try:
                et['bufs'] = self._emit_embedding(S * NI, et['t'].data_ptr(), None if et['y'] is None else et['y'].data_ptr(),
                                                  et['table'].data_ptr())
                et['ops'] = self.ops
            finally:
                self.ops, self.meta, self.flops = saved
            self._etab = et
        et['t'].view(S, NI).copy_(t_rows.to(self.device).view(S, 1).expand(S, NI))
        if et['y'] is not None:
            et['y'].view(S, NI).copy_(self.y_in.view(1, NI).expand(S, NI))
        stream = self._stream()
        for fn, args, label in et['ops']:
            rc = fn(*args, stream)
            if rc != 0:
                raise _hip.NdHipError('{} ({}, embed_table) failed: {}'.format(fn.__name__, label, _hip.last_error()))
        return et['table']

    # ------------------------------------------------------------------------------------------------ build
    def _build(self):
        m = self.model
        lib = self.lib
        NI, R = self.NI, self.R
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        mc = m.model_channels
        ed = 4 * mc
        assert mc % 4 == 0, 'model_channels must be a multiple of 4'

        # ---- K1/K2: timestep embedding MLP, class embedding, and ALL residual blocks' embedding projections at once
        half = mc // 2
        self.freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / half)).to(dev)
        res_blocks = m._residual_blocks()
        widths = [rb.step_embedding.weight.shape[0] for rb in res_blocks]
        self.e_ld = sum(widths)
        self.e_off = {}
        off = 0
        for rb, wd in zip(res_blocks, widths):
            self.e_off[id(rb)] = off
            off += wd
        self.e_all = None
        if res_blocks:
            self.e_w = torch.cat([rb.step_embedding.weight.detach() for rb in res_blocks], 0).contiguous()
            self.e_b = torch.cat([rb.step_embedding.bias.detach() for rb in res_blocks], 0).contiguous()
            self.e_all = torch.empty(NI * self.e_ld, **f32)
        self._embed_packed = {}
        self.embed_bufs = self._emit_embedding(NI, self.t_in.data_ptr(), None if self.y_in is None else self.y_in.data_ptr(),
                                               None if self.e_all is None else self.e_all.data_ptr())
        self.n_embed = len(self.ops)          # run(skip_embed=True) starts here: e_all was filled from an embed_table row
        self._etab = None

        # ---- the UNet proper
        x = Act(self.x_in, NI, R, R, self.Cin_p)
        if self.bf16:
            # x_t stays fp32 (NHWC4) for the sampler update; the UNet reads a bf16 copy padded to 8 channels (16 bytes)
            xb = Act(torch.zeros(NI * R * R * 8 // 2, **f32), NI, R, R, 8, bf16=True)
            self.keep.append(xb.t)
            self._emit(lib.nd_f32_to_bf16_rows, [self.x_in.data_ptr(), self.Cin_p, xb.ptr, 8, self.Cin, NI * R * R],
                       'x_to_bf16')
            x = xb
        skips = []
        # every downsampling block's output is a skip connection: it stays alive until the matching pop below
        for i, block in enumerate(m.downsampling):
            x = self._run_block(block, x, None, owned=False, name='downsampling.{}'.format(i))
            skips.append(x)
        x_cur = self._run_block(m.middle_block, x, None, owned=False, name='middle_block')
        for i, block in enumerate(m.upsampling):
            # torch.cat([x, xs.pop()], 1) of model.py:474 is never materialised: both sources go to the kernels
            x_cur = self._run_block(block, x_cur, skips.pop(), owned=True, skip_owned=True,
                                    name='upsampling.{}'.format(i))
        # output head: GN -> SiLU -> conv3x3 (model.py:446-449)
        h = self.groupnorm(x_cur, m.out[0], silu=True, label='out.0')
        out_act = Act(self.out, NI, R, R, self.Cout, self.Cout_p)      # fp32 in both modes
        self.conv(h, m.out[2].weight, m.out[2].bias.detach().data_ptr(), self.Cout, 3,
                  out=out_act, label='conv3x3')
        self._release(h)
        self._release(x_cur)

        # ---- GroupNorm partial-statistics arena (float64; per norm [NI][blocks][32][2], fully written by every forward:
        #      nothing to zero, no atomics)
        self.gn_stats = torch.zeros(max(64, self._gn_doubles), dtype=torch.float64, device=dev)
        base = self.gn_stats.data_ptr()
        # partial output statistics written by the position-split convs (fully rewritten every forward)
        self.ch_partials = torch.zeros(max(4, self._cs_floats), dtype=torch.float32, device=dev)
        cs_base = self.ch_partials.data_ptr()

        self.splitk_ws = torch.empty(max(4, self._splitk_floats), dtype=torch.float32, device=dev)
        sk_base = self.splitk_ws.data_ptr()

        def bind(a):
            if isinstance(a, tuple) and a and a[0] == 'splitk':
                return sk_base
            if isinstance(a, tuple) and a and a[0] == 'gnstats':
                return base + a[1] * 8
            if isinstance(a, tuple) and a and a[0] == 'chpart':
                return cs_base + a[1] * 4
            return a
        bound = []
        for fn, args, label in self.ops:
            bound.append((fn, tuple(bind(a) for a in args), label))
        self.ops = bound
        self.workspace_floats = self.pool.total
        self.buffers = self.pool.all      # owned for the plan's lifetime
        self.pool = None

    def _run_block(self, block, x, skip, owned, skip_owned=False, name=''):
        """Run one ``UsesStepsSequential``.  ``skip`` (if given) is concatenated after ``x`` on the channel axis."""
        from . import model as M
        cur, cur2 = x, skip
        cur_owned, cur2_owned = owned, skip_owned
        for j, layer in enumerate(block):
            if isinstance(layer, M.ResidualBlock):
                nxt = self._res_block(layer, cur, cur2)
            elif isinstance(layer, M.AttentionBlock):
                assert cur2 is None
                nxt = self._attn_block(layer, cur)
            elif isinstance(layer, torch.nn.Conv2d):
                assert cur2 is None
                nxt = self.conv(cur, layer.weight, layer.bias.detach().data_ptr(), layer.weight.shape[0], 3,
                                label='conv3x3', pad_c_to=cur.C, want_stats=True)
            elif isinstance(layer, M.Downsample):
                assert cur2 is None
                nxt = self._downsample(layer, cur)
            elif isinstance(layer, M.Upsample):
                assert cur2 is None
                nxt = self._upsample(layer, cur)
            else:
                raise TypeError('unsupported layer {}'.format(type(layer)))
            self.taps.append(('{}.{}'.format(name, j), len(self.ops), nxt))
            if cur_owned:
                self._release(cur)
            if cur2 is not None and cur2_owned:
                self._release(cur2)
            cur, cur2, cur_owned, cur2_owned = nxt, None, True, False
        return cur

    def _res_block(self, rb, x, x2):
        """model.py:188-211."""
        lib = self.lib
        NI = x.NI
        Cin = x.C + (0 if x2 is None else x2.C)
        Cout = rb.in_conv.weight.shape[0]
        mode = rb.resample_mode            # None | 'up' | 'down'
        assert not (mode is not None and x2 is not None)
        # h = silu(in_norm(x)); 'down' pools here, 'up' is folded into the conv's input addressing
        h0 = self.groupnorm(x, rb.in_norm, src2=x2, silu=True, pool=(mode == 'down'), label='res.in_norm')
        e_ptr = self.e_all.data_ptr() + 4 * self.e_off[id(rb)]
        adaptive = rb.use_adaptive_gn
        h1 = self.conv(h0, rb.in_conv.weight, rb.in_conv.bias.detach().data_ptr(), Cout, 3,
                       rowbias=None if adaptive else e_ptr, ld_rowbias=0 if adaptive else self.e_ld,
                       flags=_hip.CONV_IN_UP2X if mode == 'up' else 0, label='conv3x3', want_stats=True)
        self._release(h0)
        if adaptive:   # scale = first half, shift = second half (model.py:201)
            h2 = self.groupnorm(h1, rb.out_norm, scale_ptr=e_ptr, shift_ptr=e_ptr + 4 * Cout, ld_ss=self.e_ld,
                                silu=True, label='res.out_norm')
        else:
            h2 = self.groupnorm(h1, rb.out_norm, silu=True, label='res.out_norm')
        # h1 stays alive until out_conv has been emitted: h2 is applied by that conv's loader
        # skip path
        flags = 0
        res = None
        tmp = None
        if mode == 'down':
            tmp = self._new(NI, x.H // 2, x.W // 2, x.C)
            self._emit(lib.nd_avgpool2x_nhwc, [x.ptr, x.ld, tmp.ptr, tmp.ld, NI, x.H, x.W, x.C, self.dt], 'avgpool')
            xs, xs2 = tmp, None
        else:
            xs, xs2 = x, x2
        if isinstance(rb.skip, torch.nn.Conv2d):
            k = rb.skip.weight.shape[-1]
            s = self.conv(xs, rb.skip.weight, rb.skip.bias.detach().data_ptr(), Cout, k,
                          src2=xs2, flags=_hip.CONV_IN_UP2X if mode == 'up' else 0,
                          label='conv1x1' if k == 1 else 'conv3x3')
            if tmp is not None:
                self._release(tmp)
            tmp = s
            res = s
        else:
            assert xs2 is None and Cin == Cout
            res = xs
            if mode == 'up':
                flags |= _hip.CONV_RES_UP2X
        out = self.conv(h2, rb.out_conv.weight, rb.out_conv.bias.detach().data_ptr(), Cout, 3, residual=res,
                        flags=flags, label='conv3x3', want_stats=True)
        self._release(h2)
        self._release(h1)
        if tmp is not None:
            self._release(tmp)
        return out

    def _attn_block(self, ab, x):
        """model.py:260-291."""
        NI, H, W, C = x.NI, x.H, x.W, x.C
        T = H * W
        n = self.groupnorm(x, ab.norm, silu=False, label='attn.norm')
        qkv = self.conv(n, ab.qkv_nin.weight, ab.qkv_nin.bias.detach().data_ptr(), 3 * C, 1,
                        label='conv1x1')
        self._release(n)
        a = self._new(NI, H, W, C)
        nh = ab.num_heads
        hd = C // nh
        if ab.split_qkv_first:
            offs = (0, C, 2 * C, hd)
        else:
            offs = (0, hd, 2 * hd, 3 * hd)
        self._emit(self.lib.nd_attention_bf16_nhwc if self.bf16 else self.lib.nd_attention_nhwc, [qkv.ptr, qkv.ld, a.ptr, a.ld, NI, T, nh, hd, offs[0], offs[1], offs[2],
                                                offs[3], float(ab.scale)], 'attention',
                   flops=4 * NI * nh * T * T * hd)
        self.flops += 4 * NI * nh * T * T * hd
        self.conv_flops['attention'] = self.conv_flops.get('attention', 0) + 4 * NI * nh * T * T * hd
        self._release(qkv)
        out = self.conv(a, ab.proj_out.weight, ab.proj_out.bias.detach().data_ptr(), C, 1,
                        residual=x, label='conv1x1', want_stats=True)
        self._release(a)
        return out

    def _downsample(self, layer, x):
        """model.py:107-112."""
        NI = x.NI
        if layer.with_conv:
            N = layer.conv.weight.shape[0]
            w = layer.conv.weight.detach()
            assert w.is_contiguous()
            if x.H % 2 == 0 and x.W % 2 == 0:
                # stride 2 = stride 1 on the space-to-depth tensor with rearranged weights (tap index 0 of the new kernel is
                # offset -1): (dy -> parity p, tap u): 0 -> (1, 0), 1 -> (0, 1), 2 -> (1, 1); the rest is zero.  4x the
                # MACs of the strided form, on the MFMA / Winograd kernels instead of a scalar loop (79 ms -> < 1 ms for
                # 64x64x192 at B=64).
                C = x.C
                w2 = torch.zeros((N, 4, C, 3, 3), dtype=w.dtype, device=w.device)
                mp = ((1, 0), (0, 1), (1, 1))
                for dy in range(3):
                    for dx in range(3):
                        (p_, u), (q_, v) = mp[dy], mp[dx]
                        w2[:, p_ * 2 + q_, :, u, v] = w[:, :, dy, dx]
                w2 = w2.reshape(N, 4 * C, 3, 3).contiguous()
                self.keep.append(w2)
                s2d = self._new(NI, x.H // 2, x.W // 2, 4 * C)
                e = 2 if self.bf16 else 1      # pure data movement: bf16 pairs travel as fp32 words
                self._emit(self.lib.nd_space_to_depth2_nhwc, [x.ptr, x.ld // e, s2d.ptr, s2d.ld // e, NI, x.H, x.W, C // e],
                           'space_to_depth')
                out = self.conv(s2d, w2, layer.conv.bias.detach().data_ptr(), N, 3, label='conv_s2')
                self._release(s2d)
                return out
            if self.bf16:
                raise _hip.NdHipError('bf16 path: stride-2 convolution on odd sizes is not supported (use dtype fp32)')
            Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
            out = self._new(NI, Ho, Wo, N)
            self._emit(self.lib.nd_conv_direct_nhwc, [x.ptr, x.C, x.ld, w.data_ptr(),
                                                      layer.conv.bias.detach().data_ptr(), out.ptr, out.ld, NI, x.H,
                                                      x.W, N, 3, 2, 1], 'conv_s2')
            return out
        out = self._new(NI, x.H // 2, x.W // 2, x.C)
        self._emit(self.lib.nd_avgpool2x_nhwc, [x.ptr, x.ld, out.ptr, out.ld, NI, x.H, x.W, x.C, self.dt], 'avgpool')
        return out

    def _upsample(self, layer, x):
        """model.py:76-80."""
        NI = x.NI
        if layer.with_conv:
            N = layer.conv.weight.shape[0]
            return self.conv(x, layer.conv.weight, layer.conv.bias.detach().data_ptr(), N, 3,
                             flags=_hip.CONV_IN_UP2X, label='conv3x3')
        out = self._new(NI, 2 * x.H, 2 * x.W, x.C)
        e = 2 if self.bf16 else 1
        self._emit(self.lib.nd_upsample2x_nhwc, [x.ptr, x.ld // e, out.ptr, out.ld // e, NI, x.H, x.W, x.C // e], 'upsample')
        return out

    # ------------------------------------------------------------------------------------------------ run
    def _require_current_device(self):
        """Every launch goes to the CURRENT device's stream while every pointer belongs to ``self.device``: running with
        another device current would be a GPU memory fault, not a Python error."""
        if torch.cuda.current_device() != self.device.index:
            raise _hip.NdHipError('plan was built for {} but cuda:{} is current; wrap the call in '
                                  'torch.cuda.device(...)'.format(self.device, torch.cuda.current_device()))

    def run(self, skip_embed=False):
        """Launch the whole forward on the current stream: reads x_in / t_in / y_in, writes out.  ``skip_embed``: K1/K2 are
        not launched -- the caller has filled ``e_all`` with this step's row of ``embed_table``."""
        self._require_current_device()
        stream = self._stream()
        for fn, args, label in (self.ops[self.n_embed:] if skip_embed else self.ops):
            rc = fn(*args, stream)
            if rc != 0:
                raise _hip.NdHipError('{} ({}) failed: {}'.format(fn.__name__, label, _hip.last_error()))

    def run_with_taps(self):
        """Debug hook: eager run that also returns {module name: output as NCHW tensor} for every layer of every block
        (the names of the reference's forward hooks: ``downsampling.i.j``, ``middle_block.j``, ``upsampling.i.j``).
        Buffers are recycled by later launches, so each output is copied out right after its last producing launch."""
        self._require_current_device()
        stream = self._stream()
        got, k = {}, 0
        for idx, (fn, args, label) in enumerate(self.ops):
            rc = fn(*args, stream)
            if rc != 0:
                raise _hip.NdHipError('{} ({}) failed: {}'.format(fn.__name__, label, _hip.last_error()))
            while k < len(self.taps) and self.taps[k][1] == idx + 1:
                name, _, a = self.taps[k]
                flat = a.t.view(torch.bfloat16) if a.bf16 else a.t
                v = flat[:a.NI * a.H * a.W * a.ld].view(a.NI, a.H, a.W, a.ld)[..., :a.C]
                got[name] = v.permute(0, 3, 1, 2).float().contiguous()
                k += 1
        assert k == len(self.taps)
        return got

    def run_timed(self):
        """Eager run with a HIP event pair around every launch (recorded on the launch stream); returns one dict per
        launch: the plan's meta (label, entry point, flops, conv variant) plus ``ms``."""
        self._require_current_device()
        stream = self._stream()
        evs = []
        for fn, args, label in self.ops:
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args, stream)
            b.record()
            if rc != 0:
                raise _hip.NdHipError('{} ({}) failed: {}'.format(fn.__name__, label, _hip.last_error()))
            evs.append((a, b))
        torch.cuda.synchronize()
        return [dict(m, ms=a.elapsed_time(b)) for m, (a, b) in zip(self.meta, evs)]
