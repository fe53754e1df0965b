"""Gaussian diffusion sampler (DDPM / DDIM, classifier-free guidance) with the reverse loop running on the GPU.

Drop-in surface of the reference's ``nicediffusion/diffusion.py`` class ``Diffusion`` for sampling:
same constructor (diffusion.py:23-28), same public schedule attributes (float64 numpy, diffusion.py:109-130),
``denoise`` (diffusion.py:156-226) and ``diffuse`` (diffusion.py:133-153).  Differences are all below the surface:

* the per-step scalars live in one fp32 device table and the step index in a device word, instead of 4-6
  host->device ``extract`` copies per step (diffusion.py:478-496);
* one step = UNet plan + one fused sampler kernel, captured once as a hipGraph and replayed;
* x stays NHWC on the device for the whole loop; layout changes happen once at the loop's edges;
* with ``torch.distributed`` initialised, ``denoise_sharded`` splits the batch over ranks (no communication inside
  the loop) and all-gathers the finished samples over RCCL.

Training-only pieces of the reference (``loss``, VLB, classifier guidance through autograd) are out of scope.
"""
import enum
import math
import os

import numpy as np
import torch

from . import _hip


class VarType(enum.Enum):
    SMALL = enum.auto()
    LARGE = enum.auto()
    LEARNED = enum.auto()
    LEARNED_INTERPOLATION = enum.auto()

    @staticmethod
    def get_var_type(name):
        table = {'small': VarType.SMALL, 'large': VarType.LARGE, 'learned': VarType.LEARNED,
                 'learned_interpolation': VarType.LEARNED_INTERPOLATION}
        if name not in table:
            raise NotImplementedError(name)
        return table[name]


class LossType(enum.Enum):
    SIMPLE = enum.auto()
    KL = enum.auto()
    KL_RESCALED = enum.auto()
    HYBRID = enum.auto()

    @staticmethod
    def get_loss_type(name):
        table = {'simple': LossType.SIMPLE, 'KL': LossType.KL, 'KL_rescaled': LossType.KL_RESCALED,
                 'hybrid': LossType.HYBRID}
        if name not in table:
            raise NotImplementedError(name)
        return table[name]


def get_beta_schedule(schedule_method, num_steps, beta_0, beta_T):
    """Noise variances for ``num_steps`` steps: 'linear', 'constant' or 'cosine' (IDDPM eq. 17); float64."""
    if schedule_method == 'linear':
        return np.linspace(beta_0, beta_T, num_steps, dtype=np.float64)
    if schedule_method == 'constant':
        return beta_0 * np.ones(num_steps, dtype=np.float64)
    if schedule_method == 'cosine':
        def abar(u):
            return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / num_steps) / abar(i / num_steps), 0.999) for i in range(num_steps)])
    raise NotImplementedError('unimplemented variance scheduling method: {}'.format(schedule_method))


class Diffusion:
    def __init__(self, model, original_num_steps, rescaled_num_steps, sampling_var_type, loss_type, betas=None,
                 beta_schedule='linear', guidance_method=None, guidance_strength=None, classifier=None,
                 use_ddim=False, ddim_eta=None, device=None):
        self.model = model
        if device is None:
            device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        self.device = torch.device(device)
        self.model.to(self.device)
        self.model.eval()

        if guidance_method not in (None, 'classifier', 'classifier_free'):
            raise NotImplementedError(guidance_method)
        assert guidance_method is None or self.model.conditional, 'can only use guidance if model is conditional'
        self.guidance = guidance_method
        self.strength = guidance_strength
        self.classifier = classifier

        self.original_num_steps = original_num_steps
        self.rescaled_num_steps = rescaled_num_steps
        self.sampling_var_type = VarType.get_var_type(sampling_var_type)
        self.loss_type = LossType.get_loss_type(loss_type)
        if use_ddim:
            assert ddim_eta is not None, 'please supply eta if you want to use ddim'
        self.use_ddim = use_ddim
        self.ddim_eta = ddim_eta

        # ---- schedule (float64, on the host): respace the T-step chain to the kept steps (IDDPM eq. 19)
        T, S = original_num_steps, rescaled_num_steps
        if betas is None:
            betas = get_beta_schedule(beta_schedule, T, 0.0001 * 1000 / T, 0.02 * 1000 / T)
        else:
            assert len(betas) == T, 'betas must be the right length!'
            betas = np.array(betas, dtype=np.float64)
        abar_full = np.cumprod(1.0 - betas, axis=0)
        kept = list(range(T // (2 * S), T + T // (2 * S), T // S))
        kept_set = set(kept)
        respaced, prev = [], 1.0
        for i, a in enumerate(abar_full):
            if i in kept_set:
                respaced.append(1.0 - a / prev)
                prev = a
        betas = np.array(respaced)
        assert (betas > 0).all() and (betas <= 1).all(), 'betas in invalid range'

        self.betas = betas
        self.timestep_map = torch.tensor(kept, device=self.device, dtype=torch.long)
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_reciprocal_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_reciprocal_alphas_minus_one_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_mean_coef_x0 = np.sqrt(self.alphas_cumprod_prev) * betas / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef_xt = np.sqrt(alphas) * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.log_posterior_var_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))

        self.use_graph = True          # capture the step body as a hipGraph
        self.seed = None               # Philox seed for in-kernel noise; None -> drawn from torch's CPU generator
        self.first_row = 0             # row of the global batch this process's row 0 is (set by denoise_sharded)
        self._loops = {}

    # ---------------------------------------------------------------------------------------------- device tables
    def coefficient_table(self):
        """fp32 [S][8] rows consumed by nd_ddim_step / nd_ddpm_step (column meaning: include/nd_hip.h)."""
        n = len(self.betas)
        tab = np.zeros((n, _hip.COEF_COLS), dtype=np.float64)
        tab[:, 0] = self.sqrt_reciprocal_alphas_cumprod
        tab[:, 1] = self.sqrt_reciprocal_alphas_minus_one_cumprod
        tab[:, 2] = self.alphas_cumprod
        tab[:, 3] = self.alphas_cumprod_prev
        tab[:, 4] = self.posterior_mean_coef_x0
        tab[:, 5] = self.posterior_mean_coef_xt
        vt = self.sampling_var_type
        if vt == VarType.LEARNED_INTERPOLATION:
            tab[:, 6] = self.log_posterior_var_clipped
            tab[:, 7] = np.log(self.betas)
        elif vt == VarType.LARGE:
            tab[:, 6] = np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        elif vt == VarType.SMALL:
            tab[:, 6] = np.log(np.maximum(self.posterior_variance, 1e-20))
        return torch.from_numpy(tab).float()

    def _var_kind(self):
        vt = self.sampling_var_type
        if vt == VarType.LEARNED:
            return _hip.VAR_LEARNED
        if vt == VarType.LEARNED_INTERPOLATION:
            return _hip.VAR_LEARNED_INTERP
        return _hip.VAR_FIXED

    @property
    def _learned(self):
        return self.sampling_var_type in (VarType.LEARNED, VarType.LEARNED_INTERPOLATION)

    # ---------------------------------------------------------------------------------------------- forward process
    @torch.no_grad()
    def diffuse(self, x_0, steps_to_do=None, noise=None):
        """q(x_t | x_0) after ``steps_to_do`` rescaled steps (reference diffusion.py:133-153, :232-240)."""
        if steps_to_do is None or steps_to_do > self.rescaled_num_steps:
            steps_to_do = self.rescaled_num_steps
        t = steps_to_do - 1
        x_0 = x_0.to(self.device).float().contiguous()
        _hip.require_device(x_0, 'x_0')
        if noise is None:
            noise = torch.randn_like(x_0)
        noise = noise.to(self.device).float().contiguous()
        out = torch.empty_like(x_0)
        a = float(np.float32(self.sqrt_alphas_cumprod[t]))
        b = float(np.float32(self.sqrt_one_minus_alphas_cumprod[t]))
        with torch.cuda.device(x_0.device):
            _hip.check(_hip.load().nd_qsample(x_0.data_ptr(), noise.data_ptr(), out.data_ptr(), x_0.numel(), a, b,
                                              torch.cuda.current_stream().cuda_stream), 'nd_qsample')
        return out

    # ---------------------------------------------------------------------------------------------- per-step surface
    # The reference's public per-step methods (diffusion.py:232-369), for callers that walk the chain themselves (the usual
    # way to show the pred_x0 progression).  ``t`` is what the reference's loop passes (diffusion.py:216): a [B] tensor of
    # RESCALED step indices, float or integer, one per image -- the rows may differ.  One call = one eager UNet plan run
    # + one fused sampler launch; ``denoise`` remains the fast path (device step word, captured graph).
    def _step_indices(self, t, B):
        """[B] int32 device tensor of rescaled indices from the reference's float / long ``t``; a python int or 0-d tensor
        broadcasts.  torch.gather raises on an index outside the table (diffusion.py:491): so does this."""
        t = torch.as_tensor(t)
        if t.dim() == 0:
            t = t.reshape(1).expand(B)
        assert t.shape[0] == B, 'one step index per image'
        ti = t.to(self.device).long()
        lo, hi = (int(v) for v in torch.aminmax(ti))
        if lo < 0 or hi >= len(self.betas):
            raise IndexError('step index out of range: got [{}, {}], the chain has {} steps'.format(lo, hi, len(self.betas)))
        return ti.to(torch.int32).contiguous()

    def _step_tables(self):
        """Device copies of the per-step tables the per-step methods gather from (refreshed when the schedule or the
        variance type changed)."""
        key = (self.sampling_var_type, self.betas.tobytes())
        st = self.__dict__.get('_step_tabs')
        if st is None or st['key'] != key:
            dev = self.device
            st = dict(key=key, coef=self.coefficient_table().to(dev).contiguous(),
                      sa=torch.from_numpy(self.sqrt_alphas_cumprod).float().to(dev).contiguous(),
                      sb=torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).float().to(dev).contiguous(),
                      tmap=self.timestep_map.to(dev).contiguous())
            self.__dict__['_step_tabs'] = st
        return st

    @torch.no_grad()
    def diffusion_step(self, x_0, t, noise=None):
        """Sample q(x_t | x_0) with one step index per image (reference diffusion.py:232-240)."""
        x_0 = x_0.to(self.device).float().contiguous()
        _hip.require_device(x_0, 'x_0')
        B = x_0.shape[0]
        steps = self._step_indices(t, B)
        if noise is None:
            noise = torch.randn_like(x_0)
        noise = noise.to(self.device).float().contiguous()
        assert noise.shape == x_0.shape, 'noise must have the shape of x_0'
        tabs = self._step_tables()
        out = torch.empty_like(x_0)
        with torch.cuda.device(x_0.device):
            _hip.check(_hip.load().nd_qsample_steps(x_0.data_ptr(), noise.data_ptr(), out.data_ptr(), B, x_0[0].numel(),
                                                    tabs['sa'].data_ptr(), tabs['sb'].data_ptr(), steps.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), 'nd_qsample_steps')
        return out

    def _model_eps(self, x_t, steps, kwargs, guided):
        """Stage x_t / labels / timesteps into the plan of the forward batch (2B under classifier-free guidance: the
        conditional rows, then the same images with the null class, diffusion.py:280-281,343-344) and run it."""
        model = self.model
        y = (kwargs or {}).get('y')
        assert (y is not None) == model.conditional, 'pass label iff model is class-conditional'
        assert x_t.shape[2] == model.resolution and x_t.shape[3] == model.resolution, \
            'incorrect resolution: {}'.format(x_t.shape[2:])
        if self.guidance == 'classifier':
            raise NotImplementedError('classifier guidance needs autograd through a classifier (out of scope)')
        if model.conditional:
            model._check_labels(y)
        B = x_t.shape[0]
        NI = 2 * B if guided else B
        plan = model._plan(NI)
        lib = plan.lib
        C, HW = model.in_channels, model.resolution ** 2
        s = torch.cuda.current_stream().cuda_stream
        _hip.check(lib.nd_nchw_to_nhwc(x_t.data_ptr(), plan.x_in.data_ptr(), B, C, HW, plan.Cin_p, s), 'nd_nchw_to_nhwc')
        tm = self._step_tables()['tmap'][steps.long()]
        plan.t_in[:B].copy_(tm)
        if model.conditional:
            plan.y_in[:B].copy_(y.to(self.device).to(torch.int64))
        if guided:
            n = B * HW * plan.Cin_p
            plan.x_in[n:2 * n].copy_(plan.x_in[:n])
            plan.t_in[B:].copy_(tm)
            plan.y_in[B:].zero_()
        plan.run()
        return plan

    @torch.no_grad()
    def get_eps_and_log_var(self, x_t, t, kwargs):
        """(eps_pred, log_var) of the model at (x_t, t), each [B, C, R, R] (reference diffusion.py:242-264; no guidance
        mix here, as in the reference)."""
        x_t = x_t.to(self.device).float().contiguous()
        _hip.require_device(x_t, 'x_t')
        B, C, R = x_t.shape[0], self.model.in_channels, self.model.resolution
        with torch.cuda.device(x_t.device):
            steps = self._step_indices(t, B)
            plan = self._model_eps(x_t, steps, kwargs, False)
            eps = torch.empty(B, C, R, R, dtype=torch.float32, device=self.device)
            log_var = torch.empty_like(eps)
            _hip.check(plan.lib.nd_eps_log_var(plan.out.data_ptr(), plan.Cout_p, self._step_tables()['coef'].data_ptr(),
                                               steps.data_ptr(), self._var_kind(), eps.data_ptr(), log_var.data_ptr(), B,
                                               R * R, C, torch.cuda.current_stream().cuda_stream), 'nd_eps_log_var')
        return eps, log_var

    def _public_step(self, ddim, x_t, t, kwargs, clip_x, noise):
        x_t = x_t.to(self.device).float().contiguous()
        _hip.require_device(x_t, 'x_t')
        model = self.model
        B, C, R = x_t.shape[0], model.in_channels, model.resolution
        HW = R * R
        cfg = self.guidance == 'classifier_free'
        if ddim:
            assert self.ddim_eta is not None, 'please supply eta if you want to use ddim'
        with torch.cuda.device(x_t.device):
            steps = self._step_indices(t, B)
            plan = self._model_eps(x_t, steps, kwargs, cfg)
            lib = plan.lib
            s = torch.cuda.current_stream().cuda_stream
            n = B * HW * plan.Cin_p
            sample = torch.empty(n, dtype=torch.float32, device=self.device)
            pred = torch.empty(n, dtype=torch.float32, device=self.device)
            noise_ptr = None
            if noise is not None:
                nz = torch.empty(n, dtype=torch.float32, device=self.device)
                nc = noise.to(self.device).float().contiguous()
                assert tuple(nc.shape) == (B, C, R, R), 'noise must be [B, C, R, R]'
                _hip.check(lib.nd_nchw_to_nhwc(nc.data_ptr(), nz.data_ptr(), B, C, HW, plan.Cin_p, s), 'nd_nchw_to_nhwc')
                noise_ptr = nz.data_ptr()
            seed = self.seed
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            flags = _hip.STEP_PER_IMAGE | (0 if clip_x else _hip.STEP_NO_CLIP)
            eps_u = plan.out.data_ptr() + 4 * B * HW * plan.Cout_p if cfg else None
            w = float(self.strength) if cfg else 0.0
            coef = self._step_tables()['coef'].data_ptr()
            first_elem = int(self.first_row) * HW * C
            if ddim:
                rc = lib.nd_ddim_step(plan.x_in.data_ptr(), sample.data_ptr(), None, pred.data_ptr(), flags, plan.Cin_p,
                                      plan.out.data_ptr(), eps_u, plan.Cout_p, w, coef, steps.data_ptr(), float(self.ddim_eta),
                                      noise_ptr, 0, seed, None, first_elem, B, HW, C, s)
            else:
                rc = lib.nd_ddpm_step(plan.x_in.data_ptr(), sample.data_ptr(), None, pred.data_ptr(), flags, plan.Cin_p,
                                      plan.out.data_ptr(), eps_u, plan.Cout_p, w, coef, steps.data_ptr(), self._var_kind(),
                                      noise_ptr, 0, seed, None, first_elem, B, HW, C, s)
            _hip.check(rc, 'sampler step')
            outs = []
            for buf in (sample, pred):
                o = torch.empty(B, C, R, R, dtype=torch.float32, device=self.device)
                _hip.check(lib.nd_nhwc_to_nchw(buf.data_ptr(), o.data_ptr(), B, C, HW, plan.Cin_p, s), 'nd_nhwc_to_nchw')
                outs.append(o)
        return outs[0], outs[1]

    @torch.no_grad()
    def denoising_step(self, x_t, t, kwargs=None, clip_x=True, noise=None):
        """One DDPM step p(x_{t-1} | x_t): returns ``(sample, pred_x0)`` (reference diffusion.py:266-316).  ``noise``
        (extension, parity tests): the N(0,1) draw to use instead of in-kernel Philox noise."""
        return self._public_step(False, x_t, t, kwargs, clip_x, noise)

    @torch.no_grad()
    def ddim_denoising_step(self, x_t, t, kwargs=None, clip_x=True, noise=None):
        """One DDIM step: returns ``(sample, pred_x0)`` (reference diffusion.py:318-369)."""
        return self._public_step(True, x_t, t, kwargs, clip_x, noise)

    # ---------------------------------------------------------------------------------------------- reverse process
    @torch.no_grad()
    def denoise(self, x=None, kwargs=None, start_step=None, steps_to_do=None, batch_size=1, ema_params=None,
                progress=True, noise=None, trace=None, first_index=None):
        """Run the reverse chain and return x_0-ish samples [B, C, R, R] (reference diffusion.py:156-226).

        Extra keyword arguments (not in the reference): ``noise`` -- tensor [S, B, C, R, R] with the N(0,1) draw to
        use at each rescaled step (parity tests; default is in-kernel Philox noise); ``trace`` -- list that receives
        x after every step (NCHW clones; disables graph replay); ``first_index`` -- rescaled index of the first
        step to take (default ``steps_to_do - 1`` as in the reference, whose loop always ends at index 0); with it a
        test can take one teacher-forced step at any index.
        """
        if kwargs is None:
            kwargs = {}
        model = self.model
        assert ('y' in kwargs.keys() and kwargs['y'] is not None) == model.conditional, \
            'pass label iff model is class-conditional'
        if model.conditional:
            assert len(kwargs['y']) == batch_size, 'len(labels) != batch size'
        if self.guidance == 'classifier':
            raise NotImplementedError('classifier guidance needs autograd through a classifier (out of scope)')

        original = None
        if ema_params is not None:      # swap EMA weights in (diffusion.py:185-189)
            original = {}
            for name, p in model.named_parameters():
                original[name] = p.data
                p.data = ema_params[name].to(self.device)
            model.invalidate_plans()    # plans hold repacked copies of the weights that were just swapped out
            self.release()
        try:
            if start_step is None:
                start_step = self.rescaled_num_steps
            if steps_to_do is None or steps_to_do > start_step:
                steps_to_do = start_step
            if x is None:
                assert start_step == self.rescaled_num_steps, 'cannot start from noise with current step that is not T'
                x = torch.randn(batch_size, model.in_channels, model.resolution, model.resolution)
            x = x.to(self.device)
            _hip.require_device(x, 'x')
            y = kwargs.get('y')
            if model.conditional:
                model._check_labels(y)
            with torch.cuda.device(x.device):       # every launch below goes to the current device's stream
                return self._run_loop(x.float().contiguous(), y, steps_to_do, progress, noise, trace, first_index)
        finally:
            if original is not None:
                for name, p in model.named_parameters():
                    p.data = original[name]
                model.invalidate_plans()
                self.release()

    def release(self):
        """Drop the captured graph, the loop's device words and the chain's K1/K2 table (up to ND_EMBED_TABLE_MAX_GB of
        device memory held between ``denoise`` calls so that the graph can be replayed)."""
        for st in self._loops.values():
            st['graph'] = None
            st['plan'].drop_embed_table()
        self._loops = {}

    def _run_loop(self, x, y, steps_to_do, progress, noise, trace, first_index=None):
        model = self.model
        lib = _hip.load()
        B = x.shape[0]
        C, R = model.in_channels, model.resolution
        HW = R * R
        cfg = self.guidance == 'classifier_free'
        NI = 2 * B if cfg else B
        plan = model._plan(NI)
        dev = self.device
        stream = torch.cuda.current_stream().cuda_stream
        if steps_to_do <= 0:
            return x.clone()

        key = (id(plan), B)
        st = self._loops.get(key)
        if st is None or st['plan'] is not plan:
            st = dict(plan=plan, coef=None, coef_key=None, tmap=self.timestep_map.to(dev).contiguous(),
                      step=torch.zeros(1, dtype=torch.int32, device=dev),
                      seed=torch.zeros(1, dtype=torch.int64, device=dev), graph=None, graph_key=None, noise=None)
            self.release()
            self._loops = {key: st}
        # the coefficient rows depend on the schedule AND on the variance type ('large' / 'small' share a kernel
        # variance kind but not column 6): refresh them whenever either changed since they were uploaded
        coef_key = (self.sampling_var_type, self.betas.tobytes())
        if st['coef_key'] != coef_key:
            tab = self.coefficient_table().to(dev).contiguous()
            if st['coef'] is not None and st['coef'].shape == tab.shape:
                st['coef'].copy_(tab)                # same storage: a captured graph keeps working
            else:
                st['coef'], st['graph'] = tab, None
            st['coef_key'] = coef_key
            tm = self.timestep_map.to(dev).contiguous()
            if st['tmap'].shape == tm.shape:
                st['tmap'].copy_(tm)
            else:
                st['tmap'], st['graph'] = tm, None
        first = steps_to_do - 1 if first_index is None else int(first_index)
        assert 0 <= first < len(self.betas) and first - steps_to_do + 1 >= 0, 'step index out of range'
        need_noise = (not self.use_ddim) or (self.ddim_eta != 0)
        noise_ptr, noise_stride = None, 0
        if noise is not None and need_noise:
            assert noise.shape[0] > first and tuple(noise.shape[1:]) == (B, C, R, R), 'noise must be [S,B,C,R,R]'
            S = noise.shape[0]
            nb = st['noise']
            if nb is None or nb.numel() != S * B * HW * plan.Cin_p:
                nb = torch.empty(S * B * HW * plan.Cin_p, dtype=torch.float32, device=dev)
                st['noise'] = nb
                st['graph'] = None
            nz = noise.to(dev).float().contiguous()
            _hip.check(lib.nd_nchw_to_nhwc(nz.data_ptr(), nb.data_ptr(), S * B, C, HW, plan.Cin_p, stream),
                       'nd_nchw_to_nhwc')
            noise_ptr, noise_stride = nb.data_ptr(), B * HW * plan.Cin_p
        first_elem = int(self.first_row) * HW * C      # Philox counters of a shard continue the global batch's
        seed = self.seed
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        st['seed'].fill_(int(seed))     # the kernels read the Philox key from this device word: one graph, any seed
        seed_ptr = st['seed'].data_ptr()

        # ---- stage inputs: x -> NHWC (padded to 4 channels) in the plan's input buffer; labels; step counter
        _hip.check(lib.nd_nchw_to_nhwc(x.data_ptr(), plan.x_in.data_ptr(), B, C, HW, plan.Cin_p, stream),
                   'nd_nchw_to_nhwc')
        x_state = plan.x_in[:B * HW * plan.Cin_p]
        if cfg:
            plan.x_in[B * HW * plan.Cin_p:].copy_(x_state)
        # the unconditional half of a classifier-free batch: the sampler kernel writes the updated images twice
        xdup = plan.x_in.data_ptr() + 4 * B * HW * plan.Cin_p if cfg else None
        if model.conditional:
            plan.y_in[:B].copy_(y.to(torch.int64))
            if cfg:
                plan.y_in[B:].zero_()              # null class = label 0 (diffusion.py:281,344)
        st['step'].fill_(first)

        # K1/K2 depend on (t, y) only: evaluated for every step of this chain in one batched pass before the loop
        # (model.py:346-352,197); the step body copies its row by the device step word.  The table is an optimisation worth
        # ~0.2 % of a chain, so it never gets to fail one: it is kept inside ND_EMBED_TABLE_MAX_GB (default 4) AND a tenth
        # of the memory free on the device right now, and an allocation failure falls back to K1/K2 inside the forward
        # (ND_HOIST_EMBED=0 keeps them there always).  It lives as long as the captured graph that reads it: dropped with
        # the loop state (another plan, EMA swap, ``release()``)
        lo = first - steps_to_do + 1
        etab = None
        if (plan.e_all is not None and os.environ.get('ND_HOIST_EMBED', '1') != '0' and steps_to_do > 1 and (NI * plan.e_ld) % 4 == 0):
            cap = float(os.environ.get('ND_EMBED_TABLE_MAX_GB', '4')) * 2 ** 30
            have = plan._etab is not None and plan._etab['S'] == steps_to_do
            if not have:
                cap = min(cap, 0.1 * torch.cuda.mem_get_info(dev)[0])
            if plan.embed_table_bytes(steps_to_do) <= cap:
                try:
                    etab = plan.embed_table(st['tmap'][lo:first + 1])
                except torch.cuda.OutOfMemoryError:
                    plan.drop_embed_table()
                    st['graph'] = None
                    etab = None
        row_floats = NI * plan.e_ld

        eps_ptr = plan.out.data_ptr()
        eps_u_ptr = plan.out.data_ptr() + 4 * B * HW * plan.Cout_p if cfg else None
        w = float(self.strength) if cfg else 0.0
        eta = float(self.ddim_eta) if self.use_ddim else 0.0
        var_kind = self._var_kind()
        xp = plan.x_in.data_ptr()

        def body():
            s = torch.cuda.current_stream().cuda_stream
            if etab is not None:
                _hip.check(lib.nd_copy_row_by_step(etab.data_ptr(), st['step'].data_ptr(), lo, steps_to_do, row_floats,
                                                   plan.e_all.data_ptr(), s), 'nd_copy_row_by_step')
                plan.run(skip_embed=True)
            else:
                _hip.check(lib.nd_fill_timestep(st['tmap'].data_ptr(), st['step'].data_ptr(), plan.t_in.data_ptr(), NI, s),
                           'nd_fill_timestep')
                plan.run()
            if self.use_ddim:
                rc = lib.nd_ddim_step(xp, xp, xdup, None, 0, plan.Cin_p, eps_ptr, eps_u_ptr, plan.Cout_p, w, st['coef'].data_ptr(),
                                      st['step'].data_ptr(), eta, noise_ptr, noise_stride, 0, seed_ptr, first_elem, B, HW, C, s)
            else:
                rc = lib.nd_ddpm_step(xp, xp, xdup, None, 0, plan.Cin_p, eps_ptr, eps_u_ptr, plan.Cout_p, w, st['coef'].data_ptr(),
                                      st['step'].data_ptr(), var_kind, noise_ptr, noise_stride, 0, seed_ptr, first_elem, B, HW, C, s)
            _hip.check(rc, 'sampler step')
            _hip.check(lib.nd_step_advance(st['step'].data_ptr(), -1, s), 'nd_step_advance')

        def snapshot():
            o = torch.empty(B, C, R, R, dtype=torch.float32, device=dev)
            _hip.check(lib.nd_nhwc_to_nchw(xp, o.data_ptr(), B, C, HW, plan.Cin_p,
                                           torch.cuda.current_stream().cuda_stream), 'nd_nhwc_to_nchw')
            return o

        steps = range(steps_to_do)
        bar = None
        if progress:
            import tqdm
            bar = tqdm.tqdm(total=steps_to_do)
        use_graph = self.use_graph and trace is None and steps_to_do > 1
        gkey = (self.use_ddim, cfg, eta, w, var_kind, noise_ptr, noise_stride, first_elem,
                None if etab is None else (etab.data_ptr(), lo, steps_to_do))
        done = 0
        if use_graph:
            if st['graph'] is None or st['graph_key'] != gkey:
                body()                               # first step eagerly: warms up lazy kernel attributes
                done = 1
                if bar is not None:
                    bar.update(1)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                st['graph'], st['graph_key'] = g, gkey
            g = st['graph']
            for _ in range(done, steps_to_do):
                g.replay()
                if bar is not None:
                    torch.cuda.synchronize()
                    bar.update(1)
        else:
            for _ in steps:
                body()
                if trace is not None:
                    trace.append(snapshot())
                if bar is not None:
                    torch.cuda.synchronize()
                    bar.update(1)
        if bar is not None:
            bar.close()
        return snapshot()

    # ---------------------------------------------------------------------------------------------- multi-GPU
    @torch.no_grad()
    def denoise_sharded(self, x, kwargs=None, noise=None, **kw):
        """Batch-sharded ``denoise``: rank r of N (torch.distributed, RCCL on AMD GPUs) denoises rows
        [r*B/N, (r+1)*B/N) of the GLOBAL ``x`` / labels / noise and the finished samples are all-gathered, so every
        rank returns the full [B, C, R, R] result and an N-rank run is comparable row by row with a 1-rank run.
        The loop itself needs no communication (nothing on the path mixes samples)."""
        from .parallel import shard_slice, all_gather_rows
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return self.denoise(x=x, kwargs=kwargs, batch_size=x.shape[0], noise=noise, **kw)
        # (a world of ONE still takes the sharded route: a one-GPU box runs the broadcast and the all-gather on RCCL)
        rank, world = dist.get_rank(), dist.get_world_size()
        sl = shard_slice(x.shape[0], rank, world)
        lk = None
        if kwargs is not None:
            lk = {k: (v[sl] if v is not None else None) for k, v in kwargs.items()}
        ln = None if noise is None else noise[:, sl]
        # in-kernel noise: every rank must use the same Philox seed (drawn here from torch's CPU generator, which the
        # ranks seed identically to build the same global x) and continue the global element count at its first row
        saved = (getattr(self, 'seed', None), getattr(self, 'first_row', 0))
        if saved[0] is None:
            self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.first_row = sl.start
        model = getattr(self, 'model', None)
        if model is not None and hasattr(model, '_plan') and getattr(self, 'device', torch.device('cpu')).type == 'cuda':
            from .parallel import tune_on_rank0
            # every rank runs the kernels rank 0 measured fastest.  The plans are keyed by the FORWARD batch: rows of the
            # shard, twice that under classifier-free guidance (_run_loop), and ragged shards have two sizes -- rank 0
            # measures every one of them, so no rank tunes on its own
            f = 2 if self.guidance == 'classifier_free' else 1
            rows = [shard_slice(x.shape[0], r, world) for r in range(world)]
            sizes = sorted({f * (s_.stop - s_.start) for s_ in rows if s_.stop > s_.start})
            with torch.cuda.device(self.device):
                tune_on_rank0(model, sizes, own=f * (rows[0].stop - rows[0].start))
        try:
            local = self.denoise(x=x[sl], kwargs=lk, batch_size=sl.stop - sl.start, noise=ln, **kw)
        finally:
            self.seed, self.first_row = saved
        return all_gather_rows(local, x.shape[0], rank, world)

    def loss(self, *a, **k):
        raise NotImplementedError('training loss is outside this build (sampling hot path only)')
