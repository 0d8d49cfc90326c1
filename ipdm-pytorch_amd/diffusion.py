"""GaussianDiffusion: host-side mirror of the reference's Model/model.py:376-642 on the dense
sampling path.  Control flow (passes, steps, guidance scheduling) is Python as in the reference;
every tensor operation is a libipdm_hip.so call on device-resident buffers -- nothing goes through
the host inside the loop (the reference does a D2H/np.vectorize/numba/H2D round trip per step in
adaptive mode, Model/model.py:554-560).

Semantics differences, by design: all statistics (`std`, `median`) are per slice, i.e.
a batch of B slices gives exactly what the reference gives when called B times with B=1.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call, lib, ptr

# np.polyfit coefficients of the guidance curves (Utils/train_test_utils.py:842-865), highest power
# first; values as produced by the reference (tests/golden/misc.npz holds the same numbers).
CURVE_COEFFS = {
    "img": ([170.45454545463878, -857.3232323237245, 1588.825757576721, -1314.8304473312783, 432.87337662364365],
            [0.7496994267099147, -4.199781115690005, 5.908637798542919]),
    "proj": ([-71.02272727288062, 417.6136363644583, -893.418560607694, 800.875270564197, -234.09496753293],
             [2.3612714971236124, -14.22455278875205, 21.070551037502682]),
}


def _stream():
    return _lib.current_stream()


def _dcall(t, name, *args):
    """Native call on the device (and that device's current stream) of tensor `t`: opt.device='cuda:1' must work
    without the caller having made it the current device (the reference picks its GPU through opt.device only)."""
    with torch.cuda.device(t.device):
        return call(name, *args, _stream())


def cosine_lambda(ts, power, i):
    """cosine_beta_schedule(ts, schedule_power=power)[i] (Model/model.py:546,552) as a python float."""
    out = C.c_double()
    call("ipdm_cosine_lambda", int(ts), float(power), int(i), C.byref(out))
    return out.value


class NoiseSource:
    """Counter-based N(0,1) source replacing torch.randn_like (Model/model.py:440,509).

    Draw k of global slice s is a pure function of (seed, s, k): the same slice gets the same noise
    whatever the batch composition or the number of GPUs (shard invariance).  `slice_id0` is the
    global index of the first slice of the local batch."""

    def __init__(self, seed=0, slice_id0=0):
        self.seed, self.slice_id0, self.draw = int(seed), int(slice_id0), 0

    def next_like(self, x):
        out = torch.empty_like(x)
        B = x.shape[0]
        _dcall(out, "ipdm_randn", ptr(out), B, x.numel() // B, self.seed, self.slice_id0, self.draw)
        self.draw += 1
        return out


class InjectedNoise:
    """Parity mode: hands out caller-supplied draws (an iterable of tensors shaped like x) in order."""

    def __init__(self, draws):
        self._it = iter(draws)
        self.draw = 0

    def next_like(self, x):
        z = next(self._it).to(x.device, torch.float32).contiguous()
        assert z.shape == x.shape, "injected noise shape %s != %s" % (tuple(z.shape), tuple(x.shape))
        self.draw += 1
        return z


class GaussianDiffusion:
    """Mirror of Model/model.py:376-642 (cosine schedule only -- the only one the harness builds,
    Utils/train_test_utils.py:221-223,243-245)."""

    def __init__(self, timesteps=1000, beta_schedule="cosine", schedule_power=1):
        if beta_schedule != "cosine":
            raise NotImplementedError("only the cosine schedule is on the reference's sampling path")
        self.timesteps = timesteps
        self.schedule_power = schedule_power
        h = C.c_void_p()
        call("ipdm_schedule_create", int(timesteps), float(schedule_power), C.byref(h))
        self._h = h
        self._ws = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                lib().ipdm_schedule_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def coeffs(self, t):
        """(sqrt_ac, sqrt_1m_ac, sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, log_var, var) at t, float32
        (_extract, Model/model.py:424-428)."""
        out = (C.c_float * 8)()
        call("ipdm_schedule_coeffs", self._h, int(t), C.byref(out))
        return tuple(out)

    def _workspace(self, key, nbytes, device):
        w = self._ws.get((key, device))
        if w is None or w.numel() < nbytes:
            w = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self._ws[(key, device)] = w
        return w

    # ---- Model/model.py:438-445
    def q_sample(self, x_start, t, noise):
        x = x_start.contiguous()
        out = torch.empty_like(x)
        _dcall(x, "ipdm_q_sample", self._h, int(t), ptr(x), ptr(noise), ptr(out), x.numel())
        return out

    # ---- Model/model.py:492-515 (per-slice statistics)
    def p_sample_condition(self, model, x_t, x_0, t, clip_denoised=True, lambda_=1.0, noise=None, eps_pred=None):
        """One guided reverse step.  `lambda_` is a python float or a small [B,1,mh,mw] map (nearest-
        upsampled inside the kernel).  `noise`: the N(0,1) draw (a tensor)."""
        B, _, H, W = x_t.shape
        x_t = x_t.contiguous()
        if eps_pred is None:
            eps_pred = model(x_t, int(t))
        ws = self._workspace("step", lib().ipdm_ddpm_workspace_bytes(B), x_t.device)
        out = torch.empty_like(x_t)
        if isinstance(lambda_, torch.Tensor) and lambda_.dim() > 0:
            lm = lambda_.to(x_t.device, torch.float32).contiguous()
            mh, mw = lm.shape[-2], lm.shape[-1]
            _dcall(x_t, "ipdm_ddpm_step", self._h, int(t), ptr(eps_pred), ptr(x_t), ptr(x_0), ptr(noise), ptr(out), B, H, W,
                   0.0, ptr(lm), mh, mw, 1 if clip_denoised else 0, ptr(ws), ws.numel())
        else:
            _dcall(x_t, "ipdm_ddpm_step", self._h, int(t), ptr(eps_pred), ptr(x_t), ptr(x_0), ptr(noise), ptr(out), B, H, W,
                   float(lambda_), None, 0, 0, 1 if clip_denoised else 0, ptr(ws), ws.numel())
        return out

    # ---- guidance map after pass 0 (Model/model.py:574-614)
    def guidance_map(self, x, img, mode, kernel_size, amplitude):
        """Returns (Lambda [B,1,H/k,W/k] f32 -- the curve output, expmax [B] f32)."""
        B, _, H, W = x.shape
        ks = int(kernel_size)
        Lam = torch.empty((B, 1, H // ks, W // ks), dtype=torch.float32, device=x.device)
        emax = torch.empty((B,), dtype=torch.float32, device=x.device)
        ws = self._workspace("guid", lib().ipdm_guidance_workspace_bytes(B, H, W), x.device)
        p1, p2 = CURVE_COEFFS[mode]
        a1 = (C.c_double * 5)(*p1)
        a2 = (C.c_double * 3)(*p2)
        _dcall(x, "ipdm_guidance_map", ptr(x.contiguous()), ptr(img.contiguous()), ptr(Lam), ptr(emax), B, H, W, ks,
               float(amplitude), 0 if mode == "img" else 1, a1, a2, ptr(ws), ws.numel())
        return Lam, emax

    def lambda_ratio(self, Lam, i, ts):
        """condition_lambda_ratio_cuda + clip (Model/model.py:328-351,558) on the small map."""
        out = torch.empty_like(Lam)
        _dcall(Lam, "ipdm_lambda_ratio", ptr(Lam), ptr(out), Lam.numel(), int(i), int(ts))
        return out

    # ---- Model/model.py:517-642
    @torch.no_grad()
    def guided_reverse_process(self, model, img, t_start=None, clip=True, lambda_ratio=1, eta=0.5, save_states=False,
                               mode="img", constant_guidance=None, noise=None, **kwargs):
        """Same signature and return value as the reference: (list of iterates, reverse states,
        noise_strength).  Extra keyword: `noise` = NoiseSource / InjectedNoise (default: NoiseSource(0))."""
        if kwargs.get("only_convertor"):
            return [img], None, None
        normal = bool(kwargs.get("normal"))        # iterates are reported through the inverse power transform (:616-617)
        noise = noise if noise is not None else NoiseSource(0)
        img = img.to(torch.float32).contiguous()
        B = img.shape[0]
        n = img.numel()
        x = img.clone()
        guide = img.clone()
        iters_out, reverse_states = [], []
        adaptive = t_start is None
        t_list = [20] if adaptive else list(t_start)
        noise_strength = None
        it = 0
        Lam = None
        ldct = kwargs.get("ldct")
        while t_list:
            ts = t_list.pop(0)
            x = self.q_sample(x, ts, noise.next_like(x))
            for i in reversed(range(ts)):
                if constant_guidance is None:
                    if it == 0:
                        l_s = cosine_lambda(ts, lambda_ratio, i)
                    else:
                        l_s = self.lambda_ratio(Lam, i, ts)
                else:
                    l_s = constant_guidance
                x = self.p_sample_condition(model, x, guide, i, clip_denoised=clip, lambda_=l_s,
                                            noise=noise.next_like(x))
                if save_states:
                    reverse_states.append(x.detach().cpu().numpy())
            if clip:
                y = torch.empty_like(x)
                _dcall(x, "ipdm_clamp", ptr(x), ptr(y), n, 0 if mode == "img" else 1)
                x = y
            if it == 0 and constant_guidance is None:
                if mode == "img":
                    Lam, emax = self.guidance_map(x, img, "img", kwargs["kernel_size_img"], kwargs["amplitude_img"])
                    if adaptive:
                        ns = kwargs.get("noise_strength")
                        if ns == "high":
                            t_list, eta = [15, 15, 15], 0.6
                        elif ns == "mid":
                            t_list, eta = [15, 12, 10], 0.55
                        else:
                            t_list, eta = [10, 10, 10], 0.5
                else:
                    Lam, emax = self.guidance_map(x, img, "proj", kwargs["kernel_size_proj"], kwargs["amplitude_proj"])
                    if adaptive:
                        m = float(emax.max().item())       # the one device->host scalar of adaptive mode
                        # the reference decides on the whole batch's maximum (delt.max(), :596-609); when the batch is
                        # sharded over ranks the decision must not depend on the sharding: `rank_max` (a scalar MAX
                        # all-reduce, handed in by the denoiser under torch.distributed) makes it global again
                        if kwargs.get("rank_max") is not None:
                            m = float(kwargs["rank_max"](m))
                        if m >= 30:
                            t_list, noise_strength, eta = [30, 25, 20], "high", 0.6
                        elif m >= 4.5:
                            t_list, noise_strength, eta = [20, 18, 15], "mid", 0.5
                        else:
                            t_list, noise_strength, eta = [15, 15, 15], "low", 0.5
            if normal:
                from .normalize import yeo_johnson_inverse_transform
                iters_out.append(yeo_johnson_inverse_transform(x.contiguous(), kwargs["transformer"]).to(torch.float32))
            else:
                iters_out.append(x)
            if constant_guidance is None:
                if it >= 1:
                    guide = self._guide_update(mode, eta, x, img, ldct)
                if it == 0:
                    x = img.clone()
            else:
                guide = self._guide_update(mode, eta, x, img, ldct)
            it += 1
        if len(iters_out) > 1:
            avg = torch.empty_like(iters_out[-1])
            _dcall(avg, "ipdm_axpbypcz", ptr(iters_out[-1]), ptr(iters_out[-2]), None, ptr(avg), n, 0.5, 0.5, 0.0)
            iters_out.append(avg)
        if adaptive:
            return iters_out[1:], reverse_states, noise_strength
        return iters_out, reverse_states, noise_strength

    # ---- Model/model.py:654-725
    @torch.no_grad()
    def ddim_sample(self, sample_img, model, condition, t_start, condition_lambda=0.5, batch_size=1, ddim_timesteps=2,
                    ddim_discr_method="uniform", ddim_eta=0.0, clip_denoised=True, noise=None):
        """Same signature as the reference (+ `noise`).  One draw is consumed per step even when ddim_eta == 0,
        as the reference's torch.randn_like call does (:716)."""
        import numpy as np
        if ddim_discr_method == "uniform":
            seq = np.linspace(t_start - 1, 0, ddim_timesteps + 1).astype(int)[0:-1]
        elif ddim_discr_method == "quad":
            seq = ((np.linspace(0, np.sqrt(self.timesteps * .8), ddim_timesteps)) ** 2).astype(int)
        else:
            raise NotImplementedError('There is no ddim discretization method called "%s"' % ddim_discr_method)
        prev_seq = np.append(seq[1:], np.array([0]))
        noise = noise if noise is not None else NoiseSource(0)
        x = sample_img.to(torch.float32).contiguous()
        cond = condition.to(torch.float32).contiguous()
        B = x.shape[0]
        ws = self._workspace("step", lib().ipdm_ddpm_workspace_bytes(B), x.device)
        for i in range(ddim_timesteps):
            t, tp = int(seq[i]), int(prev_seq[i])
            eps_pred = model(x, t)
            z = noise.next_like(x)
            out = torch.empty_like(x)
            _dcall(x, "ipdm_ddim_step", self._h, t, tp, ptr(eps_pred), ptr(x), ptr(cond), ptr(z) if ddim_eta != 0 else None,
                   ptr(out), B, x.numel() // B, float(condition_lambda), float(ddim_eta), 1 if clip_denoised else 0,
                   ptr(ws), ws.numel())
            x = out
        return x

    # ---- Model/model.py:727-759
    @torch.no_grad()
    def sparse_guided_reverse_process(self, model, condition, t_start, condition_lambda_max=0.5, condition_lambda_min=0.25,
                                      batch_size=1, ddim_timesteps=(2,), ddim_discr_method="uniform", ddim_eta=0.0,
                                      eta=0.5, clip_denoised=True, noise=None):
        """The sparse (DDIM) sampler: same signature and return value (list of per-pass results) as the reference."""
        import numpy as np
        noise = noise if noise is not None else NoiseSource(0)
        condition = condition.to(torch.float32).contiguous()
        sample_img = self.q_sample(condition, t_start[0], noise.next_like(condition))
        condition_ = condition.clone()
        n_it = len(t_start)
        step = (condition_lambda_max - condition_lambda_min) / n_it
        lam = np.arange(condition_lambda_max, condition_lambda_min - step, -step)
        result = []
        for i, t in enumerate(t_start):
            sample_img = self.ddim_sample(sample_img=sample_img, model=model, condition=condition, t_start=t,
                                          condition_lambda=lam[i], batch_size=batch_size, ddim_timesteps=ddim_timesteps[i],
                                          ddim_discr_method=ddim_discr_method, ddim_eta=ddim_eta, clip_denoised=clip_denoised,
                                          noise=noise)
            nxt = torch.empty_like(sample_img)
            _dcall(nxt, "ipdm_axpbypcz", ptr(sample_img), ptr(condition_), None, ptr(nxt), sample_img.numel(), float(eta),
                   float(1 - eta), 0.0)
            condition = nxt
            result.append(sample_img.clone())
        return result

    def _guide_update(self, mode, eta, x, img, ldct):
        """Model/model.py:625-635."""
        out = torch.empty_like(x)
        if mode == "proj":
            _dcall(x, "ipdm_axpbypcz", ptr(x), ptr(img), None, ptr(out), x.numel(), float(eta), float(1 - eta), 0.0)
        else:
            ld = ldct.to(x.device, torch.float32).contiguous()
            _dcall(x, "ipdm_axpbypcz", ptr(x), ptr(img), ptr(ld), ptr(out), x.numel(), float(eta), float(0.95 - eta), 0.05)
        return out
